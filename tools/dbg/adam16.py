import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd import convex_adam_utils as U
from oracle import oracle as orc
orc.build()
DEV = "cuda:0"
rng = np.random.default_rng(1)
C, shp2 = 12, (18, 16, 74)
P0 = (0.5 * rng.standard_normal((3,) + shp2)).astype(np.float32)
dev = lambda a: torch.from_numpy(a).to(DEV)
h = lambda a: a.astype(np.float16).astype(np.float32)
def run(name, F2, M2):
    for storage in ("fp32", "fp16"):
        Ud, st = U.adam_run(dev(F2)[None], dev(M2)[None], dev(P0)[None], 1.25, 1, return_state=True, storage=storage)
        r = orc.adam_run(h(F2), h(M2), P0, 1.25, 1, want_grad=True)
        g = st["G"].cpu().numpy()[0]
        print(name, storage, "G equal", np.array_equal(g, r["G"]), "rel", np.abs(g - r["G"]).max() / np.abs(r["G"]).max())
one = np.ones((C,) + shp2, np.float32)
run("const F=M=0.5", 0.5 * one, 0.5 * one)
run("const F=0.5 M=0.25", 0.5 * one, 0.25 * one)
ramp = (np.arange(shp2[2], dtype=np.float32) / 128)[None, None, None, :] * one
run("ramp x, same", ramp, ramp)
chan = (np.arange(C, dtype=np.float32) / 16)[:, None, None, None] * one
run("per channel const", chan, 0.5 * chan)
rz = (np.arange(shp2[0], dtype=np.float32) / 32)[None, :, None, None] * one
run("ramp z", rz, rz)
run("random multiples of 1/256", np.round(rng.random((C,) + shp2) * 256).astype(np.float32) / 256, np.round(rng.random((C,) + shp2) * 256).astype(np.float32) / 256)
