"""CPU: EPE of the oracle's fast Adam mode vs the reference capture (tests/golden/fullsize.npz) at 1/20/40/80 iterations."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
from oracle import oracle as orc
from convexadam_amd.phantom import deformed_pair
g = np.load("tests/golden/fullsize.npz")
shape = (160, 192, 224)
cache = "/tmp/cvx_fullsize_stage.npz"
if os.path.exists(cache):
    c = np.load(cache); F2, M2, P0 = c["F2"], c["M2"], c["P0"]
else:
    fix, mov = deformed_pair(shape, 0, 4.0)
    kw = dict(mind_r=1, mind_d=2, grid_sp=6, disp_hw=6, grid_sp_adam=2, ic=True)
    t = time.time()
    out, st = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), lambda_weight=1.25, selected_niter=1, return_stages=True, **kw)
    print("pipeline", time.time() - t)
    F2, M2, P0 = st["F2"], st["M2"], st["P0"]
    np.savez(cache, F2=F2, M2=M2, P0=P0)
s = int(g["sub"])
def epe(a, b): return float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(0)).mean())
modes = sys.argv[1:] or ["exact", "fast"]
for mode in modes:
    for n in (1, 20, 40, 80):
        t = time.time()
        if mode == "fast_f16":          # fast arithmetic on features rounded once to half precision
            h = lambda x: x.astype(np.float16).astype(np.float32)  # noqa: E731
            r = orc.adam_run(h(F2), h(M2), P0, 1.25, n, mode="fast", keep_last_step=False)
        else:
            r = orc.adam_run(F2, M2, P0, 1.25, n, mode=mode, keep_last_step=False) if mode != "exact" else orc.adam_run(F2, M2, P0, 1.25, n)
        f = orc.resize_trilinear(r["U"] * np.float32(2), shape)
        print(mode, n, "epe_sub %.4e" % epe(f[:, ::s, ::s, ::s], g["c1_adam_%d_sub" % n]), "t=%.1fs" % (time.time() - t), flush=True)
