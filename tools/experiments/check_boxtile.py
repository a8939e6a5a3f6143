"""box_fwd_tile variants == the z-marching forward boxes, bit for bit, over odd shapes (through adam_run: U after n iterations)."""
import os, sys, hashlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd import convex_adam_utils as U, _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
bad = 0
VARS = (0, 1000, 2000, 3000, 1834, 2434, 3566, 3534, 1222, 2274, 3444)
for shape in [(80, 96, 112), (13, 9, 8), (5, 3, 4), (12, 8, 56), (25, 17, 60), (24, 16, 116), (2, 2, 4), (37, 40, 124), (14, 31, 52)]:
    h, w, d = shape
    F2 = torch.rand(1, 5, h, w, d, generator=g).to(dev); M2 = torch.rand(1, 5, h, w, d, generator=g).to(dev)
    P0 = (torch.randn(1, 3, h, w, d, generator=g) * 2).to(dev)
    outs = {}
    for v in VARS:
        L.cvx_set_option(b"box_fwd_tile", v)
        o, st = U.adam_run(F2, M2, P0, 1.25, 3, return_state=True, mode="fast")
        torch.cuda.synchronize()
        outs[v] = (o.clone(), st["P"].clone())
    L.cvx_set_option(b"box_fwd_tile", 0)
    for v in VARS[1:]:
        same = torch.equal(outs[v][0], outs[0][0]) and torch.equal(outs[v][1], outs[0][1])
        if not same:
            bad += 1
            diff = (outs[v][0] - outs[0][0]).abs()
            print("MISMATCH", shape, "variant", v, "max", float(diff.max()), "n", int((diff > 0).sum()), "first", (diff > 0).nonzero()[:3].tolist())
    print(shape, "ok" if not bad else "bad so far %d" % bad, flush=True)
print("RESULT", "all variants bit-identical" if bad == 0 else "%d mismatches" % bad)
