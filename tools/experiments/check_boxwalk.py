"""k_box_walk == k_box_zero bit for bit (option box_walk 1 / 0) over shapes and kernel sizes; timing at the sweep's grid sizes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd import convex_adam_utils as U, _lib
L = _lib.lib(); dev = torch.device("cuda", 0); g = torch.Generator().manual_seed(5)
bad = 0
for shape in [(80, 96, 112), (13, 9, 8), (5, 3, 4), (2, 2, 4), (37, 40, 124), (14, 31, 52), (17, 5, 36), (160, 192, 224)]:
    x = (torch.randn(1, 3, *shape, generator=g) * 2).to(dev)
    x[0, 0, : shape[0] // 2] = 0.0
    for k in (3, 5, 7):
        outs = []
        for v in (0, 1):
            L.cvx_set_option(b"box_walk", v)
            outs.append(U.box_smooth(x, k, 1).clone())
        if not torch.equal(outs[0], outs[1]):
            bad += 1; d = (outs[0] - outs[1]).abs(); print("MISMATCH", shape, k, float(d.max()), int((d > 0).sum()))
print("RESULT", "bit-identical" if not bad else "%d mismatches" % bad)
for shape in [(80, 96, 112), (160, 192, 224)]:
    x = torch.randn(1, 3, *shape, generator=g).to(dev)
    for k in (3, 5):
        for v in (0, 1):
            L.cvx_set_option(b"box_walk", v)
            U.box_smooth(x, k, 1); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): U.box_smooth(x, k, 1)
            e1.record(); torch.cuda.synchronize()
            print(shape, "k", k, "walk" if v else "naive", "%.1f us" % (e0.elapsed_time(e1) * 100))
L.cvx_set_option(b"box_walk", 1)
