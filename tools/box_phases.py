#!/usr/bin/env python
"""Per-phase shader clocks inside the steps of the marching three-box kernels (experiment build, tools/box_phases.sh):
   CONVEXADAM_HIP_LIB=convexadam_amd/csrc/libconvexadam_hip_phases.so python tools/box_phases.py"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convexadam_amd import convex_adam_utils as U
from convexadam_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
h, w, d = 80, 96, 112
F2 = torch.rand(1, 12, h, w, d, generator=g).to(dev); M2 = torch.rand(1, 12, h, w, d, generator=g).to(dev)
P0 = Fn.interpolate(torch.randn(1, 3, 5, 6, 7, generator=g) * 2.0, size=(h, w, d), mode="trilinear").to(dev)
U.adam_run(F2, M2, P0, 1.25, 5)
torch.cuda.synchronize()
n_census = 4 * (8192 + 4096)
buf = torch.zeros(n_census + 1024 * 16 * 8 * 2, dtype=torch.int64, device=dev)
L.cvx_set_option(b"census_ptr", buf.data_ptr())
U.adam_run(F2, M2, P0, 1.25, 5)
torch.cuda.synchronize()
L.cvx_set_option(b"census_ptr", 0)
a = buf.cpu().numpy().astype(np.uint64)
names = ["loader + LDS window reads", "27-tap sums", "division + stage store", "Adam update", "barrier wait"]
for kname, off in (("forward boxes", n_census), ("adjoint boxes + Adam", n_census + 1024 * 16 * 8)):
    ph = a[off: off + 1024 * 16 * 8].reshape(1024, 16, 8).astype(np.float64)
    print(kname)
    for K in range(3):
        sel = ph[:, K * 4: K * 4 + 4, :].reshape(-1, 8)
        sel = sel[sel[:, 5] > 0]
        if not len(sel):
            continue
        per = sel[:, :5] / sel[:, 5:6]
        tot = per.sum(1)
        print("  pass %d: %5d waves, clocks per emit step: " % (K + 1, len(sel)) + "; ".join("%s %.0f" % (names[i], np.median(per[:, i])) for i in range(5)) + "; total %.0f" % np.median(tot))
        old = sel[:len(sel) // 2]
