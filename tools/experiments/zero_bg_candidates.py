#!/usr/bin/env python
"""Which voxels of the zero-background pair make k_cert_plain_resolve slow?  Fast (unscaled) volume of the coarse features, then per voxel the number
of entries an exact evaluation would be asked for: s != 0 and lower(s) <= upper(min)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from convexadam_amd._lib import CorrOpts, lib, ptr, stream_ptr, workspace  # noqa: E402
from convexadam_amd.convex_adam_utils import mind_pooled  # noqa: E402
from convexadam_amd.phantom import ellipsoid_mask  # noqa: E402

L = lib()
dev = torch.device("cuda", 0)
fix, mov = bench.make_pair(dev, 0)
m = ellipsoid_mask(bench.SHAPE, 0.3).to(dev)
fz, mz = (fix * m).contiguous(), (mov * m).contiguous()
f = mind_pooled(fz[None, None], 1, 2, 6, 0, device=dev)[0].contiguous()
g = mind_pooled(mz[None, None], 1, 2, 6, 0, device=dev)[0].contiguous()
Cn, h, w, d = f.shape
hw = 6; K = 13 ** 3
ssd = torch.empty((K, h, w, d), device=dev)
nws = L.cvx_correlate_workspace_bytes(Cn, h, w, d, hw)
ws = workspace(nws, dev)
opts = CorrOpts(0, 2, 2, 0)
assert L.cvx_correlate_ex_f32(ptr(f), ptr(g), Cn, h, w, d, hw, C.byref(opts), ptr(ssd), None, ptr(ws), nws, stream_ptr(dev)) == 0
torch.cuda.synchronize()
s = ssd.reshape(K, -1)
CLO, CHI, TINY = np.float32((1 - 1 / 65536) / 729), np.float32((1 + 1 / 65536) / 729), np.float32(1e-40)
s1 = s.min(0).values
U = s1 * float(CHI) + torch.minimum(s1, torch.tensor(float(TINY), device=dev))
lower = torch.clamp(s * float(CLO) - float(TINY), min=0)
cand = ((s != 0) & (lower <= U[None])).sum(0)
tiny = ((s > 0) & (s < 1e-36)).any(0)
second = torch.where(s == s1[None], torch.full_like(s, float("inf")), s).min(0).values
print("voxels %d; min == 0: %d; columns with an entry in (0, 1e-36): %d; max candidates per voxel %d" % (s.shape[1], int((s1 == 0).sum()), int(tiny.sum()), int(cand.max())))
top = torch.argsort(cand, descending=True)[:12]
for x in top.tolist():
    col = s[:, x]
    print("  voxel %6d (z %2d y %2d x %2d): min %.4g  candidates %4d  zeros %4d  tiny %s  distinct values %d  second %.4g" %
          (x, x // (w * d), (x // d) % w, x % d, float(s1[x]), int(cand[x]), int((col == 0).sum()), bool(tiny[x]), int(torch.unique(col).numel()), float(second[x])))
