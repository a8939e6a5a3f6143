#!/usr/bin/env python
"""WHERE does the distance between adam_mode="fast" and the reference's capture of BASELINE configs[2] at 20 Adam
iterations sit (1.25e-3 voxel mean EPE against 3.3e-4 for the exact mode)?  Per-voxel endpoint error on the golden's sample grid (every 8th voxel
of the 224 x 192 x 224 field, tests/golden/fullsize.npz::c3_adam_20_sub), split by region -- inside both masks, inside one, outside both (the
replicate-filled part of the features) -- and by quantile; the same for exact - reference and fast - exact.  Runs on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd import convex_adam_MIND as M  # noqa: E402
from convexadam_amd.phantom import deformed_pair, ellipsoid_mask  # noqa: E402

DEV = torch.device("cuda", 0)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "fullsize.npz"))
s = int(g["sub"])
ref = np.moveaxis(g["c3_adam_20_sub"], 0, -1).astype(np.float64)
shape = (224, 192, 224)
fix, mov = deformed_pair(shape, 3, 10.0)
mf, mm = ellipsoid_mask(shape, 0.35), ellipsoid_mask(shape, 0.35, shift=(4, -3, 5))
kw = dict(lambda_weight=1.25, grid_sp=6, disp_hw=8, selected_niter=20, selected_smooth=0, grid_sp_adam=2, ic=True)
ff, fm = M.extract_features(fix, mov, 1, 2, True, mf, mm, device=DEV, dtype=torch.float32)
out = {}
for mode in ("exact", "fast"):
    f = M.register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], adam_mode=mode, **kw).cpu().numpy()
    out[mode] = np.moveaxis(f[:, ::s, ::s, ::s], 0, -1).astype(np.float64)
    if "full_" + mode not in out:
        out["full_" + mode] = f
a, b = mf.numpy()[::s, ::s, ::s] > 0, mm.numpy()[::s, ::s, ::s] > 0
# distance to the fixed mask's boundary in units of the control grid of the Adam stage (2 voxels): the replicate fill is flat beyond ~the MIND stencil
regions = {"inside both masks": a & b, "fixed mask only": a & ~b, "moving mask only": ~a & b, "outside both": ~a & ~b, "all": np.ones_like(a)}


def epe(x, y):
    return np.sqrt(((x - y) ** 2).sum(-1))


for name, (x, y) in {"fast  - reference": (out["fast"], ref), "exact - reference": (out["exact"], ref), "fast  - exact": (out["fast"], out["exact"])}.items():
    e = epe(x, y)
    print("%s: mean EPE %.3e" % (name, e.mean()))
    for rn, m in regions.items():
        if m.sum() == 0:
            continue
        v = e[m]
        print("    %-18s %6d voxels (%.3f)  mean %.3e  median %.3e  p99 %.3e  max %.3e  share of the total error %.3f" %
              (rn, m.sum(), m.mean(), v.mean(), np.median(v), np.quantile(v, 0.99), v.max(), v.sum() / e.sum()))
    srt = np.sort(e.ravel())[::-1]
    print("    the worst 1 %% of the voxels carry %.3f of the error, the worst 10 %% %.3f" % (srt[: len(srt) // 100].sum() / srt.sum(), srt[: len(srt) // 10].sum() / srt.sum()))
# how large is the field where the modes disagree?  (a voxel-sized step in a flat region: |u| itself is not small there)
d = epe(out["fast"], out["exact"])
mag = np.sqrt((out["exact"] ** 2).sum(-1))
worst = d > np.quantile(d, 0.99)
print("worst 1 %% of fast - exact: mean |u| %.2f voxels (all voxels %.2f); inside both masks %.3f, outside both %.3f" %
      (mag[worst].mean(), mag.mean(), (worst & a & b).sum() / worst.sum(), (worst & ~a & ~b).sum() / worst.sum()))
# full resolution: fast - exact by region (no reference needed)
fe = np.sqrt(((out["full_fast"].astype(np.float64) - out["full_exact"].astype(np.float64)) ** 2).sum(0))
A, B = mf.numpy() > 0, mm.numpy() > 0
for rn, m in {"inside both masks": A & B, "fixed mask only": A & ~B, "moving mask only": ~A & B, "outside both": ~A & ~B}.items():
    print("full resolution fast - exact, %-18s: %8d voxels, mean %.3e  p99 %.3e  max %.3e" % (rn, m.sum(), fe[m].mean(), np.quantile(fe[m], 0.99), fe[m].max()))
