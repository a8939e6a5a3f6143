"""End-to-end time of BASELINE configs[2] (224x192x224, masks, hw 8) and configs[3] (32-label features, 160x192x160) in both Adam modes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd import convex_adam_MIND as M
from convexadam_amd.convex_adam_nnUNet import extract_features as label_features
from convexadam_amd.phantom import deformed_pair, ellipsoid_mask, label_phantom
dev = torch.device("cuda:0")
def T(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
shape = (224, 192, 224)
fix, mov = deformed_pair(shape, 3, 10.0)
mf, mm = ellipsoid_mask(shape, 0.35), ellipsoid_mask(shape, 0.35, shift=(4, -3, 5))
kw = dict(lambda_weight=1.25, grid_sp=6, disp_hw=8, selected_niter=80, selected_smooth=0, grid_sp_adam=2, ic=True)
t_feat = T(lambda: M.extract_features(fix, mov, 1, 2, True, mf, mm, device=dev, dtype=torch.float32), 3)
ff, fm = M.extract_features(fix, mov, 1, 2, True, mf, mm, device=dev, dtype=torch.float32)
for mode in ("exact", "fast"):
    print("configs[2] masked 224x192x224 hw 8, 80 its, %s: features %.2f ms + pair %.2f ms" % (mode, t_feat, T(lambda: M.register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], adam_mode=mode, **kw))), flush=True)
del ff, fm
shape = (160, 192, 160)
lab = label_phantom(shape, 32, 3); labm = torch.roll(lab, (2, -3, 1), (0, 1, 2))
t_feat = T(lambda: label_features(lab, labm, device=dev), 3)
ff, fm = label_features(lab, labm, device=dev)
kw = dict(lambda_weight=1.25, grid_sp=6, disp_hw=6, selected_niter=80, selected_smooth=0, grid_sp_adam=2, ic=True)
for mode in ("exact", "fast"):
    print("configs[3] %d-channel label features 160x192x160 hw 6, 80 its, %s: features %.2f ms + pair %.2f ms" % (ff.shape[1], mode, t_feat, T(lambda: M.register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], adam_mode=mode, **kw))), flush=True)
