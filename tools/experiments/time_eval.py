"""Per-component device time of the sweep's evaluation (evaluate_item) on the sweep's synthetic label maps.
    python tools/experiments/time_eval.py [H W D]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from convexadam_amd import convexAdam_hyper_util as HU  # noqa: E402
from convexadam_amd import sweep  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (160, 192, 224)
dev = torch.device("cuda:0")
seg_f, seg_m, kf, km, nl = sweep._make_labels(shape, 0, dev)
disp = torch.zeros((3,) + shape, device=dev)
disp[0], disp[1], disp[2] = 2.0, -1.0, 3.0
disp += 0.3 * torch.randn_like(disp)


def timed(name, fn, n=10):
    fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print("%-28s %7.3f ms" % (name, (time.time() - t) / n * 1e3), flush=True)
    return r


d = disp[None]
jac = timed("jacobian_determinant_3d", lambda: HU.jacobian_determinant_3d(d, False))
timed("jacobian stats", lambda: HU.jacobian_log_std_and_folding(jac))
warped = timed("warp_labels_nearest", lambda: HU.warp_labels_nearest(seg_m, d))
timed("dice_coeff", lambda: HU.dice_coeff(seg_f, warped, nl + 1))
timed("tre_at_keypoints", lambda: HU.tre_at_keypoints(d, kf, km))
cache = {}
timed("cupy_hd95 surface (cached)", lambda: HU.cupy_hd95(seg_f, warped, nl, fixed_cache=cache))
timed("cupy_hd95 surface (no cache)", lambda: HU.cupy_hd95(seg_f, warped, nl))
cache_e = {}
timed("cupy_hd95 edt (cached)", lambda: HU.cupy_hd95(seg_f, warped, nl, fixed_cache=cache_e, method="edt"))
timed("evaluate_item", lambda: sweep.evaluate_item(disp, seg_f, seg_m, kf, km, nl, cache=cache))
surf = lambda s: int(((s[1:] != s[:-1]).sum() + (s[:, 1:] != s[:, :-1]).sum() + (s[:, :, 1:] != s[:, :, :-1]).sum()))  # noqa: E731
print("label faces between different values:", surf(seg_f), "of", seg_f.numel(), "voxels")
