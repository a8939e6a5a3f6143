#!/usr/bin/env python
"""cupy_hd95 (surface method) on full-size label pairs of growing misregistration: ms per call, and equality with the transform method.
   python tools/experiments/time_hd95.py [amp ...]      (amp = warp amplitude of phantom.warped_label_pair, 0.05 ~ 4-5 voxels)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd.phantom import warped_label_pair   # noqa: E402
from convexadam_amd import convexAdam_hyper_util as HU   # noqa: E402

dev = torch.device("cuda", 0)
shape = (160, 192, 224)
for amp in [float(a) for a in sys.argv[1:]] or [0.005, 0.02, 0.05, 0.1]:
    fx, mv = warped_label_pair(shape, 18, 11, amp)
    fx, mv = fx.to(dev), mv.to(dev)
    NL = int(os.environ.get("HD95_LABELS", "16"))          # 17 = with the label planted in a corner (84 voxels from its counterpart: the call falls back to the transforms)
    fx, mv = torch.where(fx > NL, torch.zeros_like(fx), fx), torch.where(mv > NL, torch.zeros_like(mv), mv)
    ref = HU.cupy_hd95(fx, mv, NL, method="edt")
    got = HU.cupy_hd95(fx, mv, NL, method="surface")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        HU.cupy_hd95(fx, mv, NL, method="surface")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("amp %.3f: %.2f ms per call, mean hd95 %.3f, equal to the transform method: %s" % (amp, dt * 1e3, float(got.mean()), bool(torch.equal(ref, got))), flush=True)
