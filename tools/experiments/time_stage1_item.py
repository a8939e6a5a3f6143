"""Per-stage device time of the sweep's heaviest stage-1 settings (convex stage only, lambda 0) at 160 x 192 x 224."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from convexadam_amd import sweep  # noqa: E402
from convexadam_amd.convex_adam_MIND import last_profile, register_pair_device, set_profiling  # noqa: E402

dev = torch.device("cuda", 0)
fix, mov = sweep._make_pair((160, 192, 224), 0, dev)
which = [int(v) for v in sys.argv[1:]] or [1, 0, 4]
cfgs = sweep.stage1_settings(12)
for i in which:
    cfg = cfgs[i]
    for _ in range(2):
        register_pair_device(fix, mov, lambda_weight=0, ic=True, **cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        register_pair_device(fix, mov, lambda_weight=0, ic=True, **cfg)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    set_profiling(2)
    register_pair_device(fix, mov, lambda_weight=0, ic=True, **cfg)
    torch.cuda.synchronize()
    st = {}
    for name, t in last_profile():
        st[name] = st.get(name, 0.0) + t
    set_profiling(0)
    print(i, cfg, "%.2f ms" % ms, {k: round(v, 3) for k, v in st.items()}, flush=True)
