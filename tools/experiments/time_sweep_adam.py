"""Per-iteration time of the sweep's stage-2 Adam loop for a (grid_sp_adam, smoother index) at 160 x 192 x 224:
    python tools/experiments/time_sweep_adam.py 1 6 fast_all"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from convexadam_amd import convex_adam_utils as U  # noqa: E402
from convexadam_amd import sweep  # noqa: E402

gsa, avg, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "fast_all"
dev = torch.device("cuda:0")
h, w, d = 160 // gsa, 192 // gsa, 224 // gsa
g = torch.Generator().manual_seed(1)
F = torch.rand((1, 12, h, w, d), generator=g).to(dev)
M = torch.rand((1, 12, h, w, d), generator=g).to(dev)
P0 = (torch.rand((1, 3, h, w, d), generator=g) - 0.5).to(dev)
sm = sweep.smoother_table()[avg]
U.adam_run(F, M, P0, 1.0, 3, smoother=sm, cost_scale=12.0, mode=mode)
torch.cuda.synchronize()
t = time.time()
n = 20
U.adam_run(F, M, P0, 1.0, n, smoother=sm, cost_scale=12.0, mode=mode)
torch.cuda.synchronize()
print("grid_sp_adam %d smoother %d mode %s: %.1f us per iteration" % (gsa, avg, mode, (time.time() - t) / n * 1e6))
