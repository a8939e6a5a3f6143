"""us per Adam iteration against the number of voxels (is a pair dimension inside the Adam kernels worth building?  VERDICT r3 item 6):
the half-resolution grid of the headline pair, then the same grid stacked 2 / 3 / 4 times along H (the work of B pairs in one launch)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from convexadam_amd import convex_adam_utils as U  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
for mode in ("fast", "exact"):
    for B in (1, 2, 3, 4):
        h, w, d = 80 * B, 96, 112
        F = torch.rand((1, 12, h, w, d), generator=g).to(dev)
        M = torch.rand((1, 12, h, w, d), generator=g).to(dev)
        P0 = (torch.rand((1, 3, h, w, d), generator=g) - 0.5).to(dev)
        U.adam_run(F, M, P0, 1.25, 5, mode=mode)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t = time.time()
            U.adam_run(F, M, P0, 1.25, 80, mode=mode)
            torch.cuda.synchronize()
            best = min(best, time.time() - t)
        print("%-5s B=%d grid %dx%dx%d: %.1f us/iteration, %.1f us per iteration and pair" % (mode, B, h, w, d, best / 80 * 1e6, best / 80 * 1e6 / B), flush=True)
