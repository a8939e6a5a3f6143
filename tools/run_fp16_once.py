#!/usr/bin/env python
"""One benchmark pair in fp16-storage mode (for rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE: the fused kernel's bytes per launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from convexadam_amd.convex_adam_MIND import register_pair_device
dev = torch.device("cuda", 0)
fix, mov = bench.make_pair(dev, 0)
register_pair_device(fix, mov, storage=os.environ.get("STORAGE", "fp16"), **dict(bench.CFG, selected_niter=2))
torch.cuda.synchronize()
