#!/usr/bin/env python
"""Throughput of cvx_register_pairs_f32 at the benchmark configuration for several (pairs per call, internal streams)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convexadam_amd.convex_adam_MIND import register_pair_device, register_pairs_device  # noqa: E402

dev = torch.device("cuda", 0)
pairs = [bench.make_pair(dev, i) for i in range(2)]
specs = [tuple(int(v) for v in a.split(":")) for a in (sys.argv[1:] or ["1:1", "2:2", "3:3", "4:4", "4:2", "8:4"])]
for n, ns in specs:
    fx = [pairs[i % 2][0] for i in range(n)]
    mv = [pairs[i % 2][1] for i in range(n)]
    outs = [torch.empty((3,) + bench.SHAPE, dtype=torch.float32, device=dev) for _ in range(n)]
    for _ in range(2):
        register_pairs_device(fx, mv, outs=outs, n_streams=ns, **bench.CFG)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 6
    for _ in range(reps):
        register_pairs_device(fx, mv, outs=outs, n_streams=ns, **bench.CFG)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%d pairs per call on %d streams: %.2f ms per call, %.2f ms per pair, %.1f pairs/s" % (n, ns, dt * 1e3, dt * 1e3 / n, n / dt), flush=True)
