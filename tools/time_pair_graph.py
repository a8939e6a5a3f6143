#!/usr/bin/env python
"""Does replaying one whole registration (cvx_register_pair_f32: ~330 launches on one stream) as a captured HIP graph change its
device time?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from convexadam_amd.convex_adam_MIND import register_pair_device
dev = torch.device("cuda", 0)
fix, mov = bench.make_pair(dev, 0)
out = torch.empty((3,) + bench.SHAPE, dtype=torch.float32, device=dev)
for _ in range(3):
    register_pair_device(fix, mov, out=out, **bench.CFG)
torch.cuda.synchronize()
ref = out.clone()
def timed(fn, reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("stream launches: %.3f ms per pair" % timed(lambda: register_pair_device(fix, mov, out=out, **bench.CFG)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    register_pair_device(fix, mov, out=out, **bench.CFG)          # workspace of this stream
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        register_pair_device(fix, mov, out=out, **bench.CFG)
torch.cuda.synchronize()
out.zero_(); graph.replay(); torch.cuda.synchronize()
print("graph replay   : %.3f ms per pair; same bits %s" % (timed(graph.replay), bool(torch.equal(out, ref))))
