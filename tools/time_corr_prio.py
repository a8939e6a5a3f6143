#!/usr/bin/env python
"""A/B timing of the fused correlation kernel's issue priorities (option cf_prio: base-4 digits first-round raw / box, second-round
raw / box) on the benchmark shape:   python tools/time_corr_prio.py 2020 2031 ...   (digits as written; 2020 = the default)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convexadam_amd._lib import lib, ptr, stream_ptr, workspace  # noqa: E402

dev = torch.device("cuda", 0)
L = lib()
C, h, w, d, hw = (int(v) for v in os.environ.get("CORR_SHAPE", "12:26:32:37:6").split(":"))
K = (2 * hw + 1) ** 3
g = torch.Generator().manual_seed(1)
f = torch.rand(C, h, w, d, generator=g).to(dev); m = torch.rand(C, h, w, d, generator=g).to(dev)
ssd = torch.empty((K, h, w, d), device=dev)
nws = L.cvx_correlate_workspace_bytes(C, h, w, d, hw)
ws = workspace(nws, dev)
ref = None
specs = sys.argv[1:] or ["2020"]
for rnd in range(int(os.environ.get("CORR_ROUNDS", "3"))):
    for spec in specs:
        val = int(spec, 4)
        L.cvx_set_option(b"cf_prio", val)
        for _ in range(3):
            L.cvx_correlate_f32(ptr(f), ptr(m), C, h, w, d, hw, ptr(ssd), None, ptr(ws), nws, stream_ptr(dev))
        torch.cuda.synchronize()
        if ref is None:
            ref = ssd.clone()
        same = bool(torch.equal(ssd, ref))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.cvx_correlate_f32(ptr(f), ptr(m), C, h, w, d, hw, ptr(ssd), None, ptr(ws), nws, stream_ptr(dev))
        e1.record(); torch.cuda.synchronize()
        print("cf_prio %s: %.1f us per call   same bits %s" % (spec, e0.elapsed_time(e1) / 20 * 1e3, same), flush=True)
L.cvx_set_option(b"cf_prio", int("2020", 4))
