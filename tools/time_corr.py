#!/usr/bin/env python
"""Correlation stage (cvx_correlate_f32) fused vs round-1 kernels at the shapes the callers produce:
   python tools/time_corr.py            C:h:w:d:hw ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convexadam_amd import _lib  # noqa: E402
from convexadam_amd._lib import lib, ptr, stream_ptr, workspace  # noqa: E402

dev = torch.device("cuda", 0)
specs = sys.argv[1:] or ["12:26:32:37:6", "32:26:32:37:6", "18:26:32:37:4", "12:32:38:44:5", "12:40:48:56:4", "12:53:64:74:3", "12:80:96:112:2", "12:37:32:37:8"]
L = lib()
for spec in specs:
    C, h, w, d, hw = (int(v) for v in spec.split(":"))
    K = (2 * hw + 1) ** 3
    f = torch.rand(C, h, w, d, device=dev); m = torch.rand(C, h, w, d, device=dev)
    ssd = torch.empty((K, h, w, d), device=dev)
    res = []
    for unf in (0, 1):
        L.cvx_set_option(b"corr_unfused", unf)
        nws = L.cvx_correlate_workspace_bytes(C, h, w, d, hw)
        ws = workspace(nws, dev)
        rc = L.cvx_correlate_f32(ptr(f), ptr(m), C, h, w, d, hw, ptr(ssd), None, ptr(ws), nws, stream_ptr(dev))
        if rc != 0:
            res.append(float("nan")); continue
        torch.cuda.synchronize()
        ref = ssd.clone() if unf == 0 else ref
        same = bool(torch.equal(ssd, ref))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.cvx_correlate_f32(ptr(f), ptr(m), C, h, w, d, hw, ptr(ssd), None, ptr(ws), nws, stream_ptr(dev))
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    L.cvx_set_option(b"corr_unfused", 0)
    alg = (K * h * w * d * 4 + 2 * C * h * w * d * 4) / 1e9
    print("C %2d  %3dx%3dx%3d hw %d  K*v*4 = %7.1f MB   fused %7.3f ms (%.2f TB/s, %.3f of 8)   round-1 kernels %7.3f ms   same bits %s" % (
        C, h, w, d, hw, alg * 1e3, res[0], alg / res[0], alg / res[0] / 8, res[1], same), flush=True)
