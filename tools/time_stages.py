#!/usr/bin/env python
"""Per-stage device time of the benchmark pair for option sets:  python tools/time_stages.py "" "mind_overlap=1" "adam_mode=fast,fbox_tile=1"
(names that are keyword arguments of register_pair_device -- adam_mode, corr_mode, storage -- go to the call, the rest to cvx_set_option)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convexadam_amd import _lib  # noqa: E402
from convexadam_amd.convex_adam_MIND import last_profile, register_pair_device, set_profiling  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
fix, mov = bench.make_pair(dev, 0)
ref = None
for spec in sys.argv[1:] or [""]:
    opts = dict(kv.split("=") for kv in spec.split(",") if kv)
    cfg = dict(bench.CFG)
    for k in ("adam_mode", "corr_mode", "storage"):
        if k in opts:
            cfg[k] = opts.pop(k)
    old = {k: L.cvx_get_option(k.encode()) for k in opts}
    for k, v in opts.items():
        assert L.cvx_set_option(k.encode(), int(v)) == 0, k
    try:
        for _ in range(3):
            out = register_pair_device(fix, mov, **cfg)
        torch.cuda.synchronize()
        ref = out.clone() if ref is None else ref
        if not torch.equal(out, ref):
            print("   mean EPE vs the first spec: %.3e" % float((out - ref).square().sum(0).sqrt().mean()))
        t0 = time.perf_counter()
        for _ in range(10):
            register_pair_device(fix, mov, **cfg)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        set_profiling(2)
        for _ in range(5):
            register_pair_device(fix, mov, **cfg)
        torch.cuda.synchronize()
        st = {}
        for name, t in last_profile():
            st.setdefault(name, []).append(t)
        set_profiling(0)
        print("%-40s %.3f ms/pair  same bits %s  " % (spec or "(default)", ms, bool(torch.equal(out, ref))) + " ".join("%s %.3f" % (k, sum(v) / len(v)) for k, v in st.items()), flush=True)
    finally:
        for k, v in old.items():
            L.cvx_set_option(k.encode(), v)
