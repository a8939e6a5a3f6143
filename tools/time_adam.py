#!/usr/bin/env python
"""A/B timing of the Adam loop (cvx_adam_run_f32) at the benchmark's control-grid size, one line per option set:
    python tools/time_adam.py "" "box_cpt=2" "box_cpt=2,box_wg_target=1024"
Prints us / iteration (80 iterations between two events on the launch stream) and whether P, U after 5 iterations are
bit-identical to the first option set.  Run it under rocprofv3 --kernel-trace --stats for per-kernel durations."""
import hashlib
import sys
import os

import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convexadam_amd import convex_adam_utils as U   # noqa: E402
from convexadam_amd import _lib                     # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
h, w, d = (int(v) for v in os.environ.get("ADAM_SHAPE", "80,96,112").split(","))
C = int(os.environ.get("ADAM_C", "12"))
F2 = torch.rand(1, C, h, w, d, generator=g).to(dev)
M2 = torch.rand(1, C, h, w, d, generator=g).to(dev)
P0 = Fn.interpolate(torch.randn(1, 3, 5, 6, 7, generator=g) * 2.0, size=(h, w, d), mode="trilinear").to(dev)
ref = None
reps = int(os.environ.get("ADAM_REPS", "3"))
mode = os.environ.get("ADAM_MODE", "fast")            # the bench's timed mode
storage = os.environ.get("ADAM_STORAGE", "fp32")
for spec in sys.argv[1:] or [""]:
    opts = dict(kv.split("=") for kv in spec.split(",") if kv)
    old = {k: L.cvx_get_option(k.encode()) for k in opts}
    for k, v in opts.items():
        assert L.cvx_set_option(k.encode(), int(v)) == 0, k
    try:
        out, st = U.adam_run(F2, M2, P0, 1.25, 5, return_state=True, mode=mode, storage=storage)
        torch.cuda.synchronize()
        hsh = hashlib.md5(out.cpu().numpy().tobytes()).hexdigest() + hashlib.md5(st["P"].cpu().numpy().tobytes()).hexdigest()
        ref = ref or hsh
        best = 1e9
        for _ in range(reps):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); U.adam_run(F2, M2, P0, 1.25, 80, return_state=True, mode=mode, storage=storage); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 80 * 1e3)
        print("%-50s %7.1f us / iteration   same bits: %s" % (spec or "(default)", best, hsh == ref), flush=True)
    finally:
        for k, v in old.items():
            L.cvx_set_option(k.encode(), v)
