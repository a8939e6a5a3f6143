#!/usr/bin/env python
"""Coupled-convex stage on the benchmark pair with an exact-zero background (ellipsoid mask): certified path against the exact path, per stage.
CVX_CERT_TRACE=1 prints how many voxels each certified pass listed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from convexadam_amd import _lib  # noqa: E402
from convexadam_amd.convex_adam_MIND import last_profile, register_pair_device, set_profiling  # noqa: E402
from convexadam_amd.phantom import ellipsoid_mask  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
fix, mov = bench.make_pair(dev, 0)
m = ellipsoid_mask(bench.SHAPE, 0.3).to(dev)
fz, mz = (fix * m).contiguous(), (mov * m).contiguous()
cfg = dict(bench.CFG, selected_niter=int(os.environ.get("NITER", "2")))
ref = None
for cert in (0, 1):
    L.cvx_set_option(b"corr_cert", cert)
    for _ in range(2):
        out = register_pair_device(fz, mz, **cfg)
    torch.cuda.synchronize()
    ref = out.clone() if ref is None else ref
    set_profiling(2)
    for _ in range(3):
        register_pair_device(fz, mz, **cfg)
    torch.cuda.synchronize()
    st = {}
    for name, t in last_profile():
        st.setdefault(name, []).append(t)
    set_profiling(0)
    print("corr_cert %d  same bits %s  " % (cert, bool(torch.equal(out, ref))) + " ".join("%s %.3f" % (k, sum(v) / len(v)) for k, v in st.items()), flush=True)
L.cvx_set_option(b"corr_cert", 1)
