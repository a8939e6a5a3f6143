#!/usr/bin/env python
"""Do workgroups b and b + 256 of the fused correlation kernel share a CU, and does giving them the two D-shift groups of ONE (dH, dW) pair
(option cf_map = 1: the same moving rows through one L1) shorten the launch?  Certified-fast arithmetic (cvx_corr_opts.fast = 2), benchmark geometry."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd._lib import CorrOpts, lib, ptr, stream_ptr, workspace  # noqa: E402

dev = torch.device("cuda", 0)
L = lib()
Cn, h, w, d, hw = 12, 26, 32, 37, 6
n = 2 * hw + 1
K = n ** 3
g = torch.Generator().manual_seed(1)
f = torch.rand(Cn, h, w, d, generator=g).to(dev); m = torch.rand(Cn, h, w, d, generator=g).to(dev)
ssd = torch.empty((K, h, w, d), device=dev)
nws = L.cvx_correlate_workspace_bytes(Cn, h, w, d, hw)
ws = workspace(nws, dev)
opts = CorrOpts(0, 2, 2, 0)


def call():
    rc = L.cvx_correlate_ex_f32(ptr(f), ptr(m), Cn, h, w, d, hw, C.byref(opts), ptr(ssd), None, ptr(ws), nws, stream_ptr(dev))
    assert rc == 0, rc


def census_offset():
    lpr = (d + 6) // 4; RS = 4 * lpr; ng = 3; dq = RS + 4 * ng + 4; hq, wq = h + 2 * hw, w + 2 * hw
    al = lambda x: (x + 255) // 256 * 256
    used = 0
    for nbytes in (4 * Cn * h * w * RS, 4 * (Cn * hq * wq * dq + 8), 4 * 32 * n):
        used = al(used) + nbytes
    return al(used)


ref = None
for rnd in range(3):
    for cf_map in (0, 1):
        L.cvx_set_option(b"cf_map", cf_map)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        if ref is None:
            ref = ssd.clone()
        same = bool(torch.equal(ssd, ref))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        print("cf_map %d: %.1f us per call (k_corr_prep + kernel)  same bits %s" % (cf_map, e0.elapsed_time(e1) / 20 * 1e3, same), flush=True)
# placement census
for cf_map in (0, 1):
    L.cvx_set_option(b"cf_map", cf_map); L.cvx_set_option(b"cf_census", 1)
    call(); torch.cuda.synchronize()
    off = census_offset()
    nb = 512 if cf_map else n * n * 3
    c = ws[off: off + 32 * nb].cpu().numpy().view(np.uint64).reshape(nb, 4)
    hwid, xcc = c[:, 2].astype(np.int64), c[:, 3].astype(np.int64) & 0xf
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    live = c[:, 0] != 0
    pairs = [(b, b + 256) for b in range(256) if b + 256 < nb and live[b] and live[b + 256]]
    same_cu = sum(key[a] == key[b] for a, b in pairs)
    print("cf_map %d: %d workgroups; (b, b + 256) on the same CU: %d of %d; distinct CUs used %d; duration min / median / max %.1f / %.1f / %.1f us (100 MHz ticks)" %
          (cf_map, int(live.sum()), same_cu, len(pairs), len(set(key[live].tolist())), *(np.quantile((c[live, 1] - c[live, 0]).astype(np.float64), [0, 0.5, 1]) / 100.0)))
    # which block shares a CU with block b?
    by = {}
    for b in range(nb):
        if live[b]:
            by.setdefault(int(key[b]), []).append(b)
    print("   first CUs:", [v for _, v in sorted(by.items())][:12])
L.cvx_set_option(b"cf_map", 0); L.cvx_set_option(b"cf_census", 0)
