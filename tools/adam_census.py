#!/usr/bin/env python
"""Per-workgroup residency of the three Adam-loop kernels (option census_ptr): when each workgroup started, when its first input
arrived and when it ended, relative to the earliest start of its launch (100 MHz s_memrealtime ticks -> us)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convexadam_amd import convex_adam_utils as U   # noqa: E402
from convexadam_amd import _lib                     # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
h, w, d = 80, 96, 112
F2 = torch.rand(1, 12, h, w, d, generator=g).to(dev)
M2 = torch.rand(1, 12, h, w, d, generator=g).to(dev)
P0 = Fn.interpolate(torch.randn(1, 3, 5, 6, 7, generator=g) * 2.0, size=(h, w, d), mode="trilinear").to(dev)
U.adam_run(F2, M2, P0, 1.25, 10, return_state=True)
torch.cuda.synchronize()
buf = torch.zeros(4 * (8192 + 4096), dtype=torch.int64, device=dev)
L.cvx_set_option(b"census_ptr", buf.data_ptr())
U.adam_run(F2, M2, P0, 1.25, 10, return_state=True)           # every launch overwrites: the last iteration's stamps remain
torch.cuda.synchronize()
L.cvx_set_option(b"census_ptr", 0)
c = buf.cpu().numpy().astype(np.uint64).reshape(-1, 4)
for name, lo, hi in (("forward boxes", 0, 1024), ("adjoint boxes + Adam", 1024, 2048), ("warp + gradient", 2048, 2048 + 4096)):
    r = c[lo:hi]
    r = r[r[:, 0] != 0]
    if not len(r):
        continue
    t0 = r[:, 0].min()
    st = (r[:, 0] - t0).astype(np.float64) / 100.0
    en = (r[:, 2] - t0).astype(np.float64) / 100.0
    line = "%-22s %5d workgroups  start: median %.2f p90 %.2f max %.2f us | end: min %.2f median %.2f max %.2f | life: median %.2f max %.2f" % (
        name, len(r), np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en), en.max(), np.median(en - st), (en - st).max())
    if r[:, 1].any():
        fd = (r[:, 1] - r[:, 0]).astype(np.float64) / 100.0
        line += " | first data after start: median %.2f max %.2f" % (np.median(fd), fd.max())
    print(line)
    xcc = (r[:, 3] >> np.uint64(32)) & np.uint64(0xf)
    print("    workgroups per XCC:", np.bincount(xcc.astype(np.int64), minlength=8).tolist())
    hw = r[:, 3] & np.uint64(0xffffffff)
    cu = ((xcc.astype(np.int64) << 8) | (((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64) << 5) | (((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64) << 4)
          | ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64))
    ids, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
    life = en - st
    print("    distinct CUs used: %d; workgroups per CU histogram:" % len(ids), np.bincount(cnt).tolist())
    for k in sorted(set(cnt.tolist())):
        sel = cnt[inv] == k
        print("      CUs holding %d workgroups: life median %.2f max %.2f us, end median %.2f max %.2f" % (k, np.median(life[sel]), life[sel].max(), np.median(en[sel]), en[sel].max()))
    if "warp" in name:                           # occupancy over time: resident workgroups per microsecond, starts per microsecond
        T = int(np.ceil(en.max())) + 1
        occ = np.zeros(T); sts = np.zeros(T)
        for a, b_ in zip(st, en):
            occ[int(a):int(np.ceil(b_))] += 1
            sts[int(a)] += 1
        print("    resident workgroups at each us (of %d slots at 6 per CU):" % (6 * 256), " ".join("%d" % v for v in occ))
        print("    workgroups started in each us:", " ".join("%d" % v for v in sts))
        order_ = np.argsort(st)
        third = len(st) // 3
        for nm, sel in (("first third by start", order_[:third]), ("middle third", order_[third:2 * third]), ("last third", order_[2 * third:])):
            print("    life of the %s: median %.2f p10 %.2f p90 %.2f" % (nm, np.median(life[sel]), np.percentile(life[sel], 10), np.percentile(life[sel], 90)))
    if "boxes" in name:                          # which tiles are slow?  (launch_qpr: b = (blk & 7) * (grid >> 3) + (blk >> 3); zi fastest, then xi, yi, channel)
        grid = len(r)
        blk = np.arange(grid)
        b = (blk & 7) * (grid >> 3) + (blk >> 3)
        nzc, nxt, nyt = int(os.environ.get('CENSUS_NZC', '7')), 2, 12
        zi, xi, yi, ch = b % nzc, (b // nzc) % nxt, (b // (nzc * nxt)) % nyt, b // (nzc * nxt * nyt)
        full = c[lo:lo + grid]
        lf = ((full[:, 2] - full[:, 0]).astype(np.float64)) / 100.0
        for nm, key, n in (("z chunk", zi, nzc), ("x tile", xi, nxt), ("y tile", yi, nyt), ("channel", ch, 3), ("blk>>8", blk >> 8, 2), ("blk&7", blk & 7, 8), ("(blk>>3)%8", (blk >> 3) % 8, 8)):
            print("      life by %-8s" % nm, " ".join("%.1f" % np.median(lf[key == k]) for k in range(n)), "| max", " ".join("%.1f" % lf[key == k].max() for k in range(n)))
        # partner on the same CU
        cuf = cu
        order = np.argsort(cuf, kind="stable")
        pairs = [(order[i], order[i + 1]) for i in range(len(order) - 1) if cuf[order[i]] == cuf[order[i + 1]]]
        dl = np.array([abs(lf[a] - lf[b_]) for a, b_ in pairs])
        mx = np.array([max(lf[a], lf[b_]) for a, b_ in pairs])
        zz = np.array([(zi[a], zi[b_]) for a, b_ in pairs])
        print("      CU partners: |life difference| median %.2f; pair max life median %.2f; pairs whose slower member > 16 us: %d of %d; their z chunks: %s" % (
            np.median(dl), np.median(mx), int((mx > 16).sum()), len(pairs), zz[mx > 16][:12].tolist()))
