#!/usr/bin/env python
"""Distribution of the squared surface distances that k_surface_dist_hist has to find, by misregistration (how much work lands in which stage)."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd.phantom import warped_label_pair   # noqa: E402
from convexadam_amd._lib import lib, ptr, stream_ptr, check   # noqa: E402
dev = torch.device("cuda", 0); L = lib(); H, W, D = 160, 192, 224; nl = 17
nbins = (H - 1) ** 2 + (W - 1) ** 2 + (D - 1) ** 2 + 2
for amp in [float(a) for a in sys.argv[1:]] or [0.005, 0.02, 0.05, 0.1]:
    fx, mv = warped_label_pair((H, W, D), 18, 11, amp); fx, mv = fx.to(dev), mv.to(dev)
    bits = torch.empty(int(L.cvx_label_bits_bytes(H, W, D, nl)) // 8, dtype=torch.int64, device=dev)
    check(L.cvx_label_bits_u64(ptr(fx), H, W, D, nl, ptr(bits), stream_ptr(dev)))
    hist = torch.zeros((nl, nbins), dtype=torch.int64, device=dev); flag = torch.zeros(nl, dtype=torch.int32, device=dev)
    act4 = (C.c_uint64 * 4)(*[(1 << 18) - 2, 0, 0, 0])
    check(L.cvx_surface_distance_hist_i64(ptr(mv), ptr(bits), H, W, D, nl, C.cast(act4, C.c_void_p), nbins, ptr(hist), nbins, ptr(flag), 1, 0, stream_ptr(dev)))
    h = hist.sum(0).cpu().numpy(); n = h.sum()
    # the bit-plane path: sizes of its two work lists
    bits_m = torch.empty_like(bits)
    check(L.cvx_label_bits_u64(ptr(mv), H, W, D, nl, ptr(bits_m), stream_ptr(dev)))
    nws = int(L.cvx_surface_distance_hist_bits_workspace_bytes(H, W, D, nl))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    hist2 = torch.zeros_like(hist); flag2 = torch.zeros_like(flag)
    check(L.cvx_surface_distance_hist_bits_i64(ptr(bits_m), ptr(bits), H, W, D, nl, C.cast(act4, C.c_void_p), nbins, ptr(hist2), nbins, ptr(flag2), 1, 0, ptr(ws), nws, stream_ptr(dev)))
    torch.cuda.synchronize()
    off = (-ws.data_ptr()) % 256
    cnt = ws[off:off + 4096].view(torch.int32).cpu().numpy()
    print("   bit-plane path: %d surface words, %d with bits left after level 3, %d after level 8, %d far voxels, histograms equal: %s" %
          (cnt[:256].sum(), cnt[256:512].sum(), cnt[512:768].sum(), cnt[768:1024].sum(), bool(torch.equal(hist, hist2))))
    c = np.cumsum(h)
    print("amp %.3f: %d surface voxels (%.1f %% of the volume); d2<=3 %.1f %%, <=8 %.1f %%, <=24 %.1f %%, <=63 %.1f %%, <=120 %.1f %%, max d2 %d" %
          (amp, n, 100.0 * n / (H * W * D), 100 * c[3] / n, 100 * c[8] / n, 100 * c[24] / n, 100 * c[63] / n, 100 * c[120] / n, int(np.nonzero(h)[0].max())))
