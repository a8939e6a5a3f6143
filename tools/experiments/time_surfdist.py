"""Device time of the two surface-distance launches of cupy_hd95 on the sweep's label maps + the distribution of the squared distances."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from convexadam_amd import convexAdam_hyper_util as HU  # noqa: E402
from convexadam_amd import sweep  # noqa: E402
from convexadam_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

shape = (160, 192, 224)
dev = torch.device("cuda:0")
seg_f, seg_m, kf, km, nl = sweep._make_labels(shape, 0, dev)
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
disp = torch.zeros((3,) + shape, device=dev)
disp[0], disp[1], disp[2] = 2.0, -1.0, 3.0
disp += noise * torch.randn_like(disp)
warped = HU.warp_labels_nearest(seg_m, disp[None])
L = lib()
H, W, D = shape
nbins = (H - 1) ** 2 + (W - 1) ** 2 + (D - 1) ** 2 + 2
sp = stream_ptr(dev)
act4 = (C.c_uint64 * 4)(int(os.environ.get("ACT", str((1 << (nl + 1)) - 2)), 0), 0, 0, 0)


def planes(seg):
    b = torch.empty(int(L.cvx_label_bits_bytes(H, W, D, nl)) // 8, dtype=torch.int64, device=dev)
    check(L.cvx_label_bits_u64(ptr(seg), H, W, D, nl, ptr(b), sp))
    return b


for name, b, a in (("moving surface vs fixed planes", warped, seg_f), ("fixed surface vs moving planes", seg_f, warped)):
    bits = planes(a)
    hist = torch.zeros((nl, nbins), dtype=torch.int64, device=dev)
    over = torch.zeros(nl, dtype=torch.int32, device=dev)
    run = lambda: check(L.cvx_surface_distance_hist_i64(ptr(b), ptr(bits), H, W, D, nl, C.cast(act4, C.c_void_p), nbins, ptr(hist), nbins, ptr(over), 1, 0, sp))  # noqa: E731
    run()
    torch.cuda.synchronize()
    h = hist.sum(0).cpu()
    t = time.time()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    print("%s: %.3f ms, %d surface voxels; d2 = 1..8: %s, 9..16: %d, 17..64: %d, > 64: %d" % (
        name, (time.time() - t) * 100, int(h.sum()), h[1:9].tolist(), int(h[9:17].sum()), int(h[17:65].sum()), int(h[65:].sum())), flush=True)
