"""Times the certified-fast correlation kernel (and the exact one) on the benchmark pair's pooled MIND features.
usage: python tools/experiments/cert_time.py [debug masks ...]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from convexadam_amd import _lib
from convexadam_amd._lib import lib, ptr, stream_ptr, workspace, CorrOpts, check
from convexadam_amd.convex_adam_utils import MINDSSC, avg_pool
from convexadam_amd.phantom import deformed_pair
dev = torch.device('cuda:0')
fix, mov = deformed_pair((160, 192, 224), 0, 4.0)
ff = avg_pool(MINDSSC(fix.to(dev)[None, None], 1, 2), 6).contiguous()
mm = avg_pool(MINDSSC(mov.to(dev)[None, None], 1, 2), 6).contiguous()
_, Cn, h, w, d = ff.shape
hw = 6; n = 13
ssd = torch.empty((n ** 3, h, w, d), dtype=torch.float32, device=dev)
nws = lib().cvx_correlate_workspace_bytes(Cn, h, w, d, hw)
ws = workspace(nws, dev)
def run(fast, reps=20):
    opts = CorrOpts(0, 2, fast, 0)
    for _ in range(3):
        check(lib().cvx_correlate_ex_f32(ptr(ff), ptr(mm), Cn, h, w, d, hw, C.byref(opts), ptr(ssd), None, ptr(ws), nws, stream_ptr(dev)))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(lib().cvx_correlate_ex_f32(ptr(ff), ptr(mm), Cn, h, w, d, hw, C.byref(opts), ptr(ssd), None, ptr(ws), nws, stream_ptr(dev)))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
if not os.environ.get('CC_ONLY'):
    print('exact  %.1f us (prep + tail + kernel)' % run(0))
    print('fast1  %.1f us' % run(1))
for mask in ([0] if not os.environ.get('CC_ONLY') else []) + [int(a) for a in sys.argv[1:]]:
    lib().cvx_set_option(b'cc_debug', mask)
    print('cert dbg=%2d  %.1f us (prep + kernel)' % (mask, run(2)))
lib().cvx_set_option(b'cc_debug', 0)

if os.environ.get('CC_CENSUS'):
    import numpy as np
    lib().cvx_set_option(b'cc_debug', 1)
    run(2, reps=1)
    torch.cuda.synchronize()
    # the census sits behind the staging copies at the start of the workspace
    nb = 254
    raw = ws.cpu().numpy().view(np.uint8)
    # find: stage_bytes is unknown here; scan for it via the library's own size helper is not exported -- the census offset is printed by the kernel launcher instead
    off = int(os.environ['CC_CENSUS'])
    cen = raw[off:off + nb * 32].view(np.uint64).reshape(nb, 4)
    t0 = cen[:, 0].min()
    dur = (cen[:, 1] - cen[:, 0]).astype(np.float64) / 100.0
    start = (cen[:, 0] - t0).astype(np.float64) / 100.0
    for cls in sorted(set(zip(cen[:, 2] >> 32, cen[:, 3]))):
        m = ((cen[:, 2] >> 32) == cls[0]) & (cen[:, 3] == cls[1])
        print('tiles %d shifts %d: %3d workgroups, start %.1f..%.1f us, duration %.1f / %.1f / %.1f us (min / mean / max)' % (cls[0], cls[1], m.sum(), start[m].min(), start[m].max(), dur[m].min(), dur[m].mean(), dur[m].max()))
    print('kernel span %.1f us' % ((cen[:, 1].max() - t0) / 100.0))
