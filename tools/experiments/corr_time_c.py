"""Times cvx_correlate_ex_f32 (exact / certified-fast with either kernel) on random features of a given geometry.
usage: python tools/experiments/corr_time_c.py C h w d hw [C h w d hw ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from convexadam_amd._lib import lib, ptr, stream_ptr, workspace, CorrOpts, check, CvxError
dev = torch.device('cuda:0')
args = [int(a) for a in sys.argv[1:]]
for i in range(0, len(args), 5):
    Cn, h, w, d, hw = args[i:i + 5]
    n = 2 * hw + 1
    g = torch.Generator(device='cpu').manual_seed(1)
    ff = torch.rand((Cn, h, w, d), generator=g).to(dev); mm = torch.rand((Cn, h, w, d), generator=g).to(dev)
    ssd = torch.empty((n ** 3, h, w, d), dtype=torch.float32, device=dev)
    nws = lib().cvx_correlate_workspace_bytes(Cn, h, w, d, hw)
    ws = workspace(nws, dev)
    alg = (n ** 3 * h * w * d * 4 + 2 * Cn * h * w * d * 4)
    def run(fast, reps=10):
        opts = CorrOpts(0, 2, fast, 0)
        for _ in range(2):
            check(lib().cvx_correlate_ex_f32(ptr(ff), ptr(mm), Cn, h, w, d, hw, C.byref(opts), ptr(ssd), None, ptr(ws), nws, stream_ptr(dev)))
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            check(lib().cvx_correlate_ex_f32(ptr(ff), ptr(mm), Cn, h, w, d, hw, C.byref(opts), ptr(ssd), None, ptr(ws), nws, stream_ptr(dev)))
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    out = ['C=%d %dx%dx%d hw=%d (%.0f MB):' % (Cn, h, w, d, hw, alg / 1e6)]
    t = run(0); out.append('exact %.0f us (%.3f)' % (t, alg / t / 8e6))
    lib().cvx_set_option(b'corr_fused_all', 1)
    t = run(0); out.append('exact fused-all %.0f us (%.3f)' % (t, alg / t / 8e6))
    lib().cvx_set_option(b'corr_fused_all', 0)
    for k in (1, 2):
        lib().cvx_set_option(b'corr_cert', k)
        try:
            t = run(2); out.append('cert kernel %d: %.0f us (%.3f)' % (k, t, alg / t / 8e6))
        except CvxError as e:
            out.append('cert kernel %d: unsupported' % k)
    lib().cvx_set_option(b'corr_cert', 1)
    print('  '.join(out), flush=True)
