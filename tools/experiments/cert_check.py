import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from convexadam_amd.convex_adam_utils import correlate
dev = torch.device('cuda:0')
def run(C, h, w, d, hw, seed=0, kind='rand'):
    g = torch.Generator().manual_seed(seed)
    f = torch.rand(1, C, h, w, d, generator=g)
    m = torch.rand(1, C, h, w, d, generator=g)
    if kind == 'shift':
        m = torch.roll(f, (1, -1, 2), (2, 3, 4)) + 0.01 * m
    if kind == 'zero':
        f[:, :, :, : w // 2] = 0; m[:, :, :, : w // 2] = 0
    f, m = f.to(dev), m.to(dev)
    ssd, am = correlate(f, m, hw, 1, (h, w, d), ch=C, mode='exact')
    ssdu, am2 = correlate(f, m, hw, 1, (h, w, d), ch=C, mode='certified')
    torch.cuda.synchronize()
    a = ssd.double().cpu().numpy(); b = ssdu.double().cpu().numpy() / 729.0
    den = np.maximum(a, 1e-30)
    rel = np.abs(b - a) / den
    rel[(a == 0) & (b == 0)] = 0
    zero_mismatch = int(((a == 0) != (b == 0)).sum())
    neq = int((am != am2).sum())
    print('C%d %dx%dx%d hw%d %s: max rel %.3e zero-mismatch %d argmin-mismatch %d / %d' % (C, h, w, d, hw, kind, rel.max(), zero_mismatch, neq, am.numel()), flush=True)
    return rel.max(), neq, zero_mismatch
bad = 0
for args in [(12, 6, 8, 9, 2), (12, 26, 32, 37, 6), (12, 5, 32, 13, 3), (12, 7, 16, 21, 4), (5, 4, 7, 10, 1), (12, 9, 11, 6, 2), (3, 3, 3, 3, 1), (12, 4, 64, 5, 2), (12, 8, 32, 38, 6), (12, 8, 32, 40, 6),(12, 6, 20, 39, 5), (1, 2, 2, 2, 0), (12, 10, 12, 14, 8)]:
    for kind in ('rand', 'shift', 'zero'):
        try:
            r, n, z = run(*args, kind=kind)
            if r > 1.5e-5 or n or z: bad += 1
        except Exception as e:
            print(args, kind, 'ERROR', e); bad += 1
print('bad', bad)
if len(sys.argv) > 1:
    # timing
    C, h, w, d, hw = 12, 26, 32, 37, 6
    f = torch.rand(1, C, h, w, d).to(dev); m = torch.rand(1, C, h, w, d).to(dev)
    for mode in ('exact', 'certified'):
        for _ in range(3): correlate(f, m, hw, 1, (h, w, d), ch=C, mode=mode)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(20): correlate(f, m, hw, 1, (h, w, d), ch=C, mode=mode)
        torch.cuda.synchronize(); print(mode, (time.time() - t) / 20 * 1e3, 'ms per call (incl. prep + argmin)')
