"""Times cvx_inverse_consistency_f32 (15 iterations, 26x32x37) with one launch per iteration and with the one-launch kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from convexadam_amd import convex_adam_utils as U
from convexadam_amd._lib import lib
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
a = (0.15 * torch.randn((1, 3, 26, 32, 37), generator=g)).to(dev); b = (0.15 * torch.randn((1, 3, 26, 32, 37), generator=g)).to(dev)
for mode in (0, 1, 2, 0, 1):
    lib().cvx_set_option(b'ic_fused', mode)
    for _ in range(5): U.inverse_consistency(a, b, iter=15)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): U.inverse_consistency(a, b, iter=15)
    e1.record(); torch.cuda.synchronize()
    print('ic_fused=%d: %.1f us per call (15 iterations)' % (mode, e0.elapsed_time(e1) / 50 * 1e3))
