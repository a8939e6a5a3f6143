"""Counts of the certified pipeline on the benchmark pair and the zero-background pair (CVX_CERT_TRACE=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['CVX_CERT_TRACE'] = '1'
import torch
from convexadam_amd.convex_adam_MIND import convex_adam_pt
from convexadam_amd.phantom import deformed_pair, zero_background_pair
kw = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=6, disp_hw=6, selected_niter=2, grid_sp_adam=2, ic=True)
for name, pair in (('benchmark', deformed_pair((160, 192, 224), 0, 4.0)), ('zero background', zero_background_pair((160, 192, 224), 0, 4.0))):
    print(name, flush=True)
    convex_adam_pt(pair[0], pair[1], dtype=torch.float32, device=torch.device('cuda:0'), **kw)
