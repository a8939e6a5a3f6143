"""Mirror of the operator half of the reference's `self_configuring/convexAdam_hyper_util.py` (the module the sweep
scripts import: convex_run_withconfig.py:13-16, adam_run_withconfig_shiftSpline.py:15-19):

    MINDSSC(img, radius, dilation)                 hyper_util.py:161-205   (no device argument there)
    correlate / coupled_convex / inverse_consistency hyper_util.py:209-282 (same math as convex_adam_utils)
    extract_features(...), extract_features_nnunet(pred_fixed, pred_moving, mult=10)   :109-146, :64-83
    GaussianSmoothing(sigma), kovesi_spline(sigma, n)                                  :454-488

The two smoothers are callables on (1,C,H,W,D) device tensors AND differentiable (torch.autograd.Function whose
backward is the adjoint HIP kernel), so the sweep scripts' inline autograd Adam loop
(adam_run_withconfig_shiftSpline.py:214-230) can keep calling `avgs[avg_n](net[0].weight)`; they also serve as the
`smoother=` argument of convexadam_amd.convex_adam_utils.adam_run (fused loop).  Evaluation metrics of that file
(dice_coeff, cupy_hd95, jacobian_determinant_3d, sort_rank) are out of scope.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import Smoother, check, f32c, lib, ptr, require_device_tensor, stream_ptr, workspace
from .convex_adam_utils import coupled_convex, correlate, inverse_consistency  # noqa: F401  (same operators)
from .convex_adam_utils import MINDSSC as _MINDSSC
from .convex_adam_MIND import extract_features as _extract_features
from .convex_adam_nnUNet import extract_features as _extract_features_nnunet


def MINDSSC(img, radius=2, dilation=2):
    return _MINDSSC(img, radius, dilation, device=img.device if img.device.type == "cuda" else "cuda")


def extract_features(img_fixed, img_moving, mind_r, mind_d, use_mask, mask_fixed, mask_moving):
    """hyper_util.py:109-146: CUDA + half precision in the reference; float32 on the HIP device here."""
    return _extract_features(img_fixed, img_moving, mind_r, mind_d, use_mask, mask_fixed, mask_moving,
                             device=torch.device("cuda"), dtype=torch.float32)


def extract_features_nnunet(pred_fixed, pred_moving, mult=10):
    return _extract_features_nnunet(pred_fixed, pred_moving, mult=float(mult))


def _apply(x, spec, backward):
    x = require_device_tensor(x, "x")
    _, Cn, H, W, D = [int(s) for s in x.shape]
    a = f32c(x)
    out = torch.empty_like(a)
    nws = lib().cvx_smooth_workspace_bytes(Cn, H, W, D)
    ws = workspace(nws, a.device)
    with torch.cuda.device(a.device):
        check(lib().cvx_smooth_f32(ptr(a), Cn, H, W, D, C.byref(spec), 1 if backward else 0, ptr(out), ptr(ws), nws, stream_ptr(a.device)))
    return out


class _SmoothFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return _apply(x, holder.spec, False).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return _apply(g.contiguous(), ctx.holder.spec, True).to(g.dtype), None


class _SmootherBase(torch.nn.Module):
    spec: Smoother

    def forward(self, x):
        return _SmoothFn.apply(x, self)


class GaussianSmoothing(_SmootherBase):
    """Separable 5-tap Gaussian, replicate padding (hyper_util.py:454-473); the taps are computed with the same torch
    ops as the reference so that they are the same float32 numbers."""

    def __init__(self, sigma):
        super().__init__()
        sigma = torch.tensor([sigma])
        N = torch.ceil(sigma * 3.0 / 2.0).long().item() * 2 + 1
        weight = torch.exp(-torch.pow(torch.linspace(-(N // 2), N // 2, N), 2) / (2 * torch.pow(sigma, 2)))
        weight /= weight.sum()
        if N != 5:
            raise ValueError("GaussianSmoothing: only the 5-tap kernels of the sweep (sigma <= 1.33) are built, got N=%d" % N)
        self.weight = weight
        self.spec = Smoother()
        self.spec.kind = 1
        for i in range(5):
            self.spec.gauss_w[i] = float(weight[i])


class _BoxChain(_SmootherBase):
    def __init__(self, sizes):
        super().__init__()
        self.sizes = list(sizes)
        self.spec = Smoother()
        self.spec.kind = 0
        self.spec.n_boxes = len(self.sizes)
        for i, k in enumerate(self.sizes):
            self.spec.box_k[i] = int(k)


def kovesi_spline(sigma, n=4):
    """Chain of zero-padded box filters approximating a Gaussian of width sigma (hyper_util.py:475-488):
    1.3 -> [3,3,3], 1.6 -> [3,3,3,3], 1.9 -> [3,3,3,5], 2.2 -> [3,3,5,5], 2.5 -> [3,5,5,5], 2.8 -> [5,5,5,5]."""
    w_ideal = np.sqrt(12 * sigma ** 2 / n + 1)
    w_u = int(np.ceil((w_ideal - 1) / 2) * 2 + 1)
    w_l = max(w_u - 2, 1)
    m = int(np.round((12 * sigma ** 2 - n * w_l ** 2 - 4 * n * w_l - 3 * n) / (-4 * w_l - 4)))
    sizes = [w_l] * m if w_l > 1 else []
    sizes += [w_u] * (n - m)
    if not 1 <= len(sizes) <= 4:
        raise ValueError("kovesi_spline: %d boxes not supported" % len(sizes))
    return _BoxChain(sizes)
