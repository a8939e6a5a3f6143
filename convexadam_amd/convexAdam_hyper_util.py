"""Mirror of the operator half of the reference's `self_configuring/convexAdam_hyper_util.py` (the module the sweep
scripts import: convex_run_withconfig.py:13-16, adam_run_withconfig_shiftSpline.py:15-19):

    MINDSSC(img, radius, dilation)                 hyper_util.py:161-205   (no device argument there)
    correlate / coupled_convex / inverse_consistency hyper_util.py:209-282 (same math as convex_adam_utils)
    extract_features(...), extract_features_nnunet(pred_fixed, pred_moving, mult=10)   :109-146, :64-83
    GaussianSmoothing(sigma), kovesi_spline(sigma, n)                                  :454-488

The two smoothers are callables on (1,C,H,W,D) device tensors AND differentiable (torch.autograd.Function whose
backward is the adjoint HIP kernel), so the sweep scripts' inline autograd Adam loop
(adam_run_withconfig_shiftSpline.py:214-230) can keep calling `avgs[avg_n](net[0].weight)`; they also serve as the
`smoother=` argument of convexadam_amd.convex_adam_utils.adam_run (fused loop).

Evaluation operators of that file, on the device:
    jacobian_determinant_3d(dense_flow, convert1=True)   :86-108
    dice_coeff(outputs, labels, max_label)               :53-60
    sort_rank(value)                                     :28-31   (host)
plus the three call sequences the sweep scripts write inline (convex_run_withconfig.py:141,148-150,
convex_run_paired_mind.py:167-173): warp_labels_nearest, jacobian_log_std_and_folding, tre_at_keypoints.
cupy_hd95(fixed, moving, num_labels, precision=1)    :32-51   (device feature transforms + histogram percentile)
"""
import ctypes as C
import os

import numpy as np
import torch

from ._lib import Smoother, check, f32c, lib, ptr, require_device_tensor, stream_ptr, workspace
from .convex_adam_utils import coupled_convex, correlate, gpu_usage, inverse_consistency  # noqa: F401  (same operators)
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


# ---- evaluation operators -----------------------------------------------------------------------------
def jacobian_determinant_3d(dense_flow, convert1=True):
    """hyper_util.py:86-108.  dense_flow (1,3,H,W,D) device tensor -> (H-4, W-4, D-4) float32 device tensor."""
    x = f32c(require_device_tensor(dense_flow, "dense_flow"))
    B, ch, H, W, D = [int(v) for v in x.shape]
    if B != 1 or ch != 3:
        raise ValueError("jacobian_determinant_3d: expected a (1,3,H,W,D) field")
    out = torch.empty((max(H - 4, 0), max(W - 4, 0), max(D - 4, 0)), dtype=torch.float32, device=x.device)
    if out.numel() == 0:                         # an extent of at most 4 voxels: the slice [2:-2] of the reference (:103) is empty
        return out
    with torch.cuda.device(x.device):
        check(lib().cvx_jacobian_det_f32(ptr(x), H, W, D, 1 if convert1 else 0, ptr(out), stream_ptr(x.device)))
    return out


def jacobian_log_std_and_folding(jac_det):
    """convex_run_withconfig.py:148-150: (jac_det.add(3).clamp_(1e-9, 1e9).log().std(), (jac_det < 0).float().mean()) as
    Python floats; accumulated in float64 on the device."""
    j = f32c(require_device_tensor(jac_det, "jac_det")).reshape(-1)
    n = int(j.numel())
    acc = torch.empty(3, dtype=torch.float64, device=j.device)
    with torch.cuda.device(j.device):
        check(lib().cvx_jacobian_stats_f64(ptr(j), n, ptr(acc), stream_ptr(j.device)))
    s, s2, neg = [float(v) for v in acc.cpu()]
    var = max(s2 - s * s / n, 0.0) / max(n - 1, 1)
    return var ** 0.5, neg / n


def warp_labels_nearest(seg_moving, disp_hr):
    """convex_run_withconfig.py:96,135,141: F.grid_sample(seg.view(1,1,H,W,D), grid0 + disp_hr.permute(0,2,3,4,1).flip(-1)
    .div(scale1), mode='nearest').squeeze().  seg (H,W,D), disp_hr (1,3,H,W,D) in voxels -> (H,W,D) float32."""
    from .convex_adam_utils import _base_tables
    seg = f32c(require_device_tensor(seg_moving, "seg_moving"))
    d = f32c(require_device_tensor(disp_hr, "disp_hr"))
    H, W, D = [int(v) for v in seg.shape[-3:]]
    if tuple(d.shape[-4:]) != (3, H, W, D):
        raise ValueError("warp_labels_nearest: disp_hr must be (1,3,H,W,D) matching the label map")
    bh, bw, bd = _base_tables(H, W, D, seg.device)
    out = torch.empty((H, W, D), dtype=torch.float32, device=seg.device)
    with torch.cuda.device(seg.device):
        check(lib().cvx_warp_labels_nearest_f32(ptr(seg), ptr(d), H, W, D, ptr(bh), ptr(bw), ptr(bd), ptr(out), stream_ptr(seg.device)))
    return out


def label_overlap_counts(a, b, max_label):
    """(3, max_label) int64 on the host: voxels of a, of b and of both carrying each integer value 0 .. max_label-1 -- what dice_coeff and
    cupy_hd95 both start from (one kernel, one host synchronisation; pass it to both as `counts` to share it)."""
    a = f32c(require_device_tensor(a, "outputs")).reshape(-1)
    b = f32c(require_device_tensor(b, "labels")).reshape(-1)
    if a.numel() != b.numel():
        raise ValueError("label maps differ in size")
    counts = torch.empty((3, int(max_label)), dtype=torch.int64, device=a.device)
    with torch.cuda.device(a.device):
        check(lib().cvx_label_overlap_i64(ptr(a), ptr(b), int(a.numel()), int(max_label), ptr(counts), stream_ptr(a.device)))
    return counts.cpu().numpy()


def dice_coeff(outputs, labels, max_label, counts=None):
    """hyper_util.py:53-60: per-label Dice for labels 1 .. max_label-1 (FloatTensor on the host, like the reference).
    The device returns exact voxel counts; the float32 means and the quotient follow the reference's expression.
    counts (not in the reference): label_overlap_counts(outputs, labels, max_label) if the caller already has it."""
    n = int(outputs.numel())
    if int(labels.numel()) != n:
        raise ValueError("dice_coeff: label maps differ in size")
    c = label_overlap_counts(outputs, labels, max_label) if counts is None else counts
    if c.shape != (3, int(max_label)):
        raise ValueError("dice_coeff: counts must be label_overlap_counts(outputs, labels, max_label)")
    nf = np.float32(n)
    dice = np.zeros(int(max_label) - 1, np.float32)
    for lab in range(1, int(max_label)):
        inter = np.float32(c[2, lab]) / nf
        dice[lab - 1] = (np.float32(2.0) * inter) / ((np.float32(1e-8) + np.float32(c[0, lab]) / nf) + np.float32(c[1, lab]) / nf)
    return torch.from_numpy(dice)


def tre_at_keypoints(disp_hr, key_fixed, key_moving):
    """convex_run_paired_mind.py:165-173: disp_sampled = grid_sample(disp_hr, key_fixed.flip(1)/scale1 - 1) (trilinear),
    TRE = |key_fixed - key_moving + disp_sampled|.  Key points (n,3) in voxels (H,W,D order); returns (TRE (n,), disp_sampled (n,3))
    on the host like the reference."""
    from .convex_adam_utils import grid_sample
    d = f32c(require_device_tensor(disp_hr, "disp_hr"))
    H, W, D = [int(v) for v in d.shape[-3:]]
    kf = torch.as_tensor(key_fixed, dtype=torch.float32).cpu()
    km = torch.as_tensor(key_moving, dtype=torch.float32).cpu()
    scale1 = torch.tensor([D - 1, W - 1, H - 1], dtype=torch.float32) / 2
    lms = (kf.flip(1) / scale1 - 1).view(1, -1, 1, 1, 3).to(d.device)
    samp = grid_sample(d.view(1, 3, H, W, D), lms).reshape(3, -1).t().cpu()
    return (kf - km + samp).square().sum(-1).sqrt(), samp


def sort_rank(value):
    """hyper_util.py:28-31 (host)."""
    value = torch.as_tensor(value)
    rank1 = torch.ones_like(value)
    rank1[value.sort().indices] = torch.linspace(1, .1, len(value)).to(value.device)
    return rank1


def percentile_neighbours(n, q):
    """numpy.percentile (method 'linear') on n float32 samples: indices of the two order statistics it interpolates and the
    weight, in numpy's own arithmetic -- for a float32 array the quantile q/100 and the virtual index (n-1)*q/100 are float32
    (numpy/lib/_function_base_impl.py: percentile, _get_indexes, _get_gamma)."""
    quant = np.true_divide(q, np.float32(100))
    virt = np.asanyarray((n - 1) * quant)
    if virt >= n - 1:
        return n - 1, n - 1, np.float32(virt - (-1))
    k0 = int(np.floor(virt))
    return k0, k0 + 1, np.float32(np.float64(virt) - k0)


def percentile_linear_from_sorted_pair(a, b, gamma):
    """numpy's _lerp on float32 operands: a + (b-a)*gamma, or b - (b-a)*(1-gamma) for gamma >= 0.5."""
    a32, b32, t = np.float32(a), np.float32(b), np.float32(gamma)
    diff = np.subtract(b32, a32)
    if t >= 0.5:
        return np.subtract(b32, diff * (np.float32(1) - t))
    return np.add(a32, diff * t)


def edt_squared(obj):
    """Exact squared Euclidean distance of every voxel of a (H,W,D) -- or a batch (B,H,W,D) -- device tensor to the nearest ZERO
    voxel of its volume (0 on zero voxels), int32: round(scipy.ndimage.distance_transform_edt(obj)**2) (csrc/edt.hip)."""
    o = f32c(require_device_tensor(obj, "obj"))
    if o.dim() not in (3, 4):
        raise ValueError("edt_squared: (H,W,D) or (B,H,W,D) tensor expected")
    B = int(o.shape[0]) if o.dim() == 4 else 1
    H, W, D = [int(v) for v in o.shape[-3:]]
    L = lib()
    out = torch.empty(tuple(o.shape), dtype=torch.int32, device=o.device)
    nws = L.cvx_edt_squared_workspace_bytes(B, H, W, D)
    ws = workspace(nws, o.device)
    with torch.cuda.device(o.device):
        check(L.cvx_edt_squared_i32(ptr(o), B, H, W, D, ptr(out), ptr(ws), nws, stream_ptr(o.device)))
    return out


# cupy_hd95(method="surface"): rows searched around a surface voxel before the call falls back to the distance transforms (the cost of a
# voxel grows with the square of its distance to the other surface; the transforms' cost does not depend on it)
HD95_SURFACE_MAX_RADIUS = 48
HD95_SURFACE_KERNEL = os.environ.get("CONVEXADAM_HD95_KERNEL", "bits")      # "bits" | "voxels" (csrc/surfdist.hip)
if HD95_SURFACE_KERNEL not in ("bits", "voxels"):
    raise ValueError("CONVEXADAM_HD95_KERNEL must be 'bits' or 'voxels', got %r" % (HD95_SURFACE_KERNEL,))


def cupy_hd95(fixed, moving, num_labels, precision=1, fixed_cache=None, method=None, counts=None):
    """hyper_util.py:32-51: 95th-percentile symmetric surface distance for labels 1 .. num_labels (30 where a label is absent from
    either map), float64 tensor on the device of `fixed`.  Per label on the device: masks on the nearest-upsampled grid, exact
    squared Euclidean distance transforms of the mask and of its complement (csrc/edt.hip), histogram of dist_a over the
    surface of b (edt == 1); the percentile is read from the histogram (two order statistics), so no sort and no host copy of a
    volume.  Two host synchronisations per call (label counts, results).  `precision` is any positive scale factor of
    F.interpolate's nearest mode (:33-34; the reference's call sites use the default 1).
    method (not in the reference): "surface" (default at precision 1, <= 255 labels, H, W <= 2047) computes the distances at the surface
    voxels alone -- bit planes of both maps + a ring search per surface voxel (csrc/surfdist.hip), no volume-sized transform; "edt" is the
    path described above (the only one for precision > 1).  Both give the same float64 results bit for bit (exact integer squared
    distances either way); "surface" needs two host synchronisations (label counts, results) and hands the call over to "edt" when
    a surface voxel lies more than HD95_SURFACE_MAX_RADIUS rows from the other map's label (badly registered pairs).
    counts (not in the reference): label_overlap_counts(fixed, moving, num_labels + 1) if the caller already has it (the sweep shares it
    with dice_coeff: one kernel and one synchronisation less per evaluation).
    fixed_cache (not in the reference): a dict the caller keeps per FIXED label map -- the sweep scores many fields against the same
    fixed segmentation (16 per Adam run), and what is derived from the fixed map alone does not depend on the field: its bit planes
    ("surface": 1 MB per label at 160x192x224) or the two distance transforms of every fixed label ("edt": 715 MB for 13 labels);
    computed on the first call and reused."""
    prec = float(precision)
    if not (prec > 0.0) or prec == float("inf"):
        raise ValueError("cupy_hd95: precision must be a positive number")
    p = int(prec) if prec.is_integer() else None                # None: a non-integer scale factor of F.interpolate (:33-34)
    fx = f32c(require_device_tensor(fixed, "fixed"))
    mv = f32c(require_device_tensor(moving, "moving"))
    if fx.shape != mv.shape or fx.dim() != 3:
        raise ValueError("cupy_hd95: label maps must be (H, W, D) tensors of the same shape")
    H, W, D = [int(v) for v in fx.shape]
    # upsample_nearest3d with scale_factor: extent (int64)(n * s) per axis, source index by ATen's nearest_idx (csrc/edt.hip::nearest_src)
    Ho, Wo, Do = (H * p, W * p, D * p) if p else (int(H * prec), int(W * prec), int(D * prec))
    if min(Ho, Wo, Do) < 1:
        raise RuntimeError("cupy_hd95: Input and output sizes should be greater than 0 (%dx%dx%d at precision %g)" % (H, W, D, prec))
    nbins = (Ho - 1) ** 2 + (Wo - 1) ** 2 + (Do - 1) ** 2 + 2
    nl = int(num_labels)
    L = lib()
    dev = fx.device
    sp = stream_ptr(dev)
    n = Ho * Wo * Do

    def transforms(seg, labs):
        """[len(labs)][2][Ho][Wo][Do] int32: squared distance inside / outside every listed label of `seg`, in groups that bound the scratch."""
        out = torch.empty((len(labs), 2, Ho, Wo, Do), dtype=torch.int32, device=dev)
        if p == 1:                              # masks read straight from the label map (no materialised mask volumes)
            for g0 in range(0, len(labs), 64):
                part = labs[g0:g0 + 64]
                arr = (C.c_int * len(part))(*part)
                nws = L.cvx_edt_squared_workspace_bytes(2 * len(part), Ho, Wo, Do)
                ws = workspace(nws, dev)
                check(L.cvx_edt_squared_labels_i32(ptr(seg), H, W, D, C.cast(arr, C.c_void_p), len(part), ptr(out[g0:g0 + len(part)]), ptr(ws), nws, sp))
            return out
        group = max(1, min(len(labs), int((4 << 30) // (40 * n))))
        obj = torch.empty((group, 2, Ho, Wo, Do), dtype=torch.float32, device=dev)
        nws = L.cvx_edt_squared_workspace_bytes(2 * group, Ho, Wo, Do)
        ws = workspace(nws, dev)
        for g0 in range(0, len(labs), group):
            part = labs[g0:g0 + group]
            for i, lab in enumerate(part):
                if p and p <= 8:
                    check(L.cvx_label_mask_f32(ptr(seg), H, W, D, lab, p, ptr(obj[i, 0]), ptr(obj[i, 1]), None, sp))
                else:
                    check(L.cvx_label_mask_scaled_f32(ptr(seg), H, W, D, lab, Ho, Wo, Do, float(np.float32(1.0 / prec)), ptr(obj[i, 0]), ptr(obj[i, 1]), None, sp))
            check(L.cvx_edt_squared_i32(ptr(obj), 2 * len(part), Ho, Wo, Do, ptr(out[g0:g0 + len(part)]), ptr(ws), nws, sp))
        return out

    if method is None:
        method = "surface" if (p == 1 and nl <= 255 and H <= 2047 and W <= 2047) else "edt"
    if method not in ("surface", "edt"):
        raise ValueError("cupy_hd95: method must be 'surface' or 'edt'")
    if method == "surface":
        if not (p == 1 and 1 <= nl <= 255 and H <= 2047 and W <= 2047):
            raise NotImplementedError("cupy_hd95: method 'surface' needs precision 1, 1 .. 255 labels and H, W <= 2047")
        with torch.cuda.device(dev):
            # label range (F.one_hot, :33) and presence in one pass: voxel counts of the integer values 0 .. num_labels in both maps
            cnt = label_overlap_counts(fx, mv, nl + 1) if counts is None else counts
            if cnt.shape != (3, nl + 1):
                raise ValueError("cupy_hd95: counts must be label_overlap_counts(fixed, moving, num_labels + 1)")
            if int(cnt[0].sum()) != n or int(cnt[1].sum()) != n:
                raise RuntimeError("cupy_hd95: class values must be in 0 .. num_labels (F.one_hot, :33)")
            present = [i for i in range(1, nl + 1) if cnt[0, i] > 0 and cnt[1, i] > 0]
            res = None
            if present:
                nwords = int(L.cvx_label_bits_bytes(H, W, D, nl)) // 8

                def planes(seg):
                    b = torch.empty(nwords, dtype=torch.int64, device=dev)
                    check(L.cvx_label_bits_u64(ptr(seg), H, W, D, nl, ptr(b), sp))
                    return b

                if fixed_cache is not None:
                    key = ("bits", nl)
                    if key not in fixed_cache:
                        fixed_cache[key] = planes(fx)
                    bits_f = fixed_cache[key]
                else:
                    bits_f = planes(fx)
                bits_m = planes(mv)
                hist = torch.zeros((nl, 2, nbins), dtype=torch.int64, device=dev)
                tail = torch.zeros(nl * 2 * 3 + nl, dtype=torch.int64, device=dev)       # out3 [nl][2][3] | overflow [nl][2] int32
                out3, flag = tail[:nl * 6], tail[nl * 6:].view(torch.int32)
                act = [0, 0, 0, 0]
                for lab in present:
                    act[lab >> 6] |= 1 << (lab & 63)
                act4 = (C.c_uint64 * 4)(*act)
                # k = 0: dist1[surf2] (surface of the moving map against the fixed planes), k = 1: dist2[surf1] (:48)
                # (HD95_SURFACE_KERNEL = "bits", the default: both maps as bit planes, 64 voxels per thread; "voxels": one lane per voxel of
                # the label map -- the round-4 kernel, same counts)
                use_bits = HD95_SURFACE_KERNEL == "bits" and nl * H * W * ((D + 63) // 64) < (1 << 32)       # (the word lists index 32 bits)
                nws = int(L.cvx_surface_distance_hist_bits_workspace_bytes(H, W, D, nl)) if use_bits else 0
                ws = workspace(nws, dev) if nws else None
                for k, (seg_b, bits_b, bits_a) in enumerate(((mv, bits_m, bits_f), (fx, bits_f, bits_m))):
                    if use_bits:
                        check(L.cvx_surface_distance_hist_bits_i64(ptr(bits_b), ptr(bits_a), H, W, D, nl, C.cast(act4, C.c_void_p), nbins,
                                                                   C.c_void_p(hist.data_ptr() + 8 * k * nbins), 2 * nbins,
                                                                   C.c_void_p(flag.data_ptr() + 4 * k), 2, int(HD95_SURFACE_MAX_RADIUS), ptr(ws), nws, sp))
                    else:
                        check(L.cvx_surface_distance_hist_i64(ptr(seg_b), ptr(bits_a), H, W, D, nl, C.cast(act4, C.c_void_p), nbins,
                                                              C.c_void_p(hist.data_ptr() + 8 * k * nbins), 2 * nbins,
                                                              C.c_void_p(flag.data_ptr() + 4 * k), 2, int(HD95_SURFACE_MAX_RADIUS), sp))
                quant = float(np.true_divide(95, np.float32(100)))               # numpy: q / float32(100) for float32 data
                check(L.cvx_hist_percentile_neighbours_batch_i64(ptr(hist), nbins, 2 * nl, quant, ptr(out3), sp))
                host = tail.cpu().numpy()
                flags = host[nl * 6:].view(np.int32)[[2 * (lab - 1) + k for lab in present for k in range(2)]]
                if np.any(flags == 2):               # a surface voxel farther than HD95_SURFACE_MAX_RADIUS rows from its target: the ring
                    # the search would crawl: the transforms take over.  The fixed map's transforms are NOT kept in the caller's cache here
                    # (715 MB per pair at 160x192x224, for a case that should be rare)
                    return cupy_hd95(fixed, moving, num_labels, precision, None, method="edt", counts=cnt)
                if np.any(flags != 0):
                    raise RuntimeError("cupy_hd95: squared distance exceeds the histogram range")
                res = host[:nl * 6].reshape(nl, 2, 3)[[lab - 1 for lab in present]]
        return _hd95_from_order_stats(res, present, nl, precision, dev)
    with torch.cuda.device(dev):
        # label range (F.one_hot, :33) and presence in one pass: voxel counts of labels 0 .. num_labels in both maps
        cnt = label_overlap_counts(fx, mv, nl + 1) if counts is None else counts
        if cnt.shape != (3, nl + 1):
            raise ValueError("cupy_hd95: counts must be label_overlap_counts(fixed, moving, num_labels + 1)")
        if int(cnt[0].sum()) != int(fx.numel()) or int(cnt[1].sum()) != int(fx.numel()):
            raise RuntimeError("cupy_hd95: class values must be in 0 .. num_labels (F.one_hot, :33)")
        present = [i for i in range(1, nl + 1) if cnt[0, i] > 0 and cnt[1, i] > 0]
        if present:
            if fixed_cache is not None:
                key = ("edt", prec, nl)
                if key not in fixed_cache:
                    labs_f = [i for i in range(1, nl + 1) if cnt[0, i] > 0]
                    fixed_cache[key] = (labs_f, transforms(fx, labs_f))
                labs_f, dist_f_all = fixed_cache[key]
                dist_f = [dist_f_all[labs_f.index(lab)] for lab in present]
            else:
                dist_f_all = transforms(fx, present)
                dist_f = [dist_f_all[j] for j in range(len(present))]
            dist_m = transforms(mv, present)
            hist = torch.empty((2 * len(present), nbins), dtype=torch.int64, device=dev)
            flag = torch.zeros(2 * len(present), dtype=torch.int32, device=dev)
            out3 = torch.empty((len(present), 2, 3), dtype=torch.int64, device=dev)
            quant = float(np.true_divide(95, np.float32(100)))               # numpy: q / float32(100) for float32 data
            # dist1[surf2] and dist2[surf1] (:48) for every label: one launch over a device table of volume addresses
            tab = []
            for j, lab in enumerate(present):
                pair = (dist_f[j], dist_m[j])
                for k in range(2):
                    tab += [pair[k][0].data_ptr(), pair[k][1].data_ptr(), pair[1 - k][0].data_ptr()]
            tab_d = torch.tensor(tab, dtype=torch.int64).to(dev)
            check(L.cvx_surface_hist_batch_i64(ptr(tab_d), 2 * len(present), n, nbins, ptr(hist), ptr(flag), sp))
            check(L.cvx_hist_percentile_neighbours_batch_i64(ptr(hist), nbins, 2 * len(present), quant, ptr(out3), sp))
            res = out3.cpu().numpy()
            if int(flag.cpu().max()) != 0:
                raise RuntimeError("cupy_hd95: squared distance exceeds the histogram range")
        if not present:
            res = None
    return _hd95_from_order_stats(res, present, nl, precision, dev)


def _hd95_from_order_stats(res, present, nl, precision, dev):
    """res [len(present)][2][3] = (lower neighbour, upper neighbour, count) of the squared surface distances per direction -> the
    reference's per-label values (hyper_util.py:48-51): 30 for a label absent from either map."""
    hd95 = np.full(nl, 30, np.float64)
    for j, lab in enumerate(present):
        pk = []
        for k in range(2):
            b0, b1, m = [int(v) for v in res[j, k]]
            if m == 0:
                pk.append(float("nan"))                                  # numpy: percentile of an empty selection
                continue
            _, _, gamma = percentile_neighbours(m, 95)
            # float32 distances (float64_distances=False, :40): correctly rounded square roots of exact integers
            lo = np.sqrt(np.float64(b0)).astype(np.float32)
            hi = np.sqrt(np.float64(b1)).astype(np.float32)
            pk.append(float(percentile_linear_from_sorted_pair(lo, hi, gamma)))
        hd95[lab - 1] = np.maximum(pk[0], pk[1])
    # true division on the host (like the CPU capture of the reference); torch's device kernel would multiply by the reciprocal,
    # 1 ulp off for a precision that is not a power of two
    return torch.as_tensor(hd95 * 1 / precision).to(dev)
