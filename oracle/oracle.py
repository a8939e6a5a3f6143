"""ctypes/numpy front-end of oracle/libcvx_oracle.so (the C restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package `convexadam_amd` imports this module; it is
used by tests/, by __graft_entry__.smoke() as the checker and by bench.py's `cpu_baseline` leg.
Every function mirrors one reference operator; see cvx_oracle.c for the reference file:line cites.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcvx_oracle.so")
_lib = None


class Smoother(C.Structure):
    _fields_ = [("kind", C.c_int), ("n_boxes", C.c_int), ("box_k", C.c_int * 4), ("gauss_w", C.c_float * 5)]


def make_smoother(boxes=None, gauss_w=None):
    sm = Smoother()
    if gauss_w is not None:
        sm.kind = 1
        for i in range(5):
            sm.gauss_w[i] = float(gauss_w[i])
    else:
        sm.kind = 0
        sm.n_boxes = len(boxes)
        for i, k in enumerate(boxes):
            sm.box_k[i] = int(k)
    return sm

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "cvx_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libcvx_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_expf_array.argtypes = [_f32p, _f32p, C.c_int64]
        L.orc_linspace_pm1.argtypes = [C.c_int, _f32p]
        L.orc_affine_base.argtypes = [C.c_int, _f32p]
        L.orc_disp_mesh.argtypes = [C.c_int, _f32p]
        L.orc_box_zero.argtypes = [_f32p, _f32p] + [C.c_int] * 5
        L.orc_box_zero_backward.argtypes = [_f32p, _f32p] + [C.c_int] * 5
        L.orc_box_grow.argtypes = [_f32p, _f32p] + [C.c_int] * 5
        L.orc_avgpool_stride.argtypes = [_f32p, _f32p] + [C.c_int] * 5
        L.orc_box3_replicate.argtypes = [_f32p, _f32p] + [C.c_int] * 3
        L.orc_mindssc.argtypes = [_f32p] + [C.c_int] * 5 + [_f32p, C.POINTER(C.c_float)]
        L.orc_correlate.argtypes = [_f32p, _f32p] + [C.c_int] * 5 + [_f32p, _i64p]
        L.orc_correlate_ex.argtypes = [_f32p, _f32p] + [C.c_int] * 7 + [_f32p, _i64p]
        L.orc_coupled_convex.argtypes = [_f32p, _i64p, _f32p] + [C.c_int] * 4 + [_f32p]
        L.orc_grid_sample.argtypes = [_f32p] + [C.c_int] * 4 + [_f32p] + [C.c_int] * 3 + [_f32p]
        L.orc_inverse_consistency.argtypes = [_f32p, _f32p] + [C.c_int] * 4 + [_f32p, _f32p]
        L.orc_resize_trilinear.argtypes = [_f32p] + [C.c_int] * 4 + [_f32p] + [C.c_int] * 3
        L.orc_adam_run.argtypes = [_f32p, _f32p] + [C.c_int] * 4 + [_f32p, _f32p, _f32p, C.c_float, C.c_int,
                                                                   C.c_int, C.c_float, _f32p, C.c_void_p, C.c_void_p]
        L.orc_smooth.argtypes = [_f32p, _f32p] + [C.c_int] * 4 + [C.POINTER(Smoother), C.c_int]
        L.orc_adam_run_smoother.argtypes = [_f32p, _f32p] + [C.c_int] * 4 + [_f32p, _f32p, _f32p, C.c_float, C.c_int,
                                                                            C.c_int, C.c_float, _f32p, C.c_void_p, C.c_void_p, C.POINTER(Smoother)]
        L.orc_fast_box3x3.argtypes = [_f32p, _f32p] + [C.c_int] * 4
        L.orc_adam_run_fast.argtypes = [_f32p, _f32p] + [C.c_int] * 4 + [_f32p, _f32p, _f32p, C.c_float, C.c_int,
                                                                        C.c_int, C.c_float, _f32p, C.c_void_p, C.c_int, C.c_int]
        L.orc_fast_boxchain.argtypes = [_f32p, _f32p] + [C.c_int] * 5 + [C.c_void_p, C.c_int]
        L.orc_adam_run_fast_smoother.argtypes = [_f32p, _f32p] + [C.c_int] * 4 + [_f32p, _f32p, _f32p, C.c_float, C.c_int,
                                                                                 C.c_int, C.c_float, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_label_features.argtypes = [_f32p, _f32p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_label_features.restype = C.c_int
        L.orc_feature_transform.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_mind_tables.argtypes = [_i32p, _i32p, _i32p]
        L.orc_set_sqrt_table.argtypes = [C.c_void_p]
        L.orc_set_exp_table.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_set_mean_threads.argtypes = [C.c_int, C.c_int]
        L.orc_torch_sum.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int]
        L.orc_torch_sum.restype = C.c_float
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


_sqrt_tables = None


def sqrt_codes_from_low_bitmaps(normal, denormal):
    """Packed bit maps of the classes whose root is one ulp LOW (tests/golden/mkl_vssqrt_low.npz) -> the 2-bit code table."""
    low = np.concatenate([np.unpackbits(np.ascontiguousarray(normal, np.uint8), bitorder="little"),
                          np.unpackbits(np.ascontiguousarray(denormal, np.uint8), bitorder="little")]).astype(np.uint8) * 2
    c = low.reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)


def set_sqrt_table(codes=None, denormal=None):
    """Reference-build sqrt for the Adam update: `codes` = 2-bit table (orc_set_sqrt_table), or (normal, denormal) low bit maps of
    tests/golden/mkl_vssqrt_low.npz; None = IEEE sqrt (default)."""
    global _sqrt_tables
    if codes is None:
        _sqrt_tables = None
        lib().orc_set_sqrt_table(None)
        return
    if denormal is not None:
        codes = sqrt_codes_from_low_bitmaps(codes, denormal)
    _sqrt_tables = np.ascontiguousarray(codes, np.uint8)                                   # keep alive
    assert _sqrt_tables.size == 3 << 21
    lib().orc_set_sqrt_table(_sqrt_tables.ctypes.data_as(C.c_void_p))


_exp_table = None


def set_exp_table(table=None, first=0, count=0):
    """MKL-vsExp correction for MINDSSC: 2 bits per argument (see orc_set_exp_table); None = orc_expf (default)."""
    global _exp_table
    if table is None:
        _exp_table = None
        lib().orc_set_exp_table(None, 0, 0)
        return
    _exp_table = np.ascontiguousarray(table, np.uint8)
    lib().orc_set_exp_table(_exp_table.ctypes.data_as(C.c_void_p), int(first), int(count))


def set_mean_threads(threads=0, vec=8):
    """MINDSSC's global mean as torch evaluates it with `threads` threads (0 = the exactly rounded mean, default)."""
    lib().orc_set_mean_threads(int(threads), int(vec))


def torch_sum(x, threads, vec=8):
    x = _f(x).reshape(-1)
    return float(lib().orc_torch_sum(x, x.size, int(threads), int(vec)))


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


def expf(x):
    x = _f(x); out = np.empty_like(x); lib().orc_expf_array(x.reshape(-1), out.reshape(-1), x.size); return out


def linspace_pm1(S):
    out = np.empty(S, np.float32); lib().orc_linspace_pm1(S, out); return out


def affine_base(S):
    out = np.empty(S, np.float32); lib().orc_affine_base(S, out); return out


def disp_mesh(hw):
    """(3, n^3) search mesh, convex_adam_MIND.py:127."""
    n = 2 * hw + 1
    out = np.empty((3, n ** 3), np.float32); lib().orc_disp_mesh(hw, out.reshape(-1)); return out


def box_zero(x, k=3):
    x = _f(x); c, h, w, d = x.shape
    out = np.empty_like(x); lib().orc_box_zero(x.reshape(-1), out.reshape(-1), c, h, w, d, k); return out


def box_grow(x, k):
    """avg_pool3d(k EVEN, stride 1, padding k//2): (c, h, w, d) -> (c, h+1, w+1, d+1) (convex_adam_MIND.py:184-191 with an even selected_smooth)."""
    x = _f(x); c, h, w, d = x.shape
    assert k > 0 and k % 2 == 0
    out = np.empty((c, h + 1, w + 1, d + 1), np.float32); lib().orc_box_grow(x.reshape(-1), out.reshape(-1), c, h, w, d, k); return out


def box_zero_backward(x, k=3):
    x = _f(x); c, h, w, d = x.shape
    out = np.empty_like(x); lib().orc_box_zero_backward(x.reshape(-1), out.reshape(-1), c, h, w, d, k); return out


def avgpool_stride(x, g):
    x = _f(x); c, h, w, d = x.shape
    out = np.empty((c, h // g, w // g, d // g), np.float32)
    lib().orc_avgpool_stride(x.reshape(-1), out.reshape(-1), c, h, w, d, g); return out


def box3_replicate(x):
    x = _f(x); h, w, d = x.shape
    out = np.empty_like(x); lib().orc_box3_replicate(x.reshape(-1), out.reshape(-1), h, w, d); return out


def feature_transform(obj):
    """Index of the nearest ZERO element of `obj` for every voxel, (3,H,W,D) int32: restatement of
    scipy.ndimage.distance_transform_edt(obj, return_indices=True)[1] including its tie-breaking (cvx_oracle.c)."""
    o = _f(np.asarray(obj) != 0)
    H, W, D = o.shape
    feat = np.empty((3, H, W, D), np.int32)
    lib().orc_feature_transform(o, H, W, D, feat.ctypes.data_as(C.c_void_p))
    return feat


def replicate_fill(img, mask):
    """Masked replicate fill of convex_adam_MIND.py:40-51 (even extents): erode the mask (replicate box3 > 0.9), find for
    every half-resolution voxel the nearest in-mask voxel (Euclidean feature transform with scipy's tie-breaking), gather,
    x2 trilinear up-sample, keep the original values inside the eroded mask."""
    img = _f(img); mask = _f(mask); H, W, D = img.shape
    m = (box3_replicate(mask) > np.float32(0.9)).astype(np.float32)
    idx = feature_transform(m[::2, ::2, ::2] == 0).astype(np.int64)
    lin = idx[0] * D // 2 * W // 2 + idx[1] * D // 2 + idx[2]
    half = img[::2, ::2, ::2].reshape(-1)[lin].astype(np.float32)
    up = resize_trilinear(half[None], (2 * half.shape[0], 2 * half.shape[1], 2 * half.shape[2]))[0]
    return np.where(m != 0, img, up).astype(np.float32), m


def mindssc(img, radius=2, dilation=2, return_mean=False):
    """img (H,W,D) -> (12,H,W,D); convex_adam_utils.py:24-68."""
    img = _f(img); h, w, d = img.shape
    out = np.empty((12, h, w, d), np.float32); gm = C.c_float(0)
    lib().orc_mindssc(img.reshape(-1), h, w, d, radius, dilation, out.reshape(-1), C.byref(gm))
    return (out, gm.value) if return_mean else out


def correlate(fix, mov, disp_hw, cost="ssd", n_box=2):
    """fix/mov (C,h,w,d) -> ssd (n^3,h,w,d), argmin (h,w,d) int64; convex_adam_utils.py:72-89.
    cost "sad" / n_box 1: the variants of the challenge scripts (l2r_2021_convexAdam_task3_docker.py:54,56; task2:60)."""
    fix = _f(fix); mov = _f(mov); c, h, w, d = fix.shape; n = 2 * disp_hw + 1
    ssd = np.empty((n ** 3, h, w, d), np.float32); am = np.empty((h, w, d), np.int64)
    lib().orc_correlate_ex(fix.reshape(-1), mov.reshape(-1), c, h, w, d, disp_hw, 1 if cost == "sad" else 0, int(n_box),
                           ssd.reshape(-1), am.reshape(-1))
    return ssd, am


def coupled_convex(ssd, argmin, mesh, disp_hw):
    """-> (3,h,w,d); convex_adam_utils.py:93-109."""
    ssd = _f(ssd); _, h, w, d = ssd.shape
    out = np.empty((3, h, w, d), np.float32)
    lib().orc_coupled_convex(ssd.reshape(-1), np.ascontiguousarray(argmin, np.int64).reshape(-1), _f(mesh).reshape(-1),
                             h, w, d, disp_hw, out.reshape(-1))
    return out


def grid_sample(vol, grid):
    """vol (C,h,w,d), grid (ho,wo,do,3) normalised (x,y,z) -> (C,ho,wo,do)."""
    vol = _f(vol); grid = _f(grid); c, h, w, d = vol.shape; ho, wo, do_, _ = grid.shape
    out = np.empty((c, ho, wo, do_), np.float32)
    lib().orc_grid_sample(vol.reshape(-1), c, h, w, d, grid.reshape(-1), ho, wo, do_, out.reshape(-1)); return out


def inverse_consistency(f1, f2, iters=20):
    f1 = _f(f1); f2 = _f(f2); _, h, w, d = f1.shape
    o1 = np.empty_like(f1); o2 = np.empty_like(f2)
    lib().orc_inverse_consistency(f1.reshape(-1), f2.reshape(-1), h, w, d, iters, o1.reshape(-1), o2.reshape(-1))
    return o1, o2


def resize_trilinear(x, size):
    x = _f(x); c, h, w, d = x.shape; H, W, D = size
    out = np.empty((c, H, W, D), np.float32)
    lib().orc_resize_trilinear(x.reshape(-1), c, h, w, d, out.reshape(-1), H, W, D); return out


def smooth(x, smoother, backward=False):
    x = _f(x); c, h, w, d = x.shape
    out = np.empty_like(x)
    lib().orc_smooth(x.reshape(-1), out.reshape(-1), c, h, w, d, C.byref(smoother), 1 if backward else 0)
    return out


def fast_box3x3(x):
    """Fast-mode restatement of box3(box3(box3(x))) (zero padding per stage): separable chained sums, one final scale."""
    x = _f(x); c, h, w, d = x.shape
    out = np.empty_like(x); lib().orc_fast_box3x3(x.reshape(-1), out.reshape(-1), c, h, w, d); return out


def fast_boxchain(x, sizes, reverse=False):
    """Fast-mode restatement of a chain of zero-padded box filters (kovesi_spline): separable sums, one final scale; reverse = adjoint order."""
    x = _f(x); c, h, w, d = x.shape
    out = np.empty_like(x)
    arr = (C.c_int * len(sizes))(*[int(k) for k in sizes])
    lib().orc_fast_boxchain(x.reshape(-1), out.reshape(-1), c, h, w, d, len(sizes), C.cast(arr, C.c_void_p), 1 if reverse else 0)
    return out


def adam_run(F2, M2, P, lambda_weight, niter, m=None, v=None, step0=0, cost_scale=12.0, want_grad=False, smoother=None, mode="exact",
             keep_last_step=True):
    """Runs `niter` Adam iterations in place on copies; returns dict(P, m, v, U, G, loss).
    mode="fast": the tolerance-graded throughput arithmetic (orc_adam_run_fast; three 3^3 boxes only), "fast_all": the same with the
    forward boxes in separable arithmetic too (faster, but outside the acceptance criteria); keep_last_step=False skips the
    gradient + update of the final iteration like the whole-pair pipeline does (U is the same either way)."""
    F2 = _f(F2); M2 = _f(M2); c, h, w, d = F2.shape
    P = _f(P).copy(); m = np.zeros_like(P) if m is None else _f(m).copy(); v = np.zeros_like(P) if v is None else _f(v).copy()
    U = np.zeros_like(P); G = np.zeros_like(P) if want_grad else None; loss = np.zeros(max(niter, 1), np.float32)
    if mode in ("fast", "fast_all"):
        lib().orc_adam_run_fast_smoother(F2.reshape(-1), M2.reshape(-1), c, h, w, d, P.reshape(-1), m.reshape(-1), v.reshape(-1),
                                         float(lambda_weight), int(niter), int(step0), float(cost_scale), U.reshape(-1),
                                         G.ctypes.data_as(C.c_void_p) if G is not None else None, 1 if keep_last_step else 0,
                                         1 if mode == "fast_all" else 0, C.cast(C.pointer(smoother), C.c_void_p) if smoother is not None else None)
        return dict(P=P, m=m, v=v, U=U, G=G, loss=None)
    assert mode == "exact", mode
    lib().orc_adam_run_smoother(F2.reshape(-1), M2.reshape(-1), c, h, w, d, P.reshape(-1), m.reshape(-1), v.reshape(-1),
                                float(lambda_weight), int(niter), int(step0), float(cost_scale), U.reshape(-1),
                                G.ctypes.data_as(C.c_void_p) if G is not None else None, loss.ctypes.data_as(C.c_void_p),
                                C.byref(smoother) if smoother is not None else None)
    return dict(P=P, m=m, v=v, U=U, G=G, loss=loss[:niter])


def label_features(lab_fix, lab_mov, mult=10.0):
    lf = _f(lab_fix).reshape(-1); lm = _f(lab_mov).reshape(-1); V = lf.size
    mx = int(max(lf.max(), lm.max()))
    Cn = lib().orc_label_features(lf, lm, V, mx, mult, None, None, None)
    ff = np.empty((Cn,) + tuple(np.shape(lab_fix)), np.float32); fm = np.empty_like(ff); pres = np.empty(Cn, np.int32)
    lib().orc_label_features(lf, lm, V, mx, mult, ff.ctypes.data_as(C.c_void_p), fm.ctypes.data_as(C.c_void_p),
                             pres.ctypes.data_as(C.c_void_p))
    return ff, fm, pres


# ---- whole pipeline (convex_adam_MIND.py:64-202), composed from the operators above -------------
def _h(x):
    """fp16 storage of a float32 array (round to nearest even), values back in float32."""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def convex_adam_pipeline(img_fixed, img_moving, mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=6, disp_hw=4,
                         selected_niter=80, selected_smooth=0, grid_sp_adam=2, ic=True, features=None,
                         return_stages=False, cost="ssd", n_box=2, n_spline_pools=3, storage="fp32", adam_mode="exact"):
    """float32 restatement of convex_adam_pt(); returns (H,W,D,3) float64 like the reference.
    cost / n_box / n_spline_pools: the operator variants of the challenge scripts (l2r_2021_convexAdam_task3_docker.py:54,56,191;
    task2:60); storage="fp16": pooled features and cost volume rounded to half precision (float32 accumulation)."""
    q = _h if storage == "fp16" else (lambda a: a)
    st = {}
    if features is None:
        img_fixed = _f(img_fixed); img_moving = _f(img_moving)
        ffix = mindssc(img_fixed, mind_r, mind_d); fmov = mindssc(img_moving, mind_r, mind_d)
    else:
        ffix, fmov = _f(features[0]), _f(features[1])
    H, W, D = ffix.shape[1:]
    fs = q(avgpool_stride(ffix, grid_sp)); ms = q(avgpool_stride(fmov, grid_sp))
    h, w, d = fs.shape[1:]
    mesh = disp_mesh(disp_hw)

    def corr(a, b):
        ssd, am = correlate(a, b, disp_hw, cost=cost, n_box=n_box)
        if storage == "fp16":
            ssd = _h(ssd)
            am = ssd.reshape(ssd.shape[0], -1).argmin(0).reshape(am.shape).astype(np.int64)     # first minimum of the stored values
        return ssd, am

    ssd, am = corr(fs, ms)
    soft = coupled_convex(ssd, am, mesh, disp_hw)
    st.update(fs=fs, ms=ms, argmin=am, soft=soft)
    if ic:
        scale = (np.array([h - 1, w - 1, d - 1], np.float32) / np.float32(2)).reshape(3, 1, 1, 1)
        ssd_, am_ = corr(ms, fs)
        soft_ = coupled_convex(ssd_, am_, mesh, disp_hw)
        i1, _ = inverse_consistency((soft / scale)[::-1], (soft_ / scale)[::-1], 15)
        disp_hr = resize_trilinear((i1[::-1] * scale) * np.float32(grid_sp), (H, W, D))
        st.update(soft_=soft_, ice=i1)
    else:
        disp_hr = soft
    st.update(disp_hr0=disp_hr)
    if lambda_weight > 0:
        g = grid_sp_adam
        F2 = q(avgpool_stride(ffix, g)); M2 = q(avgpool_stride(fmov, g))
        disp_lr = resize_trilinear(disp_hr, (H // g, W // g, D // g))
        P0 = disp_lr / np.float32(g)
        r = adam_run(F2, M2, P0, lambda_weight, selected_niter, smoother=make_smoother([3, 3]) if n_spline_pools == 2 else None,
                     mode=adam_mode)
        st.update(P0=P0, U=r["U"], F2=F2, M2=M2)
        disp_hr = resize_trilinear(r["U"] * np.float32(g), (H, W, D))
        if selected_smooth > 0:
            for _ in range(3):      # an even kernel grows the field by one voxel per axis and pool (the reference overwrites its own "+1", :185-189)
                disp_hr = box_zero(disp_hr, selected_smooth) if selected_smooth % 2 else box_grow(disp_hr, selected_smooth)
    out = np.stack([disp_hr[0], disp_hr[1], disp_hr[2]], 3).astype(np.float64)
    return (out, st) if return_stages else out
