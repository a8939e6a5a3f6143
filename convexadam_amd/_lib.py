"""ctypes binding of libconvexadam_hip.so (include/convexadam_hip.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  If the shared object is
missing (not built) or an operator is called with tensors that do not live on a HIP device, the call
fails loudly.  PyTorch is used only for device memory, streams and dtype conversion.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CONVEXADAM_HIP_LIB") or os.path.join(_HERE, "csrc", "libconvexadam_hip.so")     # (override: race-stress build)

ABI_VERSION = 2          # CVX_ABI_VERSION of include/convexadam_hip.h this binding was written against
CVX_OK, CVX_ERR_INVALID_ARG, CVX_ERR_WORKSPACE, CVX_ERR_LAUNCH, CVX_ERR_UNSUPPORTED = 0, -1, -2, -3, -4


class CvxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libconvexadam_hip error %d: %s" % (code, msg))
        self.code = code


class Smoother(C.Structure):
    """struct cvx_smoother (include/convexadam_hip.h)."""
    _fields_ = [("kind", C.c_int), ("n_boxes", C.c_int), ("box_k", C.c_int * 4), ("gauss_w", C.c_float * 5)]


class CorrOpts(C.Structure):
    """struct cvx_corr_opts (include/convexadam_hip.h)."""
    _fields_ = [("cost", C.c_int), ("n_box", C.c_int), ("fast", C.c_int), ("f16", C.c_int)]


class PairParams(C.Structure):
    """struct cvx_pair_params (include/convexadam_hip.h)."""
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("D", C.c_int), ("mind_r", C.c_int), ("mind_d", C.c_int),
                ("lambda_weight", C.c_float), ("grid_sp", C.c_int), ("disp_hw", C.c_int), ("selected_niter", C.c_int),
                ("selected_smooth", C.c_int), ("grid_sp_adam", C.c_int), ("ic", C.c_int), ("n_feat", C.c_int),
                ("cost_scale", C.c_float), ("cost", C.c_int), ("n_box", C.c_int), ("n_spline_pools", C.c_int), ("corr_fast", C.c_int),
                ("fp16_storage", C.c_int), ("ctx", C.c_void_p), ("adam_fast", C.c_int), ("reserved_", C.c_int * 3)]


_vp, _i, _f, _sz, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "cvx_version": (_i, []),
    "cvx_last_error": (C.c_char_p, []),
    "cvx_device_count": (_i, []),
    "cvx_context_create": (_vp, []),
    "cvx_context_destroy": (None, [_vp]),
    "cvx_context_bind": (_vp, [_vp]),
    "cvx_context_set_option": (_i, [_vp, C.c_char_p, C.c_longlong]),
    "cvx_context_get_option": (C.c_longlong, [_vp, C.c_char_p]),
    "cvx_context_set_adam_sqrt_table": (_i, [_vp, _vp, _vp]),
    "cvx_context_set_mind_exp_table": (_i, [_vp, _vp, C.c_uint, C.c_uint, _vp]),
    "cvx_set_adam_sqrt_table": (_i, [_vp]),
    "cvx_set_mind_exp_table": (_i, [_vp, C.c_uint, C.c_uint]),
    "cvx_expf_f32": (_i, [_vp, _vp, _sz, _vp]),
    "cvx_set_option": (_i, [C.c_char_p, C.c_longlong]),
    "cvx_get_option": (C.c_longlong, [C.c_char_p]),
    "cvx_affine_base_host": (None, [_i, _vp]),
    "cvx_disp_mesh_host": (None, [_i, _vp]),
    "cvx_disp_mesh_f32": (_i, [_i, _vp, _vp]),
    "cvx_affine_base_f32": (_i, [_i, _vp, _vp]),
    "cvx_mindssc_workspace_bytes": (_sz, [_i] * 5),
    "cvx_mindssc_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "cvx_mindssc_pooled_scratch_bytes": (_sz, [_i] * 7),
    "cvx_mindssc_pooled_f32": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "cvx_avgpool_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cvx_round_f16_f32": (_i, [_vp, _i64, _vp]),
    "cvx_pack_field_f64": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "cvx_box_smooth_workspace_bytes": (_sz, [_i] * 5),
    "cvx_box_smooth_f32": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "cvx_box_grow_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cvx_mask_erode_f32": (_i, [_vp, _i, _i, _i, _f, _vp, _vp]),
    "cvx_gather_f32": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "cvx_select_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "cvx_label_histogram_i64": (_i, [_vp, _i64, _i, _vp, _vp]),
    "cvx_label_weights_host": (_i, [_vp, _vp, _i, _vp, _vp]),
    "cvx_label_features_f32": (_i, [_vp, _i64, _i, _vp, _vp, _f, _vp, _vp]),
    "cvx_correlate_workspace_bytes": (_sz, [_i] * 5),
    "cvx_correlate_ex_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cvx_correlate_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "cvx_coupled_convex_workspace_bytes": (_sz, [_i] * 4),
    "cvx_coupled_convex_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "cvx_coupled_convex_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "cvx_inverse_consistency_workspace_bytes": (_sz, [_i] * 3),
    "cvx_inverse_consistency_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cvx_resize_trilinear_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "cvx_grid_sample_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "cvx_adam_workspace_bytes": (_sz, [_i] * 4),
    "cvx_adam_run_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                              _vp, _vp, _sz, _vp]),
    "cvx_smooth_workspace_bytes": (_sz, [_i] * 4),
    "cvx_smooth_f32": (_i, [_vp, _i, _i, _i, _i, C.POINTER(Smoother), _i, _vp, _vp, _sz, _vp]),
    "cvx_adam_run_smoother_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                       _vp, C.POINTER(Smoother), _vp, _sz, _vp]),
    "cvx_adam_run_ex_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                 _vp, C.POINTER(Smoother), _i, _vp, _sz, _vp]),
    "cvx_adam_run_fast_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                   _vp, _vp, _sz, _vp]),
    "cvx_adam_run_fast_all_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                       _vp, _vp, _sz, _vp]),
    "cvx_adam_run_mode_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                   _vp, C.POINTER(Smoother), _i, _vp, _sz, _vp]),
    "cvx_smooth_fast_f32": (_i, [_vp, _i, _i, _i, C.POINTER(Smoother), _i, _vp, _vp]),
    "cvx_box3_fast_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "cvx_register_pair_workspace_bytes": (_sz, [C.POINTER(PairParams)]),
    "cvx_register_pair_f32": (_i, [_vp, _vp, _vp, _vp, C.POINTER(PairParams), _vp, _vp, _vp, _sz, _vp]),
    "cvx_register_pair_snapshots_workspace_bytes": (_sz, [_vp, _i, _vp, _i]),
    "cvx_register_pair_snapshots_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "cvx_register_pairs_f32": (_i, [_i, _vp, _vp, _vp, _vp, C.POINTER(PairParams), _vp, _vp, _vp, _sz, _i, _vp]),
    "cvx_last_pair_profile": (_i, [_vp, _vp, _i]),
    "cvx_set_profiling": (None, [_i]),
    "cvx_jacobian_det_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "cvx_jacobian_stats_f64": (_i, [_vp, _i64, _vp, _vp]),
    "cvx_warp_labels_nearest_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cvx_label_overlap_i64": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "cvx_map_coordinates_linear_f64": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "cvx_feature_transform_workspace_bytes": (_sz, [_i, _i, _i]),
    "cvx_feature_transform_i32": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "cvx_feature_flat_index_i64": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cvx_label_mask_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "cvx_label_mask_scaled_f32": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "cvx_edt_sqdist_i32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "cvx_surface_hist_i64": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "cvx_hist_order_stats_i64": (_i, [_vp, _i, _i64, _i64, _vp, _vp]),
    "cvx_hist_percentile_neighbours_i64": (_i, [_vp, _i, _f, _vp, _vp]),
    "cvx_surface_hist_batch_i64": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp]),
    "cvx_hist_percentile_neighbours_batch_i64": (_i, [_vp, _i, _i, _f, _vp, _vp]),
    "cvx_edt_squared_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "cvx_edt_squared_i32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "cvx_edt_squared_labels_i32": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "cvx_label_bits_bytes": (_sz, [_i, _i, _i, _i]),
    "cvx_label_bits_u64": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "cvx_surface_distance_hist_i64": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _i, _i, _vp]),
    "cvx_surface_distance_hist_bits_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "cvx_surface_distance_hist_bits_i64": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _i, _i, _vp, _sz, _vp]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Loads the HIP library (raises if it has not been built -- never falls back)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError("%s not found: build it with `python -m convexadam_amd.csrc.build` "
                                       "(there is no CPU fallback)" % LIB_PATH)
                L = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)
                    fn.restype = res
                    fn.argtypes = args
                if L.cvx_version() != ABI_VERSION:
                    raise RuntimeError("%s reports ABI version %d, this binding needs %d (struct layouts differ): rebuild with "
                                       "`python -m convexadam_amd.csrc.build`" % (LIB_PATH, L.cvx_version(), ABI_VERSION))
                _lib = L
    return _lib


def check(rc):
    if rc != CVX_OK:
        raise CvxError(rc, lib().cvx_last_error().decode("utf-8", "replace"))


def require_device_tensor(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if t.device.type != "cuda":
        raise RuntimeError("%s lives on %s: the convexadam_amd operators run only on a HIP (ROCm 'cuda') device; "
                           "there is no CPU path" % (name, t.device))
    return t


def f32c(t):
    """float32, contiguous view/copy of a device tensor."""
    return t.detach().to(torch.float32).contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_tls = threading.local()


def workspace(nbytes, device):
    """Grow-only scratch buffer per (thread, device, stream): the library never allocates.  Per THREAD as well (round 5): two Python
    threads calling the drop-in API on the same stream interleave their launches (ctypes releases the GIL inside the C call); with a
    shared scratch buffer one pair's kernels would run between the other's on the same memory.  Every other piece of a call's data --
    inputs, outputs, optimiser state -- already belongs to the call; the buffers of a thread are freed with the thread."""
    ws = getattr(_tls, "workspaces", None)
    if ws is None:
        ws = _tls.workspaces = {}
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = ws.get(key)
    if buf is None or buf.numel() < nbytes:
        ws[key] = buf = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
    return buf


def release_workspaces():
    """Drops the calling thread's scratch buffers."""
    if getattr(_tls, "workspaces", None):
        _tls.workspaces.clear()
