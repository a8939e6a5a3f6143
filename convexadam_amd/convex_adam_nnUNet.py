"""Multi-channel (label / nnUNet feature) variant, mirror of the reference's
`convexAdam.convex_adam_nnUNet` (src/convexAdam/convex_adam_nnUNet.py): extract_features (:19-38) and
convex_adam (:41-159).  The weighted one-hot features are built by HIP kernels (label histogram,
expansion) and the registration itself is the same C-ABI pipeline with C = number of labels present.
The reference stores features in fp16; this engine keeps float32 throughout.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from ._lib import check, f32c, lib, ptr, require_device_tensor, stream_ptr
from .convex_adam_MIND import register_pair_device


def extract_features(pred_fixed, pred_moving, mult=10.0, device=None):
    """Label maps (H,W,D) -> weighted one-hot features (1,C,H,W,D) x 2; C = labels present in either map,
    w_c = (n_c_fix + n_c_mov + eps)^-0.3 normalised to mean 1, features = mult * w_c * onehot.
    (convex_adam_nnUNet.py:19-38; `mult` = 10 there, a parameter in self_configuring/convexAdam_hyper_util.py:64-83)"""
    device = torch.device(device if device is not None else "cuda")
    lf = require_device_tensor(f32c(pred_fixed.to(device)), "pred_fixed")
    lm = require_device_tensor(f32c(pred_moving.to(device)), "pred_moving")
    H, W, D = [int(s) for s in lf.shape[-3:]]
    V = H * W * D
    max_label = int(max(float(lf.max()), float(lm.max())))
    hist = torch.zeros((2, max_label + 1), dtype=torch.int64, device=device)
    L = lib()
    with torch.cuda.device(device):
        check(L.cvx_label_histogram_i64(ptr(lf), V, max_label, ptr(hist[0]), stream_ptr(device)))
        check(L.cvx_label_histogram_i64(ptr(lm), V, max_label, ptr(hist[1]), stream_ptr(device)))
    h_host = hist.cpu().numpy()
    present = np.zeros(max_label + 1, np.int32)
    weights = np.zeros(max_label + 1, np.float32)
    Cn = L.cvx_label_weights_host(h_host[0].ctypes.data_as(C.c_void_p), h_host[1].ctypes.data_as(C.c_void_p), max_label,
                                  present.ctypes.data_as(C.c_void_p), weights.ctypes.data_as(C.c_void_p))
    present_d = torch.from_numpy(present[:Cn].copy()).to(device)
    weights_d = torch.from_numpy(weights[:Cn].copy()).to(device)
    ff = torch.empty((1, Cn, H, W, D), dtype=torch.float32, device=device)
    fm = torch.empty_like(ff)
    with torch.cuda.device(device):
        check(L.cvx_label_features_f32(ptr(lf), V, Cn, ptr(present_d), ptr(weights_d), float(mult), ptr(ff), stream_ptr(device)))
        check(L.cvx_label_features_f32(ptr(lm), V, Cn, ptr(present_d), ptr(weights_d), float(mult), ptr(fm), stream_ptr(device)))
    return ff, fm


def convex_adam_pt(pred_fixed, pred_moving, lambda_weight, grid_sp, disp_hw, selected_niter, selected_smooth,
                   grid_sp_adam=2, ic=True, device="cuda"):
    """Tensor-level entry: label maps in, np.ndarray (H,W,D,3) float64 out (values quantised through fp16
    like the reference's `.cpu().half()` at :151-154)."""
    ff, fm = extract_features(pred_fixed, pred_moving, device=device)
    smooth = selected_smooth if selected_smooth in (3, 5) else 0       # only 3 and 5 act in the reference (:136-144)
    # the packaged nnUNet path keeps the MIND cost scale 12 even when C != 12 (:127)
    disp = register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], lambda_weight=lambda_weight, grid_sp=grid_sp,
                                disp_hw=disp_hw, selected_niter=selected_niter, selected_smooth=smooth,
                                grid_sp_adam=grid_sp_adam, ic=ic, cost_scale=12.0)
    return disp.permute(1, 2, 3, 0).half().cpu().numpy().astype(float)


def convex_adam(path_pred_fixed, path_pred_moving, lambda_weight, grid_sp, disp_hw, selected_niter, selected_smooth,
                grid_sp_adam=2, ic=True, result_path='./'):
    """File wrapper (convex_adam_nnUNet.py:41-159): NIfTI label maps in, `disp.nii.gz` out."""
    from .nifti_io import load_affine, load_fdata, save_image      # nibabel when installed, else the built-in NIfTI-1 reader / writer
    pred_fixed = torch.from_numpy(load_fdata(path_pred_fixed)).float()
    pred_moving = torch.from_numpy(load_fdata(path_pred_moving)).float()
    torch.cuda.synchronize()
    t0 = time.time()
    displacements = convex_adam_pt(pred_fixed, pred_moving, lambda_weight, grid_sp, disp_hw, selected_niter, selected_smooth,
                                   grid_sp_adam, ic)
    torch.cuda.synchronize()
    print('case time: ', time.time() - t0)
    save_image(displacements, load_affine(path_pred_fixed), os.path.join(result_path, 'disp.nii.gz'))
