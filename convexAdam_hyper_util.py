"""Shim: `from convexAdam_hyper_util import ...` (what the reference's sweep scripts write,
self_configuring/convex_run_withconfig.py:13-16) resolves to the HIP-backed mirror."""
from convexadam_amd.convexAdam_hyper_util import *  # noqa: F401,F403
from convexadam_amd.convexAdam_hyper_util import (GaussianSmoothing, MINDSSC, correlate, coupled_convex,  # noqa: F401
                                                  extract_features, extract_features_nnunet, inverse_consistency,
                                                  kovesi_spline, jacobian_determinant_3d, dice_coeff, sort_rank, cupy_hd95,
                                                  warp_labels_nearest, jacobian_log_std_and_folding, tre_at_keypoints)
