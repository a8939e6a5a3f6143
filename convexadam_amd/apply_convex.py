"""Mirror of the reference's `src/convexAdam/apply_convex.py::apply_convex` (:13-24) on the HIP device.

    apply_convex(disp, moving) -> np.ndarray     warped = scipy.ndimage.map_coordinates(moving, disp + identity, order=1)

`disp` is the (H,W,D,3) field convex_adam_pt returns (channel a = displacement along axis a, voxels); `moving` any
(H,W,D) array or tensor.  Like scipy, the interpolation runs in float64; numpy inputs are converted with
`astype(float)` exactly as validate_image does (convex_adam_utils.py:268-279), tensors keep their dtype for the result.
apply_convex_original_moving (:27-78) resamples with SimpleITK and rotates by the direction cosines: host-side geometry,
out of scope.
"""
import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr
from .convex_adam_utils import validate_image


def apply_convex(disp, moving, device=None) -> np.ndarray:
    # validate_image(img, dtype=float): numpy input becomes float64 (convex_adam_utils.py:276), tensors pass through
    mov_t = validate_image(moving.astype(float) if isinstance(moving, np.ndarray) else moving)
    disp_t = validate_image(disp.astype(float) if isinstance(disp, np.ndarray) else disp)
    if disp_t.dim() != 4 or disp_t.shape[-1] != 3 or tuple(disp_t.shape[:3]) != tuple(mov_t.shape):
        raise ValueError("apply_convex: disp must be (H,W,D,3) matching moving (H,W,D)")
    dev = torch.device(device) if device is not None else (mov_t.device if mov_t.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    out_dtype = mov_t.dtype
    m = mov_t.to(dev, torch.float64).contiguous()
    d = disp_t.to(dev, torch.float64).contiguous()
    H, W, D = [int(v) for v in m.shape]
    out = torch.empty_like(m)
    with torch.cuda.device(dev):
        check(lib().cvx_map_coordinates_linear_f64(ptr(m), ptr(d), H, W, D, ptr(out), stream_ptr(dev)))
    return out.to(out_dtype).cpu().numpy()
