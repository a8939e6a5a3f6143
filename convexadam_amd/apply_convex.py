"""Mirror of the reference's `src/convexAdam/apply_convex.py::apply_convex` (:13-24) on the HIP device.

    apply_convex(disp, moving) -> np.ndarray     warped = scipy.ndimage.map_coordinates(moving, disp + identity, order=1)

`disp` is the (H,W,D,3) field convex_adam_pt returns (channel a = displacement along axis a, voxels); `moving` any
(H,W,D) array, image or tensor.  Like scipy, the interpolation runs in float64; non-tensor inputs of any dtype are converted with
`astype(float)` exactly as validate_image does (convex_adam_utils.py:268-279) and give a float64 result; tensors keep their dtype
for the result (integer tensors are rounded to nearest like scipy's integer output arrays).
apply_convex_original_moving (:27-78): the field is first carried onto the grid, axes and voxel size of the original moving image
(host-side SimpleITK geometry, convex_adam_utils.rescale_displacement_field), then the warp above runs on the device.
"""
import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr
from .convex_adam_utils import rescale_displacement_field, validate_image


def apply_convex(disp, moving, device=None) -> np.ndarray:
    # validate_image(img, dtype=float): every non-tensor input (numpy, SimpleITK, nibabel; any integer type) becomes float64
    # (convex_adam_utils.py:268-279), tensors pass through with their dtype
    mov_t = validate_image(moving)
    disp_t = validate_image(disp)
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
    if not out_dtype.is_floating_point:
        out = torch.trunc(out + torch.where(out > 0, 0.5, -0.5))   # scipy's integer outputs: (type)(t > 0 ? t + 0.5 : t - 0.5), ni_interpolation.c
    return out.to(out_dtype).cpu().numpy()


def apply_convex_original_moving(disp, moving_image_original, fixed_image_original, fixed_image_resampled):
    """Warp the ORIGINAL moving image (its own grid, orientation and spacing) with a field estimated on the resampled fixed grid
    (apply_convex.py:27-78): images in (SimpleITK, or convexadam_amd.imageio.Image), float32 image with the moving image's geometry out."""
    from .imageio import Image
    field = validate_image(disp).cpu().numpy()
    field = rescale_displacement_field(field, moving_image_original, fixed_image_original, fixed_image_resampled)
    warped = apply_convex(disp=field, moving=moving_image_original)
    if isinstance(moving_image_original, Image):
        out = Image(warped.astype(np.float32))
    else:
        import SimpleITK as sitk  # noqa: N813
        out = sitk.GetImageFromArray(warped.astype(np.float32))
    out.CopyInformation(moving_image_original)
    return out


def main(argv=None):
    """python -m convexAdam.apply_convex --input_field disp.nii.gz --input_moving moving.nii.gz --output_warped warped.nii.gz
    (apply_convex.py:81-97): NIfTI through nibabel when installed, else through the built-in reader / writer."""
    import argparse
    from . import nifti_io
    ap = argparse.ArgumentParser()
    ap.add_argument("--input_field", required=True, help="input convex displacement field (.nii.gz) full resolution")
    ap.add_argument("--input_moving", required=True, help="input moving scan (.nii.gz)")
    ap.add_argument("--output_warped", required=True, help="output warped scan (.nii.gz)")
    a = ap.parse_args(argv)
    moving = nifti_io.load_fdata(a.input_moving).astype("float32")
    disp = nifti_io.load_fdata(a.input_field).astype("float32")
    warped = apply_convex(disp=disp, moving=moving)
    nifti_io.save_image(warped, nifti_io.load_affine(a.input_moving), a.output_warped)
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
