"""Translation-only registration wrapper around convex_adam_pt (reference: src/convexAdam/convex_adam_translation.py:12-131).

Host-side geometry on SimpleITK images; the registration itself runs through convexadam_amd.convex_adam_MIND.convex_adam_pt
(HIP).  SimpleITK is imported when a function that needs it is called, so the module also imports on hosts without it.

    index_translation_to_world_translation(index_translation, direction)        :12-29
    apply_translation(moving_image, translation_ijk)                            :32-54
    convex_adam_translation(fixed_image, moving_image, segmentation=None, co_moving_images=None)    :57-114
    convex_adam_translation_from_file(...)                                      :117-146
"""
import numpy as np

from .convex_adam_MIND import convex_adam_pt
from .convex_adam_utils import _sitk, resample_img, resample_moving_to_fixed


def index_translation_to_world_translation(index_translation, direction):
    """Translation along the image axes (i, j, k; mm) -> world axes (x, y, z; mm): direction-cosine matrix times the vector."""
    n = int(np.sqrt(len(direction)))
    return np.array(direction).reshape((n, n)) @ np.array(index_translation)


def apply_translation(moving_image, translation_ijk=(0, 0, 0)):
    """Copy of `moving_image` whose origin is shifted by the world-space equivalent of `translation_ijk` (mm along the image axes)."""
    sitk = _sitk()
    moved = sitk.Image(moving_image)
    shift = index_translation_to_world_translation(translation_ijk, moved.GetDirection()[0:9])
    origin = np.array(moved.GetOrigin(), dtype=float)
    origin[0:3] -= shift
    moved.SetOrigin(tuple(origin))
    return moved


def field_to_translation(displacement_field, spacing_xyz, mask=None):
    """Mean displacement (over `mask` if given) of a (H,W,D,3) field in voxels of a 1 mm grid -> whole-voxel translation of an image
    with spacing `spacing_xyz`, returned in mm as (x, y, z)   (:88-103)."""
    field = np.asarray(displacement_field)
    mean_zyx = np.mean(field[mask], axis=0) if mask is not None else np.mean(field, axis=(0, 1, 2))
    spacing_zyx = np.array(list(spacing_xyz)[::-1])
    voxels = np.round(mean_zyx / spacing_zyx, decimals=0)
    return tuple(list((voxels * spacing_zyx)[::-1]))


def convex_adam_translation(fixed_image, moving_image, segmentation=None, co_moving_images=None):
    """Register `moving_image` to `fixed_image` with convex_adam_pt on a 1 mm grid, reduce the field to one whole-voxel translation
    (mean over the segmentation if given) and apply it to the moving image and to the co-moving images.
    Returns (translation_xyz in mm, moved image, moved co-moving images)."""
    sitk = _sitk()
    fixed_1mm = resample_img(fixed_image, spacing=(1.0, 1.0, 1.0))
    moving_1mm = resample_moving_to_fixed(fixed_1mm, moving_image)
    field = convex_adam_pt(img_fixed=fixed_1mm, img_moving=moving_1mm)
    mask = None
    if segmentation is not None:
        # linear resampling blurs the labels: everything above zero counts
        mask = sitk.GetArrayFromImage(resample_moving_to_fixed(moving=segmentation, fixed=fixed_1mm)) > 0
    translation_xyz = field_to_translation(field, moving_image.GetSpacing(), mask)
    moved = apply_translation(moving_image=moving_image, translation_ijk=translation_xyz)
    if co_moving_images is not None:
        for i, image in enumerate(co_moving_images):
            co_moving_images[i] = apply_translation(moving_image=image, translation_ijk=translation_xyz)
    return translation_xyz, moved, co_moving_images


def convex_adam_translation_from_file(fixed_path="/input/fixed.mha", moving_path="/input/moving.mha",
                                      segmentation_path="/input/segmentation.nii.gz", moving_output_path="/output/moving_warped.mha",
                                      co_moving_paths=None, co_moving_output_paths=None):
    sitk = _sitk()
    co = [sitk.ReadImage(str(p)) for p in co_moving_paths] if co_moving_paths is not None else None
    translation_xyz, moved, co = convex_adam_translation(
        fixed_image=sitk.ReadImage(str(fixed_path)), moving_image=sitk.ReadImage(str(moving_path)),
        segmentation=sitk.ReadImage(str(segmentation_path)) if segmentation_path is not None else None, co_moving_images=co)
    sitk.WriteImage(moved, str(moving_output_path))
    if co is not None:
        for image, path in zip(co, co_moving_output_paths):
            sitk.WriteImage(image, str(path))
    return translation_xyz
