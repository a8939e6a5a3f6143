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
from .convex_adam_utils import _is_builtin, _sitk, resample_img, resample_moving_to_fixed


def index_translation_to_world_translation(index_translation, direction):
    """Translation along the image axes (i, j, k; mm) -> world axes (x, y, z; mm): direction-cosine matrix times the vector."""
    n = int(np.sqrt(len(direction)))
    return np.array(direction).reshape((n, n)) @ np.array(index_translation)


def apply_translation(moving_image, translation_ijk=(0, 0, 0)):
    """Copy of `moving_image` whose origin is shifted by the world-space equivalent of `translation_ijk` (mm along the image axes)."""
    moved = moving_image.copy() if _is_builtin(moving_image) else _sitk().Image(moving_image)
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
    fixed_1mm = resample_img(fixed_image, spacing=(1.0, 1.0, 1.0))
    moving_1mm = resample_moving_to_fixed(fixed_1mm, moving_image)
    field = convex_adam_pt(img_fixed=fixed_1mm, img_moving=moving_1mm)
    mask = None
    if segmentation is not None:
        # linear resampling blurs the labels: everything above zero counts
        from .imageio import get_array
        mask = get_array(resample_moving_to_fixed(moving=segmentation, fixed=fixed_1mm)) > 0
    translation_xyz = field_to_translation(field, moving_image.GetSpacing(), mask)
    moved = apply_translation(moving_image=moving_image, translation_ijk=translation_xyz)
    if co_moving_images is not None:
        for i, image in enumerate(co_moving_images):
            co_moving_images[i] = apply_translation(moving_image=image, translation_ijk=translation_xyz)
    return translation_xyz, moved, co_moving_images


def convex_adam_translation_from_file(fixed_path="/input/fixed.mha", moving_path="/input/moving.mha",
                                      segmentation_path="/input/segmentation.nii.gz", moving_output_path="/output/moving_warped.mha",
                                      co_moving_paths=None, co_moving_output_paths=None):
    """File front end (:117-146).  SimpleITK reads and writes when it is installed; otherwise the built-in MetaImage / NIfTI readers
    and the MetaImage writer of convexadam_amd.imageio do."""
    try:
        import SimpleITK as sitk  # noqa: N813
        read, write = (lambda p: sitk.ReadImage(str(p))), (lambda img, p: sitk.WriteImage(img, str(p)))
    except ImportError:
        from .imageio import read_image, write_mha
        read, write = (lambda p: read_image(str(p))), (lambda img, p: write_mha(img, str(p)))
    co = [read(p) for p in co_moving_paths] if co_moving_paths is not None else None
    translation_xyz, moved, co = convex_adam_translation(
        fixed_image=read(fixed_path), moving_image=read(moving_path),
        segmentation=read(segmentation_path) if segmentation_path is not None else None, co_moving_images=co)
    write(moved, moving_output_path)
    if co is not None:
        for image, path in zip(co, co_moving_output_paths):
            write(image, path)
    return translation_xyz


def main(argv=None):
    """python -m convexAdam.convex_adam_translation (convex_adam_translation.py:149-166)."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixed_path", default="/input/fixed.mha")
    ap.add_argument("--moving_path", default="/input/moving.mha")
    ap.add_argument("--segmentation_path", default=None)
    ap.add_argument("--moving_output_path", default="/output/moving_warped.mha")
    ap.add_argument("--co_moving_paths", nargs="+", default=None)
    ap.add_argument("--co_moving_output_paths", nargs="+", default=None)
    a = ap.parse_args(argv)
    print(convex_adam_translation_from_file(a.fixed_path, a.moving_path, a.segmentation_path, a.moving_output_path, a.co_moving_paths,
                                            a.co_moving_output_paths))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
