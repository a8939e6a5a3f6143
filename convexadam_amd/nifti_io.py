"""Minimal NIfTI-1 single-file reader / writer ("NIfTI writers without nibabel").

The file wrappers of the reference (`convex_adam`, convex_adam_MIND.py:205-248; `convex_adam_nnUNet.convex_adam`) read two
volumes with `nib.load(path).get_fdata()` and write `disp.nii.gz` as `nib.Nifti1Image(disp, affine)`.  nibabel is used when it is
installed; this module covers hosts without it with the subset those calls need:

    load(path)  -> (data float64 in nibabel's axis order, affine 4x4 float64)      .nii / .nii.gz, either endianness, scl_slope/inter
    save(data, affine, path)                                                      float32/float64/int/uint arrays, sform = affine

Layout written: 348-byte header, vox_offset 352, sform_code 2 ("aligned", what nibabel stores for an affine given at construction),
qform from the same affine when it is a rotation + positive scaling, pixdim = column norms.
"""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32,
           1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def load(path):
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError("%s: too short for a NIfTI-1 header" % path)
    end = "<" if struct.unpack("<i", raw[0:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[0:4])[0] != 348 or raw[344:347] != b"n+1":
        raise ValueError("%s: not a single-file NIfTI-1 image" % path)
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(end + "2h", raw[70:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "2h", raw[252:256])
    if datatype not in _DTYPES:
        raise ValueError("%s: NIfTI datatype %d is not supported" % (path, datatype))
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dt, n, int(vox_offset)).reshape(shape, order="F").astype(np.float64)
    if slope not in (0.0,) and not np.isnan(slope) and (slope != 1.0 or inter != 0.0):
        data = data * float(slope) + float(inter)                  # get_fdata() applies the scaling
    if sform_code > 0:
        affine = np.eye(4)
        affine[0], affine[1], affine[2] = (struct.unpack(end + "4f", raw[o:o + 16]) for o in (280, 296, 312))
    elif qform_code > 0:
        b, c, d, qx, qy, qz = struct.unpack(end + "6f", raw[256:280])
        a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        affine = np.eye(4)
        affine[:3, :3] = R * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac])
        affine[:3, 3] = (qx, qy, qz)
    else:
        affine = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0])
    return np.ascontiguousarray(data), affine.astype(np.float64)


def _quaternion(affine):
    """(b, c, d, qfac, zooms) of the rotation part of `affine`, or None if it is not rotation x positive scaling."""
    M = np.asarray(affine, np.float64)[:3, :3]
    zooms = np.sqrt((M * M).sum(0))
    if np.any(zooms == 0):
        return None
    R = M / zooms
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R = R.copy(); R[:, 2] = -R[:, 2]; qfac = -1.0
    if not np.allclose(R @ R.T, np.eye(3), atol=1e-4):
        return None
    tr = 1.0 + R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0.5:
        a = 0.5 * np.sqrt(tr)
        b, c, d = 0.25 * (R[2, 1] - R[1, 2]) / a, 0.25 * (R[0, 2] - R[2, 0]) / a, 0.25 * (R[1, 0] - R[0, 1]) / a
    else:
        xd, yd, zd = 1.0 + R[0, 0] - (R[1, 1] + R[2, 2]), 1.0 + R[1, 1] - (R[0, 0] + R[2, 2]), 1.0 + R[2, 2] - (R[0, 0] + R[1, 1])
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd); c = 0.25 * (R[0, 1] + R[1, 0]) / b; d = 0.25 * (R[0, 2] + R[2, 0]) / b; a = 0.25 * (R[2, 1] - R[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd); b = 0.25 * (R[0, 1] + R[1, 0]) / c; d = 0.25 * (R[1, 2] + R[2, 1]) / c; a = 0.25 * (R[0, 2] - R[2, 0]) / c
        else:
            d = 0.5 * np.sqrt(zd); b = 0.25 * (R[0, 2] + R[2, 0]) / d; c = 0.25 * (R[1, 2] + R[2, 1]) / d; a = 0.25 * (R[1, 0] - R[0, 1]) / d
        if a < 0:
            b, c, d = -b, -c, -d
    return float(b), float(c), float(d), qfac, zooms


def save(data, affine, path):
    arr = np.asarray(data)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype not in _CODES:
        arr = arr.astype(np.float64)
    if not 1 <= arr.ndim <= 7:
        raise ValueError("NIfTI-1 images have 1 to 7 dimensions")
    affine = np.asarray(affine, np.float64)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<2h", hdr, 70, _CODES[arr.dtype], arr.dtype.itemsize * 8)
    q = _quaternion(affine)
    zooms = q[4] if q else np.sqrt((affine[:3, :3] ** 2).sum(0))
    pixdim = [q[3] if q else 1.0] + [float(z) for z in zooms] + [1.0] * 4
    struct.pack_into("<8f", hdr, 76, *pixdim)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)             # vox_offset, scl_slope, scl_inter
    hdr[123] = 2                                                    # xyzt_units: millimetres
    struct.pack_into("<2h", hdr, 252, 2 if q else 0, 2)             # qform_code, sform_code ("aligned")
    if q:
        struct.pack_into("<6f", hdr, 256, q[0], q[1], q[2], *[float(v) for v in affine[:3, 3]])
    for row, off in zip(range(3), (280, 296, 312)):
        struct.pack_into("<4f", hdr, off, *[float(v) for v in affine[row]])
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")                                        # header extension flag, data starts at 352
        f.write(np.asfortranarray(arr).astype(arr.dtype.newbyteorder("<"), copy=False).tobytes(order="F"))


def load_fdata(path):
    """nib.load(path).get_fdata() -- through nibabel when it is installed."""
    try:
        import nibabel as nib
    except ImportError:
        return load(path)[0]
    return nib.load(path).get_fdata()


def load_affine(path):
    try:
        import nibabel as nib
    except ImportError:
        return load(path)[1]
    return nib.load(path).affine


def save_image(data, affine, path):
    """nib.save(nib.Nifti1Image(data, affine), path) -- through nibabel when it is installed."""
    try:
        import nibabel as nib
    except ImportError:
        return save(data, affine, path)
    nib.save(nib.Nifti1Image(data, affine), path)
