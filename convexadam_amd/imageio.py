"""Images with geometry and the MetaImage (.mha / .mhd) file format, without SimpleITK.

The reference reads and writes its volumes through SimpleITK (`sitk.ReadImage` / `WriteImage`, src/convexAdam/convex_adam_translation.py:
117-146; its tests read `.mha`, tests/test_convex_adam_mind.py) and does its geometry -- resampling to 1 mm, onto another grid, carrying
a displacement field to the moving image's frame -- with SimpleITK's resampler (convex_adam_utils.py:282-351, apply_convex.py:27-78).
SimpleITK is an optional dependency here: when it is installed the helpers of convex_adam_utils use it exactly as the reference does;
when it is not, they work on the `Image` class below, which carries the same geometry under the same accessor names
(GetSize / GetSpacing / GetOrigin / GetDirection / SetOrigin / CopyInformation, x-y-z order; array in z-y-x order like
sitk.GetArrayFromImage) and a linear resampler with ITK's conventions:

    physical point of index i:   p = origin + Direction . (spacing * i)
    resample(src -> grid):        value(i) = trilinear interpolation of src at the continuous index of p(i), 0 outside the buffer
                                  (ITK: inside means -0.5 <= index <= size - 0.5 per axis; neighbours beyond the edge are clamped)

MetaImage: text header (`Key = Value` lines, `ElementDataFile = LOCAL` last) followed by the raw voxels, x fastest; both byte orders,
optional zlib compression (`CompressedData = True`), scalar element types; `TransformMatrix`, `Offset`, `ElementSpacing` hold direction
cosines, origin and spacing.  The interpolation itself is scipy.ndimage.map_coordinates(order=1) on the host -- geometry glue, not the
hot path.  Border handling of the resampler is a restatement of ITK's documented behaviour and is NOT pinned against SimpleITK in this
image (SimpleITK is absent); interior voxels are pinned by the analytic tests in tests/test_host_logic.py.
"""
import os
import zlib

import numpy as np

_MET = {"MET_CHAR": np.int8, "MET_UCHAR": np.uint8, "MET_SHORT": np.int16, "MET_USHORT": np.uint16, "MET_INT": np.int32,
        "MET_UINT": np.uint32, "MET_LONG": np.int32, "MET_ULONG": np.uint32, "MET_LONG_LONG": np.int64, "MET_ULONG_LONG": np.uint64,
        "MET_FLOAT": np.float32, "MET_DOUBLE": np.float64}
_MET_OF = {np.dtype(np.int8): "MET_CHAR", np.dtype(np.uint8): "MET_UCHAR", np.dtype(np.int16): "MET_SHORT", np.dtype(np.uint16): "MET_USHORT",
           np.dtype(np.int32): "MET_INT", np.dtype(np.uint32): "MET_UINT", np.dtype(np.int64): "MET_LONG_LONG",
           np.dtype(np.uint64): "MET_ULONG_LONG", np.dtype(np.float32): "MET_FLOAT", np.dtype(np.float64): "MET_DOUBLE"}


class Image:
    """3-D scalar image: `array` (z, y, x) + spacing / origin (x, y, z) + direction cosines (9 values, row-major)."""

    def __init__(self, array, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
        self.array = np.ascontiguousarray(array)
        if self.array.ndim != 3:
            raise ValueError("Image expects a 3-D array (z, y, x)")
        self._spacing = tuple(float(v) for v in spacing)
        self._origin = tuple(float(v) for v in origin)
        self._direction = tuple(float(v) for v in direction)

    # SimpleITK-compatible accessors ------------------------------------------------------------------------------------------
    def GetSize(self):
        return tuple(int(v) for v in self.array.shape[::-1])

    def GetSpacing(self):
        return self._spacing

    def GetOrigin(self):
        return self._origin

    def GetDirection(self):
        return self._direction

    def SetOrigin(self, origin):
        self._origin = tuple(float(v) for v in origin)

    def SetSpacing(self, spacing):
        self._spacing = tuple(float(v) for v in spacing)

    def SetDirection(self, direction):
        self._direction = tuple(float(v) for v in direction)

    def CopyInformation(self, other):
        if tuple(other.GetSize()) != self.GetSize():
            raise ValueError("CopyInformation: sizes differ %s vs %s" % (other.GetSize(), self.GetSize()))
        self._spacing, self._origin, self._direction = tuple(other.GetSpacing()), tuple(other.GetOrigin()), tuple(other.GetDirection())

    def copy(self):
        return Image(self.array.copy(), self._spacing, self._origin, self._direction)

    # geometry -------------------------------------------------------------------------------------------------------------------
    def index_to_physical_matrix(self):
        """(A, o) with p = A @ index_xyz + o."""
        D = np.array(self._direction, dtype=np.float64).reshape(3, 3)
        return D * np.array(self._spacing, dtype=np.float64)[None, :], np.array(self._origin, dtype=np.float64)


def get_array(img):
    """(z, y, x) array of an `Image` or a SimpleITK image."""
    if isinstance(img, Image):
        return img.array
    import SimpleITK as sitk  # noqa: N813
    return sitk.GetArrayFromImage(img)


def resample(src, spacing, size, direction, origin, default=0.0):
    """Linear resampling of `src` onto the grid (size, spacing, direction, origin), identity transform, `default` outside
    (sitk.ResampleImageFilter with sitkLinear as the reference configures it, convex_adam_utils.py:282-306)."""
    from scipy.ndimage import map_coordinates
    out = Image(np.zeros(tuple(int(v) for v in size)[::-1], np.float64), spacing, origin, direction)
    Ao, oo = out.index_to_physical_matrix()
    As, os_ = src.index_to_physical_matrix()
    M = np.linalg.solve(As, Ao)                         # source index = M @ out index + t
    t = np.linalg.solve(As, oo - os_)
    nx, ny, nz = out.GetSize()
    k, j, i = np.meshgrid(np.arange(nz, dtype=np.float64), np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")
    idx = np.stack([i, j, k], 0).reshape(3, -1)          # x, y, z
    ci = M @ idx + t[:, None]
    sx, sy, sz = src.GetSize()
    lim = np.array([sx, sy, sz], np.float64)[:, None]
    inside = np.all((ci >= -0.5) & (ci <= lim - 0.5), axis=0)
    cic = np.clip(ci, 0.0, lim - 1.0)                    # neighbours beyond the edge are clamped
    vals = map_coordinates(np.asarray(src.array, np.float64), cic[::-1], order=1, mode="nearest")
    vals = np.where(inside, vals, float(default))
    arr = vals.reshape(nz, ny, nx)
    if np.issubdtype(src.array.dtype, np.floating):
        arr = arr.astype(src.array.dtype)
    else:                                                # ITK casts the interpolated value back to the pixel type with rounding
        arr = np.rint(arr).astype(src.array.dtype)
    out.array = arr
    return out


# ---- MetaImage ---------------------------------------------------------------------------------------------------------------------
def read_mha(path):
    """Reads a 3-D scalar MetaImage (.mha, or .mhd with its raw file)."""
    with open(path, "rb") as f:
        raw = f.read()
    hdr, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("latin-1").strip()
        pos = end + 1
        if not line:
            continue
        key, _, val = line.partition("=")
        hdr[key.strip()] = val.strip()
        if key.strip() == "ElementDataFile":
            break
    if int(hdr.get("NDims", 3)) != 3:
        raise ValueError("read_mha: only 3-D images (NDims = %s)" % hdr.get("NDims"))
    if int(hdr.get("ElementNumberOfChannels", 1)) != 1:
        raise ValueError("read_mha: only scalar images")
    size = [int(v) for v in hdr["DimSize"].split()]
    dt = np.dtype(_MET[hdr["ElementType"]])
    msb = hdr.get("BinaryDataByteOrderMSB", hdr.get("ElementByteOrderMSB", "False")).lower() == "true"
    dt = dt.newbyteorder(">" if msb else "<")
    if hdr["ElementDataFile"] == "LOCAL":
        data = raw[pos:]
    else:
        with open(os.path.join(os.path.dirname(os.path.abspath(path)), hdr["ElementDataFile"]), "rb") as f:
            data = f.read()
    if hdr.get("CompressedData", "False").lower() == "true":
        data = zlib.decompress(data)
    n = size[0] * size[1] * size[2]
    arr = np.frombuffer(data, dtype=dt, count=n).reshape(size[2], size[1], size[0]).astype(dt.newbyteorder("="))
    spacing = [float(v) for v in hdr.get("ElementSpacing", hdr.get("ElementSize", "1 1 1")).split()]
    origin = [float(v) for v in hdr.get("Offset", hdr.get("Position", hdr.get("Origin", "0 0 0"))).split()]
    tm = [float(v) for v in hdr.get("TransformMatrix", hdr.get("Rotation", hdr.get("Orientation", "1 0 0 0 1 0 0 0 1"))).split()]
    # MetaIO stores the matrix column-major with respect to ITK's direction (each group of three values is one axis direction)
    direction = np.array(tm, np.float64).reshape(3, 3).T.reshape(-1)
    return Image(arr, spacing, origin, direction)


def write_mha(img, path, compress=False):
    """Writes an `Image` (or anything with the same accessors and a z-y-x array) as a single-file MetaImage."""
    arr = np.ascontiguousarray(get_array(img))
    if arr.dtype not in _MET_OF:
        arr = arr.astype(np.float32)
    arr = arr.astype(arr.dtype.newbyteorder("<"))
    nx, ny, nz = arr.shape[2], arr.shape[1], arr.shape[0]
    D = np.array(img.GetDirection(), np.float64).reshape(3, 3).T.reshape(-1)
    data = arr.tobytes()
    lines = ["ObjectType = Image", "NDims = 3", "BinaryData = True", "BinaryDataByteOrderMSB = False",
             "CompressedData = %s" % ("True" if compress else "False")]
    if compress:
        data = zlib.compress(data)
        lines.append("CompressedDataSize = %d" % len(data))
    fmt = lambda v: " ".join(repr(float(x)) if float(x) != int(x) else str(int(x)) for x in v)   # noqa: E731
    lines += ["TransformMatrix = " + fmt(D), "Offset = " + fmt(img.GetOrigin()), "CenterOfRotation = 0 0 0", "AnatomicalOrientation = RAI",
              "ElementSpacing = " + fmt(img.GetSpacing()), "DimSize = %d %d %d" % (nx, ny, nz), "ElementType = " + _MET_OF[np.dtype(arr.dtype.newbyteorder("="))],
              "ElementDataFile = LOCAL"]
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("latin-1"))
        f.write(data)


def read_image(path):
    """.mha / .mhd through this module, .nii / .nii.gz through nifti_io (affine -> spacing / origin / direction in ITK's LPS frame)."""
    p = str(path)
    if p.endswith((".mha", ".mhd")):
        return read_mha(p)
    from . import nifti_io
    data = nifti_io.load_fdata(p)
    aff = np.array(nifti_io.load_affine(p), np.float64)
    lps = np.diag([-1.0, -1.0, 1.0, 1.0]) @ aff                                    # NIfTI is RAS, ITK is LPS
    spacing = np.linalg.norm(lps[:3, :3], axis=0)
    direction = lps[:3, :3] / spacing[None, :]
    return Image(np.ascontiguousarray(np.transpose(data, (2, 1, 0))), spacing, lps[:3, 3], direction.reshape(-1))
