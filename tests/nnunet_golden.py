"""Helpers for tests/golden/nnunet.npz (BASELINE configs[3] captured end to end from the reference by make_golden_nnunet.py)."""
import numpy as np


def features(g):
    """The reference's weighted one-hot features of tests/golden/nnunet.npz, rebuilt exactly from the label maps and the captured
    per-channel value 10 * w_c (convex_adam_nnUNet.py:35-36); channel order = ascending label present in either map (:26-30)."""
    lf, lm = g["lab_fix"].astype(np.float32), g["lab_mov"].astype(np.float32)
    present = np.union1d(np.unique(lf), np.unique(lm))
    assert present.size == int(g["n_ch"])
    ff = np.stack([(lf == c).astype(np.float32) * g["feat_max"][i] for i, c in enumerate(present)])
    fm = np.stack([(lm == c).astype(np.float32) * g["feat_max"][i] for i, c in enumerate(present)])
    assert np.allclose(ff.astype(np.float64).sum((1, 2, 3)), g["feat_fix_sum"], rtol=1e-12)
    assert np.allclose(fm.astype(np.float64).sum((1, 2, 3)), g["feat_mov_sum"], rtol=1e-12)
    return lf, lm, ff, fm


def field_checks(g, name, out, exact):
    """out (H,W,D,3) against the captured field `name`: bit for bit (full or every second voxel + float64 sums of the whole field)
    when `exact`, else by mean end-point error."""
    ref = g[name]
    sub = out if ref.shape == out.shape else out[::2, ::2, ::2]
    if exact:
        assert np.array_equal(sub.astype(np.float32), ref), "%s: max |diff| %g" % (name, np.abs(sub - ref).max())
        o = np.ascontiguousarray(out, dtype=np.float64)            # (same summation order as the capture: numpy sums pairwise in memory order)
        assert np.allclose(o.sum((0, 1, 2)), g[name + "_sum"], rtol=1e-14, atol=0) and np.allclose((o * o).sum((0, 1, 2)), g[name + "_sumsq"], rtol=1e-14, atol=0)
    return float(np.sqrt(((sub.astype(np.float64) - ref.astype(np.float64)) ** 2).sum(-1)).mean())
