"""CPU restatement (numpy, float32 unless noted) of the evaluation operators around the hot path -- SURVEY 8(f).1/(f).3.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/, never by the product.

Every function cites the reference lines it follows; all of them are pinned bit-for-bit (or to the stated tolerance)
against tests/golden/metrics.npz, which was captured from the reference itself (tests/golden/make_golden.py --metrics)."""
import numpy as np

from . import oracle as _o

f32 = np.float32


def _central(x, axis):
    """Conv3d with weights (-0.5, 0, 0.5) and zero padding along `axis` (hyper_util:92-100): 0.5*(x[i+1] - x[i-1]);
    exact in any evaluation order because the taps are powers of two."""
    nxt = np.zeros_like(x)
    prv = np.zeros_like(x)
    sl = [slice(None)] * x.ndim
    a, b = list(sl), list(sl)
    a[axis], b[axis] = slice(0, -1), slice(1, None)
    nxt[tuple(a)] = x[tuple(b)]
    prv[tuple(b)] = x[tuple(a)]
    return (f32(0.5) * nxt + f32(-0.5) * prv).astype(f32)


def jacobian_determinant_3d(flow, convert1=True):
    """hyper_util:86-108.  flow (3,H,W,D) float32 -> (H-4, W-4, D-4) float32."""
    flow = np.asarray(flow, f32)
    _, H, W, D = flow.shape
    pix = flow * (np.array([H - 1, W - 1, D - 1], f32) / f32(2)).reshape(3, 1, 1, 1) if convert1 else flow
    pix = pix.astype(f32)
    J = [[_central(pix[c], i) for c in range(3)] for i in range(3)]          # J[i][c] = d pix_c / d axis_i
    for i in range(3):
        J[i][i] = (J[i][i] + f32(1)).astype(f32)
    c = (slice(2, -2),) * 3
    J = [[J[i][k][c] for k in range(3)] for i in range(3)]
    t0 = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1])
    t1 = J[1][0] * (J[0][1] * J[2][2] - J[0][2] * J[2][1])
    t2 = J[2][0] * (J[0][1] * J[1][2] - J[0][2] * J[1][1])
    return ((t0 - t1) + t2).astype(f32)


def jacobian_stats(jac_det):
    """convex_run_withconfig.py:148-150: std (unbiased) of log(clamp(jac+3, 1e-9, 1e9)) and the folding fraction.
    The logarithm and the reduction order of torch are not restated: float64 accumulation, compare at rel 1e-5."""
    j = np.asarray(jac_det, f32)
    l = np.log(np.clip((j + f32(3)).astype(f32), f32(1e-9), f32(1e9)).astype(np.float64))
    return float(l.std(ddof=1)), float((j < 0).mean())


def warp_labels_nearest(seg, disp):
    """convex_run_withconfig.py:96,139,141: F.grid_sample(seg, grid0 + disp.permute(0,2,3,4,1).flip(-1).div(scale1),
    mode='nearest') with align_corners=False, zeros padding.  seg (H,W,D) float32, disp (3,H,W,D) float32 in voxels."""
    seg = np.asarray(seg, f32)
    disp = np.asarray(disp, f32)
    H, W, D = seg.shape
    sc = [f32(S - 1) / f32(2) for S in (H, W, D)]
    base = [_o.affine_base(S) for S in (H, W, D)]
    out = np.zeros_like(seg)
    idx = []
    ok = np.ones(seg.shape, bool)
    for a, S in enumerate((H, W, D)):
        shp = [1, 1, 1]
        shp[a] = S
        g = (base[a].reshape(shp) + (disp[a] / sc[a]).astype(f32)).astype(f32)
        pos = (((g + f32(1)) * f32(S)).astype(f32) - f32(1)).astype(f32) / f32(2)       # unnormalize, align_corners=False
        r = np.rint(pos.astype(f32))                                                       # std::nearbyint: half to even
        ok &= (r >= 0) & (r <= S - 1)
        idx.append(np.clip(r, 0, S - 1).astype(np.int64))
    out[ok] = seg[idx[0][ok], idx[1][ok], idx[2][ok]]
    return out


def dice_coeff(outputs, labels, max_label):
    """hyper_util:53-60.  float32 means of 0/1 indicators (= exact counts / N while counts < 2^24)."""
    o = np.asarray(outputs).reshape(-1)
    t = np.asarray(labels).reshape(-1)
    n = f32(o.size)
    dice = np.zeros(max_label - 1, f32)
    for lab in range(1, max_label):
        i, j = (o == lab), (t == lab)
        inter = f32(np.count_nonzero(i & j)) / n
        dice[lab - 1] = (f32(2.0) * inter) / ((f32(1e-8) + f32(np.count_nonzero(i)) / n) + f32(np.count_nonzero(j)) / n)
    return dice


def sample_field_at_points(field, pts):
    """convex_run_paired_mind.py:167-168: grid_sample(disp_hr, (key.flip(1)/scale1 - 1)) (bilinear, zeros,
    align_corners=False).  field (3,H,W,D), pts (n,3) in voxel coordinates (H,W,D order) -> (n,3)."""
    field = np.asarray(field, f32)
    pts = np.asarray(pts, f32)
    _, H, W, D = field.shape
    scale1 = np.array([D - 1, W - 1, H - 1], f32) / f32(2)
    g = ((pts[:, ::-1] / scale1).astype(f32) - f32(1)).astype(f32)
    return _o.grid_sample(field, g.reshape(-1, 1, 1, 3)).reshape(3, -1).T.copy()


def tre(key_fixed, key_moving, disp_sampled):
    """convex_run_paired_mind.py:173."""
    d = (np.asarray(key_fixed, f32) - np.asarray(key_moving, f32) + np.asarray(disp_sampled, f32)).astype(f32)
    sq = (d * d).astype(f32)
    s = ((sq[:, 0] + sq[:, 1]).astype(f32) + sq[:, 2]).astype(f32)
    return np.sqrt(s).astype(f32)


def _fma(a, b, c):
    """float32 fma: the exact product of two float32 values fits a float64 (24 + 24 bits) and the sum is rounded once
    to float64 before the final float32 rounding; double rounding cannot occur for |terms| in this module's range
    (checked against torch.linspace in tests/test_oracle_vs_golden.py)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def linspace(start, end, steps):
    """torch.linspace on CPU (float32): step = (end - start) / (steps - 1); first half fma(step, i, start), second half
    fma(-step, steps - 1 - i, end) (RangeFactories kernel; same form as oracle/cvx_oracle.c::orc_linspace_pm1)."""
    start, end = f32(start), f32(end)
    step = f32((end - start) / f32(steps - 1))
    i = np.arange(steps)
    lo = _fma(step, i.astype(f32), start)
    hi = _fma(-step, (steps - 1 - i).astype(f32), end)
    return np.where(i < steps // 2, lo, hi).astype(f32)


def sort_rank(value):
    """hyper_util:28-31: rank1[value.sort().indices] = linspace(1, .1, n)."""
    v = np.asarray(value, f32)
    r = np.ones_like(v)
    r[np.argsort(v, kind="stable")] = linspace(1.0, 0.1, v.size)
    return r


def apply_convex(disp, moving):
    """apply_convex.py:13-24: scipy.ndimage.map_coordinates(moving, disp + identity, order=1) (mode 'constant', cval 0)
    restated: float64 arithmetic; a sample whose coordinate leaves [0, n-1] along any axis is 0; otherwise
    sum over the 8 taps (axis 0 slowest) of ((value * w0) * w1) * w2 with t = c - floor(c), w = (1 - t, 1 - (1 - t)):
    scipy's spline weights are completed so that they sum to exactly one (ni_splines.c), which differs from t in the
    last bit for some t."""
    disp = np.asarray(disp, np.float64)
    mov = np.asarray(moving)
    H, W, D = mov.shape
    idn = np.meshgrid(np.arange(H), np.arange(W), np.arange(D), indexing="ij")
    c = [disp[..., a] + idn[a] for a in range(3)]
    inside = np.ones(mov.shape, bool)
    lo, t = [], []
    for a, S in enumerate((H, W, D)):
        inside &= (c[a] >= 0) & (c[a] <= S - 1)
        f = np.floor(c[a])
        lo.append(np.clip(f, 0, S - 1).astype(np.int64))
        t.append(c[a] - f)
    out = np.zeros(mov.shape, np.float64)
    m64 = mov.astype(np.float64)
    for i in (0, 1):
        wi = 1.0 - (1.0 - t[0]) if i else 1.0 - t[0]
        zi = np.minimum(lo[0] + i, H - 1)
        for j in (0, 1):
            wj = 1.0 - (1.0 - t[1]) if j else 1.0 - t[1]
            yj = np.minimum(lo[1] + j, W - 1)
            for k in (0, 1):
                wk = 1.0 - (1.0 - t[2]) if k else 1.0 - t[2]
                xk = np.minimum(lo[2] + k, D - 1)
                out += ((m64[zi, yj, xk] * wi) * wj) * wk
    out[~inside] = 0.0
    return out


def hd95(fixed, moving, num_labels, precision=1):
    """cupy_hd95 (self_configuring/convexAdam_hyper_util.py:32-51) with numpy/scipy in the place of cupy/cupyx (absent here; same
    definitions: float32 Euclidean distances, linear-interpolation percentile).  fixed, moving: (H,W,D) integer label maps."""
    from scipy.ndimage import distance_transform_edt
    fixed, moving = np.asarray(fixed).astype(np.int64), np.asarray(moving).astype(np.int64)
    prec = float(precision)

    def up(mask):
        """upsample_nearest3d with scale_factor `precision` (F.interpolate, :33-34): extent (int64)(n * s); source index by ATen's
        nearest_idx (UpSample.h): dst when the extent is unchanged, dst >> 1 when it doubles, else min(floor(dst * float32(1 / s)), n - 1)
        evaluated in float32."""
        idx = []
        for n in mask.shape:
            no = int(n * prec)
            dst = np.arange(no)
            if no == n:
                idx.append(dst)
            elif no == 2 * n:
                idx.append(dst >> 1)
            else:
                idx.append(np.minimum(np.floor(dst.astype(np.float32) * np.float32(1.0 / prec)).astype(np.int64), n - 1))
        return mask[np.ix_(*idx)]

    def edt32(m):
        return distance_transform_edt(m).astype(np.float32)

    out = np.zeros(int(num_labels), np.float64)
    for i in range(int(num_labels)):
        f, m = up((fixed == i + 1).astype(np.uint8)), up((moving == i + 1).astype(np.uint8))
        if f.sum() > 0 and m.sum() > 0:
            d1 = edt32(f); s1 = d1 == 1; d1 = d1 + edt32(1 - f)
            d2 = edt32(m); s2 = d2 == 1; d2 = d2 + edt32(1 - m)
            out[i] = np.maximum(np.percentile(d1[s2], 95), np.percentile(d2[s1], 95))
        else:
            out[i] = 30
    return out * 1 / precision

