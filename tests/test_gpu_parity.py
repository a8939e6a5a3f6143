"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the C ABI, against
  (1) the CPU oracle on the same seeded inputs -- BIT-EXACT (np.array_equal) for every operator, because the
      kernels follow the oracle's / ATen's evaluation order and share the exp() restatement;
  (2) the golden vectors captured from the upstream reference (tests/golden), with the tolerances of
      tests/test_oracle_vs_golden.py (bit-exact except MKL's exp in MIND and sqrt in Adam);
  (3) size-independent properties at BASELINE.json's full sizes (identity -> 0, translation recovery,
      argmin consistency, inverse-consistency residual).
Nothing here reads /root/reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def epe(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(-1)).mean())


@pytest.fixture(scope="module")
def U():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from convexadam_amd import convex_adam_utils
    return convex_adam_utils


@pytest.fixture(scope="module")
def M():
    from convexadam_amd import convex_adam_MIND
    return convex_adam_MIND


def test_library_sees_the_gpu():
    from convexadam_amd import _lib
    assert _lib.lib().cvx_device_count() >= 1


# ---- (1) HIP vs oracle, bit-exact --------------------------------------------------------------------
@pytest.mark.parametrize("shape,r,d", [((20, 18, 23), 1, 2), ((33, 40, 70), 1, 2), ((20, 18, 23), 2, 2), ((17, 9, 66), 1, 1),
                                       ((12, 13, 14), 3, 3), ((9, 70, 8), 2, 1), ((10, 11, 70), 3, 4), ((6, 20, 33), 1, 4)])
def test_mindssc_vs_oracle(U, orc, shape, r, d):
    """(radius 3 with dilation 4 needs the 32-column tile of the tiled stencil: its image tile with a 7-voxel halo does not fit
    the LDS at 64 columns)"""
    from convexadam_amd.phantom import phantom
    img = phantom(shape, 3, 30)
    out = host(U.MINDSSC(img[None, None].to(DEV), r, d, device=DEV))[0]
    ref = orc.mindssc(img.numpy(), r, d)
    assert np.array_equal(out, ref), "max |diff| %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("shape", [(1, 1, 4), (2, 3, 4), (3, 9, 8), (5, 8, 64), (7, 17, 68), (21, 9, 132), (40, 24, 64), (45, 30, 128),
                                   (9, 8, 192), (16, 1, 60)])
def test_mindssc_marching_kernel_shapes(U, orc, shape):
    """r = 1, d = 2 with rows of a multiple of 4 voxels take the z-marching stencil (mindmarch.hip): single planes, tiles that
    overhang in y and x, several z chunks with a partial last one, the last < 32 voxels of the volume (interleaved channel sum)."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[2])
    img = (rng.standard_normal(shape) * 10).astype(np.float32)
    out = host(U.MINDSSC(dev(img)[None, None], 1, 2, device=DEV))[0]
    ref = orc.mindssc(img, 1, 2)
    assert np.array_equal(out, ref), "max |diff| %g" % np.abs(out - ref).max()


def test_mindssc_flat_and_clamped_regions(U, orc):
    """Zero background (mind_var = 0 -> clamped to 0.001*mean) and a bright blob: exercises both clamp bounds."""
    img = np.zeros((24, 20, 28), np.float32)
    img[6:18, 5:15, 8:20] = np.random.default_rng(0).normal(100, 30, (12, 10, 12)).astype(np.float32)
    out = host(U.MINDSSC(dev(img)[None, None], 1, 2, device=DEV))[0]
    assert np.array_equal(out, orc.mindssc(img, 1, 2))
    assert np.all(out[:, 0, 0, 0] == 1.0)


@pytest.mark.parametrize("g", [2, 3, 6])
def test_avgpool_vs_oracle(U, orc, g):
    x = np.random.default_rng(g).standard_normal((5, 13, 14, 15)).astype(np.float32)
    assert np.array_equal(host(U.avg_pool(dev(x)[None], g))[0], orc.avgpool_stride(x, g))


@pytest.mark.parametrize("g,shape", [(2, (3, 12, 12, 16)), (4, (3, 12, 8, 16)), (6, (2, 12, 18, 24)), (8, (2, 16, 8, 24)), (6, (1, 13, 12, 14))])
def test_avgpool_even_windows_vs_oracle(U, orc, g, shape):
    """Even D selects the compile-time-window kernels (8-byte row loads); remainders of H, W, D are ignored."""
    x = np.random.default_rng(g + shape[1]).standard_normal(shape).astype(np.float32)
    assert np.array_equal(host(U.avg_pool(dev(x)[None], g))[0], orc.avgpool_stride(x, g))


@pytest.mark.parametrize("C,shape,hw", [(3, (5, 60, 37), 1), (2, (4, 30, 61), 1), (2, (3, 70, 9), 2)])
def test_correlate_tiled_rows_vs_oracle(U, orc, C, shape, hw):
    """Planes too large for one workgroup of the marching box kernel: y tiles with recomputed halo rows."""
    rng = np.random.default_rng(C * 10 + hw)
    f = rng.random((C,) + shape, dtype=np.float32)
    m = rng.random((C,) + shape, dtype=np.float32)
    ssd, am = U.correlate(dev(f)[None], dev(m)[None], hw, 1, shape, C)
    rs, ra = orc.correlate(f, m, hw)
    assert np.array_equal(host(ssd), rs), "max |diff| %g" % np.abs(host(ssd) - rs).max()
    assert np.array_equal(host(am), ra)


@pytest.mark.parametrize("C,shape,hw", [(12, (12, 10, 14), 2), (12, (7, 9, 11), 3), (12, (9, 8, 37), 4), (20, (7, 5, 9), 1),
                                        (3, (5, 6, 7), 2), (33, (6, 5, 8), 2), (12, (4, 4, 4), 6), (1, (3, 3, 3), 0),
                                        # every search width of the pipeline at realistic coarse shapes (BASELINE configs 1-3: rows of 37)
                                        (12, (9, 8, 37), 5), (12, (9, 8, 37), 6), (12, (9, 8, 37), 7), (12, (9, 8, 37), 8),
                                        (12, (13, 16, 20), 6), (12, (13, 16, 20), 8), (32, (13, 16, 20), 5), (32, (9, 8, 37), 7),
                                        (14, (26, 32, 37), 6), (5, (11, 12, 13), 8),
                                        # planes taller than one role of the fused kernel holds -> y tiles with recomputed halo rows (the coarse
                                        # grids of the sweep's grid_sp 2..5 at 160 x 192 x 224 have 38 .. 96 rows of 44 .. 112 voxels)
                                        (12, (6, 40, 37), 3), (12, (5, 23, 74), 2), (7, (4, 38, 44), 4), (12, (3, 96, 112), 1), (12, (9, 33, 37), 6),
                                        (32, (5, 48, 56), 2), (18, (6, 40, 37), 3), (64, (4, 9, 10), 2), (67, (3, 5, 6), 1),
                                        # the interleaved-order tail (last ncols mod 32 elements) spans several planes on tiny grids
                                        (12, (3, 3, 3), 0), (12, (5, 2, 3), 1), (20, (7, 1, 3), 0),
                                        # search half-widths beyond the reference's usual range (no limit there; 15 here)
                                        (12, (5, 6, 9), 9), (6, (4, 5, 13), 11), (20, (3, 4, 5), 10), (12, (4, 3, 6), 15)])
def test_correlate_vs_oracle(U, orc, C, shape, hw):
    """Includes C >= 16 (ATen cascade sum), ragged inner sizes (interleaved tail rule), D not a multiple of 4,
    search windows larger than the volume, the degenerate hw = 0, y-tiled planes and multi-plane tails: all through the fused kernel."""
    from convexadam_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(C * 100 + hw)
    f = rng.random((C,) + shape, dtype=np.float32)
    m = rng.random((C,) + shape, dtype=np.float32)
    rs, ra = orc.correlate(f, m, hw)
    # default path (C >= 16: the round-1 kernels, faster there) and the fused kernel forced for every C
    for fused_all in ((0, 1) if C >= 16 else (0,)):
        L.cvx_set_option(b"corr_fused_all", fused_all)
        try:
            if fused_all or C < 16:
                assert L.cvx_correlate_workspace_bytes(C, *shape, hw) < 4 * (2 * hw + 1) ** 3 * shape[0] * shape[1] * shape[2] + (1 << 22), \
                    "the fused kernel (no raw-SSD intermediate in the workspace) must cover this shape"
            ssd, am = U.correlate(dev(f)[None], dev(m)[None], hw, 1, shape, C)
        finally:
            L.cvx_set_option(b"corr_fused_all", 0)
        assert np.array_equal(host(ssd), rs), "max |diff| %g" % np.abs(host(ssd) - rs).max()
        assert np.array_equal(host(am), ra)


@pytest.mark.parametrize("case", [0, 3, 4])
def test_nan_in_the_cost_volume(U, orc, case):
    """Nulls in the features: NaNs propagate through the raw SSD and both boxes exactly as in the oracle (which equals the reference
    there, tests/test_oracle_vs_reference_live.py), and argmin follows torch: the FIRST NaN of a column wins -- plain argmin, the
    pruned coupled-convex passes (a voxel with a NaN keeps its winner) and the streaming ones (option no_prune).  (+-Inf voxels are
    outside the contract: the 3-instruction exact division of the box filters turns Inf / 27 into NaN; INTEGRATION.md.)"""
    from convexadam_amd import _lib
    rng = np.random.default_rng(case)
    shape, hw = (5, 6, 7), 2
    f = rng.random((12,) + shape, dtype=np.float32)
    m = rng.random((12,) + shape, dtype=np.float32)
    if case == 0:
        m[3, 2, 3, 4] = np.nan                                            # some displacements of the neighbouring voxels see the NaN
    elif case == 3:
        f[:, 2, 2, 2] = np.nan                                            # every displacement of the neighbourhood is NaN
    else:
        m[0, 0, 0, 0] = np.nan
        m[5, 4, 5, 6] = np.nan
        f[2, 1, 4, 3] = np.nan
    rs, ra = orc.correlate(f, m, hw)
    mesh = orc.disp_mesh(hw)
    want = orc.coupled_convex(rs, ra, mesh, hw)
    ssd, am = U.correlate(dev(f)[None], dev(m)[None], hw, 1, shape, 12)
    assert np.array_equal(host(ssd), rs, equal_nan=True) and np.array_equal(host(am), ra)
    for no_prune in (0, 1):
        _lib.lib().cvx_set_option(b"no_prune", no_prune)
        try:
            soft = U.coupled_convex(ssd, am, dev(mesh)[:, :, None], 1, shape)
        finally:
            _lib.lib().cvx_set_option(b"no_prune", 0)
        assert np.array_equal(host(soft)[0], want, equal_nan=True), no_prune
    # the caller's argmin only SEEDS the first smoothing step (convex_adam_utils.py:96); it need not be the first NaN of a NaN column:
    # every later pass recomputes the argmin and finds the first NaN (the pruned passes used to keep the caller's index)
    K = rs.shape[0]
    nan_cols = np.isnan(rs.reshape(K, -1)).any(0).reshape(shape)
    bad = ra.copy()
    bad[nan_cols] = (bad[nan_cols] + 7) % K
    assert nan_cols.any() and not np.array_equal(bad, ra)
    want_bad = orc.coupled_convex(rs, bad, mesh, hw)
    for no_prune in (0, 1):
        _lib.lib().cvx_set_option(b"no_prune", no_prune)
        try:
            soft = U.coupled_convex(ssd, dev(bad), dev(mesh)[:, :, None], 1, shape)
        finally:
            _lib.lib().cvx_set_option(b"no_prune", 0)
        assert np.array_equal(host(soft)[0], want_bad, equal_nan=True), ("caller argmin off the first NaN", no_prune)
    ssd16, am16 = U.correlate(dev(f)[None], dev(m)[None], hw, 1, shape, 12, storage="fp16")       # half-precision volume: same rule
    ref16 = rs.astype(np.float16)
    assert np.array_equal(host(ssd16), ref16, equal_nan=True)
    col = ref16.astype(np.float32).reshape(ref16.shape[0], -1)
    first = np.where(np.isnan(col).any(0), np.isnan(col).argmax(0), np.nanargmin(np.where(np.isnan(col), np.inf, col), 0))
    assert np.array_equal(host(am16).reshape(-1), first)


@pytest.mark.parametrize("cost,n_box", [("sad", 1), ("ssd", 1), ("sad", 2)])
@pytest.mark.parametrize("C,shape,hw", [(12, (9, 8, 37), 6), (12, (7, 9, 11), 3), (5, (6, 7, 9), 1), (12, (13, 16, 20), 4), (12, (5, 6, 41), 2),
                                        (12, (5, 40, 37), 2), (20, (4, 23, 74), 1)])
def test_correlate_variants_vs_oracle(U, orc, cost, n_box, C, shape, hw):
    """the SAD cost (l2r_2021 task 3 :54) and the single box filter (task 2 :60) in the fused kernel, bit-identical
    to the oracle (which is pinned to the reference scripts' own functions, tests/golden/variants.npz)."""
    rng = np.random.default_rng(C * 100 + hw)
    f = rng.random((C,) + shape, dtype=np.float32)
    m = rng.random((C,) + shape, dtype=np.float32)
    ssd, am = U.correlate(dev(f)[None], dev(m)[None], hw, 1, shape, C, cost=cost, n_box=n_box)
    rs, ra = orc.correlate(f, m, hw, cost=cost, n_box=n_box)
    assert np.array_equal(host(ssd), rs), "max |diff| %g" % np.abs(host(ssd) - rs).max()
    assert np.array_equal(host(am), ra)


def test_correlate_variants_vs_reference_golden(U, golden):
    g = golden("variants")
    for tag, cost in (("sad1", "sad"), ("ssd1", "ssd"), ("sad1_w", "sad")):
        f, m, hw = g[tag + "_fix"], g[tag + "_mov"], int(g[tag + "_hw"])
        ssd, am = U.correlate(dev(f)[None], dev(m)[None], hw, 1, f.shape[1:], 12, cost=cost, n_box=1)
        step = 7 if tag.endswith("_w") else 1
        assert np.array_equal(host(ssd)[::step], g[tag + "_ssd"]) and np.array_equal(host(am), g[tag + "_argmin"]), tag


@pytest.mark.parametrize("C,shape,hw", [(12, (9, 8, 37), 6), (12, (13, 16, 20), 4), (12, (26, 32, 37), 6), (7, (6, 7, 9), 2)])
def test_correlate_fast_mode_close_to_exact(U, orc, C, shape, hw):
    """mode="fast" (FMA + separable box sums): same real-arithmetic volume, differences in the last bits only; on a volume with a
    clear minimum the argmin is unchanged."""
    rng = np.random.default_rng(C + hw)
    f = rng.random((C,) + shape, dtype=np.float32)
    m = np.roll(f, (1, -1, 2), (1, 2, 3)) + 0.05 * rng.random((C,) + shape, dtype=np.float32)
    ssd, am = U.correlate(dev(f)[None], dev(m)[None], hw, 1, shape, C, mode="fast")
    rs, ra = orc.correlate(f, m, hw)
    rel = np.abs(host(ssd) - rs).max() / rs.max()
    flips = int((host(am) != ra).sum())
    print("fast vs exact: max rel diff %.2e, argmin flips %d of %d" % (rel, flips, ra.size))
    assert rel < 2e-6 and flips == 0


def test_argmin_ties_resolve_to_lowest_k(U):
    f = torch.zeros(1, 12, 6, 6, 6, device=DEV)
    ssd, am = U.correlate(f, f, 2, 1, (6, 6, 6), 12)      # every displacement costs 0 -> first index wins
    assert int(ssd.abs().max()) == 0 and int(am.max()) == 0


@pytest.mark.parametrize("shape,hw", [((12, 10, 14), 2), ((7, 9, 11), 3), ((9, 8, 37), 4), ((9, 8, 37), 5), ((9, 8, 37), 6), ((9, 8, 37), 7),
                                      ((9, 8, 37), 8), ((13, 16, 20), 6), ((13, 16, 20), 8)])
def test_coupled_convex_vs_oracle(U, orc, shape, hw):
    rng = np.random.default_rng(hw)
    f = rng.random((12,) + shape, dtype=np.float32)
    m = np.roll(f, (1, -2, 3), (1, 2, 3)) + 0.05 * rng.random((12,) + shape, dtype=np.float32)   # a real minimum: pruning has work to do
    rs, ra = orc.correlate(f, m, hw)
    mesh = orc.disp_mesh(hw)
    out = U.coupled_convex(dev(rs), dev(ra), dev(mesh)[:, :, None], 1, shape)
    assert np.array_equal(host(out)[0], orc.coupled_convex(rs, ra, mesh, hw))


@pytest.mark.parametrize("kind", ["foreign_argmin", "flat", "plateaus", "hw0", "zero_columns", "zero_columns_foreign", "signed_zero_columns"])
def test_coupled_convex_pruning_edge_cases(U, orc, kind):
    """The pruned (branch-and-bound) passes must return the reference's argmin for inputs that defeat the bound: an `argmin`
    argument that is not the argmin of the volume (the lower bound then comes from a streaming pass), a completely flat
    volume (every displacement ties: lowest index wins, every voxel keeps its whole window), plateaus of equal cost, and the
    single-displacement window.  zero_columns: many displacements tie at the minimum (zero background) -- boxes that the previous winner's
    cost leaves too large are closed again with the cost of the displacement nearest to the smoothed field (cand_box), also with a
    foreign `argmin` and with -0.0 entries (which compare equal to +0.0 but order below it in the keys)."""
    rng = np.random.default_rng(7)
    shape, hw = (6, 8, 12), 3
    if kind == "hw0":
        hw = 0
    n = 2 * hw + 1
    K = n ** 3
    if kind == "flat":
        ssd = np.full((K,) + shape, 0.75, np.float32)
    elif kind == "plateaus":
        ssd = (rng.integers(0, 3, (K,) + shape) * 0.5).astype(np.float32)
    else:
        ssd = rng.random((K,) + shape, dtype=np.float32)
    if kind.startswith("zero_columns") or kind == "signed_zero_columns":
        ssd[:, :, :5, :] = 0.0                                              # a zero background beside structured columns
        ssd[:, 2:4, 5:, 3:7] = 0.0                                          # and an island inside them
        ssd[K // 3, 1, 2, :] = 1e-30                                        # all-zero but for one tiny entry: an ordinary column
        if kind == "signed_zero_columns":
            ssd[::7, :, 1, :] = -0.0                                        # -0.0 entries: compare equal, yet the column is not "all +0.0"
    am = ssd.reshape(K, -1).argmin(0).reshape(shape).astype(np.int64)
    if kind in ("foreign_argmin", "zero_columns_foreign"):
        am = rng.integers(0, K, shape).astype(np.int64)
    mesh = orc.disp_mesh(hw)
    out = U.coupled_convex(dev(ssd), dev(am), dev(mesh)[:, :, None], 1, shape)
    assert np.array_equal(host(out)[0], orc.coupled_convex(ssd, am, mesh, hw))


def test_coupled_convex_bounded_worst_case(U, orc):
    """Flat cost regions keep whole search windows; once the listed boxes exceed the cost of a coalesced scan the pass streams the
    volume instead (k_argmin4_stream) -- same result, bounded time.  A zero-background volume drives every pass down that path."""
    rng = np.random.default_rng(3)
    shape, hw = (10, 12, 16), 4
    K = (2 * hw + 1) ** 3
    ssd = rng.random((K,) + shape, dtype=np.float32)
    ssd[:, :, :7, :] = 0.25                                                 # more than half of the voxels: every displacement ties
    am = ssd.reshape(K, -1).argmin(0).reshape(shape).astype(np.int64)
    mesh = orc.disp_mesh(hw)
    out = U.coupled_convex(dev(ssd), dev(am), dev(mesh)[:, :, None], 1, shape)
    assert np.array_equal(host(out)[0], orc.coupled_convex(ssd, am, mesh, hw))


def test_inverse_consistency_vs_oracle(U, orc):
    rng = np.random.default_rng(5)
    a = (0.2 * rng.standard_normal((3, 9, 11, 13))).astype(np.float32)
    b = (0.2 * rng.standard_normal((3, 9, 11, 13))).astype(np.float32)
    for it in (1, 2, 15):
        o1, o2 = U.inverse_consistency(dev(a)[None], dev(b)[None], iter=it)
        r1, r2 = orc.inverse_consistency(a, b, it)
        assert np.array_equal(host(o1)[0], r1) and np.array_equal(host(o2)[0], r2)


@pytest.mark.parametrize("src,dst", [((9, 11, 13), (36, 30, 42)), ((36, 30, 42), (18, 15, 21)), ((5, 6, 7), (5, 12, 7)), ((26, 32, 37), (160, 192, 224))])
def test_resize_vs_oracle(U, orc, src, dst):
    x = np.random.default_rng(1).standard_normal((3,) + src).astype(np.float32)
    assert np.array_equal(host(U.resize_trilinear(dev(x)[None], dst))[0], orc.resize_trilinear(x, dst))


@pytest.mark.parametrize("src", [(9, 11, 13), (1, 3, 2), (2, 1, 1), (5, 6, 130), (80, 96, 112)])
def test_resize_factor_two_kernel(U, orc, src):
    """Exact factor-2 up-sampling of a 3-channel field takes k_resize_up2 (2 x 2 x 2 outputs per thread from one 27-tap neighbourhood):
    bit-identical to the oracle and to the one-thread-per-output kernel, including non-finite taps (a tap that the reference multiplies
    by a zero weight must be the SAME tap: 0 * inf is NaN) and one-voxel axes."""
    from convexadam_amd import _lib
    rng = np.random.default_rng(sum(src))
    x = rng.standard_normal((3,) + src).astype(np.float32)
    flat = x.reshape(-1)
    flat[rng.integers(0, flat.size, max(2, flat.size // 50))] = np.inf          # scattered non-finite taps, some on the borders
    x[:, 0, 0, 0] = -np.inf
    x[:, -1, -1, -1] = np.nan
    dst = tuple(2 * s for s in src)
    got = host(U.resize_trilinear(dev(x)[None], dst))[0]
    _lib.lib().cvx_set_option(b"resize_up2", 0)
    try:
        plain = host(U.resize_trilinear(dev(x)[None], dst))[0]
    finally:
        _lib.lib().cvx_set_option(b"resize_up2", 1)
    ref = orc.resize_trilinear(x, dst)
    assert np.array_equal(got, plain, equal_nan=True) and np.array_equal(np.isnan(got), np.isnan(plain))
    assert np.array_equal(got, ref, equal_nan=True) and np.array_equal(np.isnan(got), np.isnan(ref))
    y = rng.standard_normal((3,) + src).astype(np.float32)
    assert np.array_equal(host(U.resize_trilinear(dev(y)[None], dst))[0], orc.resize_trilinear(y, dst))


@pytest.mark.parametrize("C,src,dst", [(1, (7, 9, 40), (14, 9, 300)), (5, (6, 5, 9), (11, 13, 70)), (2, (4, 4, 4), (4, 4, 4))])
def test_resize_channel_counts_and_long_rows(U, orc, C, src, dst):
    """The kernels specialise C = 1 and C = 3; other counts take the generic channel loop; rows longer than a workgroup."""
    x = np.random.default_rng(C).standard_normal((C,) + src).astype(np.float32)
    assert np.array_equal(host(U.resize_trilinear(dev(x)[None], dst))[0], orc.resize_trilinear(x, dst))


def test_grid_sample_vs_oracle(U, orc):
    rng = np.random.default_rng(2)
    vol = rng.standard_normal((4, 7, 8, 9)).astype(np.float32)
    grid = (rng.random((5, 6, 7, 3), dtype=np.float32) * 2.6 - 1.3).astype(np.float32)   # some samples outside
    assert np.array_equal(host(U.grid_sample(dev(vol)[None], dev(grid)[None]))[0], orc.grid_sample(vol, grid))


@pytest.mark.parametrize("k,passes", [(3, 1), (3, 3), (5, 3)])
def test_box_smooth_vs_oracle(U, orc, k, passes):
    x = np.random.default_rng(k).standard_normal((3, 10, 11, 12)).astype(np.float32)
    r = x
    for _ in range(passes):
        r = orc.box_zero(r, k)
    assert np.array_equal(host(U.box_smooth(dev(x)[None], k, passes))[0], r)


@pytest.mark.parametrize("niter", [1, 3, 10])
def test_adam_vs_oracle(U, orc, golden, niter):
    g = golden("adam")
    Ud, st = U.adam_run(dev(g["F2"])[None], dev(g["M2"])[None], dev(g["P0"])[None], float(g["lam"]), niter, return_state=True)
    r = orc.adam_run(g["F2"], g["M2"], g["P0"], float(g["lam"]), niter, want_grad=True)
    assert np.array_equal(host(Ud)[0], r["U"])
    assert np.array_equal(host(st["G"])[0], r["G"])
    assert np.array_equal(host(st["P"])[0], r["P"])
    assert np.array_equal(host(st["m"])[0], r["m"]) and np.array_equal(host(st["v"])[0], r["v"])


@pytest.mark.parametrize("shape", [(6, 9, 30), (5, 10, 28), (7, 9, 61), (6, 17, 60), (5, 9, 124), (4, 9, 126), (4, 8, 130), (14, 8, 12), (3, 3, 5)])
def test_adam_grid_shapes_vs_oracle(U, orc, shape):
    """Control grids that select every variant of the three-box kernels: 8 / 16 / 32 quads per row, rows that are /
    are not multiples of 4 voxels (16-byte vs scalar global access), two Adam elements per thread (8 x 124 > 960),
    rows longer than 126 voxels (tiled LDS kernel), several z chunks, a grid smaller than one tile.  C = 5 pads the
    feature chunks with zero channels."""
    rng = np.random.default_rng(sum(shape))
    C = 5
    F2 = rng.random((C,) + shape, dtype=np.float32)
    M2 = rng.random((C,) + shape, dtype=np.float32)
    P0 = (0.7 * rng.standard_normal((3,) + shape)).astype(np.float32)
    Ud, st = U.adam_run(dev(F2)[None], dev(M2)[None], dev(P0)[None], 1.25, 3, return_state=True)
    r = orc.adam_run(F2, M2, P0, 1.25, 3, want_grad=True)
    assert np.array_equal(host(Ud)[0], r["U"])
    assert np.array_equal(host(st["G"])[0], r["G"])
    assert np.array_equal(host(st["P"])[0], r["P"])
    assert np.array_equal(host(st["m"])[0], r["m"]) and np.array_equal(host(st["v"])[0], r["v"])


@pytest.mark.parametrize("variant", [1000, 2000, 1834, 2274, 1111, 2999])
@pytest.mark.parametrize("shape", [(13, 9, 8), (5, 3, 4), (12, 8, 56), (25, 17, 60), (24, 16, 116), (2, 2, 4), (14, 31, 52), (30, 20, 112), (4, 8, 132)])
def test_forward_box_tiles_vs_oracle(U, orc, shape, variant):
    """The forward three-box pass as independent tiles (boxtile.hip; option box_fwd_tile = kind * 1000 + z segments per row of the
    three passes; -1 = automatic: the benchmark grid takes kind 2): 12 x 8 x 56 and 12 x 16 x 56 tiles, every segment length from two
    planes to a whole tile, grids smaller than a tile, ragged last tiles in all three directions, rows of several x tiles and beyond
    the marching kernel's 126 voxels -- U, G, P, m, v against the oracle after three iterations."""
    from convexadam_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(sum(shape) + variant)
    C = 4
    F2 = rng.random((C,) + shape, dtype=np.float32)
    M2 = rng.random((C,) + shape, dtype=np.float32)
    P0 = (0.7 * rng.standard_normal((3,) + shape)).astype(np.float32)
    old = L.cvx_get_option(b"box_fwd_tile"), L.cvx_get_option(b"box_bwd_tile")
    assert L.cvx_set_option(b"box_fwd_tile", variant) == 0 and L.cvx_set_option(b"box_bwd_tile", variant) == 0
    try:                                                           # (the same tiles run the exact adjoint boxes + Adam update: box_bwd_tile)
        Ud, st = U.adam_run(dev(F2)[None], dev(M2)[None], dev(P0)[None], 1.25, 3, return_state=True)
    finally:
        L.cvx_set_option(b"box_fwd_tile", old[0]); L.cvx_set_option(b"box_bwd_tile", old[1])
    r = orc.adam_run(F2, M2, P0, 1.25, 3, want_grad=True)
    assert np.array_equal(host(Ud)[0], r["U"])
    assert np.array_equal(host(st["G"])[0], r["G"])
    assert np.array_equal(host(st["P"])[0], r["P"])
    assert np.array_equal(host(st["m"])[0], r["m"]) and np.array_equal(host(st["v"])[0], r["v"])


@pytest.mark.parametrize("xsplit", [-1, 0, 2, 3])
@pytest.mark.parametrize("shape", [(5, 9, 64), (30, 20, 112), (4, 9, 100), (13, 8, 68)])
def test_adam_x_tiles_vs_oracle(U, orc, shape, xsplit):
    """Rows longer than 62 voxels: the marching three-box kernels cut them into x tiles (<= 56 columns + a 4-column halo per side,
    two half-size workgroups per CU); option box_xsplit: -1 automatic, 0 one tile per row, n tiles.  Forward boxes, adjoint boxes and
    the fused Adam update against the oracle, tile edges inside and at the border of the volume, several z chunks."""
    from convexadam_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(sum(shape) + xsplit)
    C = 4
    F2 = rng.random((C,) + shape, dtype=np.float32)
    M2 = rng.random((C,) + shape, dtype=np.float32)
    P0 = (0.7 * rng.standard_normal((3,) + shape)).astype(np.float32)
    old, old_yt = L.cvx_get_option(b"box_xsplit"), L.cvx_get_option(b"box_yt")
    L.cvx_set_option(b"box_xsplit", xsplit)
    if shape[0] == 13:
        L.cvx_set_option(b"box_yt", 4)                          # 4-row tiles combined with x tiles
    try:
        Ud, st = U.adam_run(dev(F2)[None], dev(M2)[None], dev(P0)[None], 1.25, 3, return_state=True)
        sm = host(U.box_smooth(dev(P0)[None], 3, 3))[0]
    finally:
        L.cvx_set_option(b"box_xsplit", old)
        L.cvx_set_option(b"box_yt", old_yt)
    r = orc.adam_run(F2, M2, P0, 1.25, 3, want_grad=True)
    assert np.array_equal(host(Ud)[0], r["U"]) and np.array_equal(host(st["G"])[0], r["G"])
    assert np.array_equal(host(st["P"])[0], r["P"]) and np.array_equal(host(st["m"])[0], r["m"]) and np.array_equal(host(st["v"])[0], r["v"])
    b = P0
    for _ in range(3):
        b = orc.box_zero(b, 3)
    assert np.array_equal(sm, b)


def test_adam_box_kernel_variants_with_two_workgroups_per_cu(U, orc):
    """A control grid large enough for two marching workgroups per CU (80 x 64 x 112: 48 columns of 10 z chunks): the default work
    list with UNEVEN z chunks (first dispatch round long, second short: option box_uneven = length ratio in percent), equal chunks,
    extreme ratios, two columns per thread (box_cpt), 16-row tiles and the priority play (box_prio) all give the oracle's bits."""
    from convexadam_amd import _lib
    L = _lib.lib()
    shape = (80, 64, 112)
    rng = np.random.default_rng(80)
    C = 4
    F2 = rng.random((C,) + shape, dtype=np.float32)
    M2 = rng.random((C,) + shape, dtype=np.float32)
    P0 = (0.7 * rng.standard_normal((3,) + shape)).astype(np.float32)
    r = orc.adam_run(F2, M2, P0, 1.25, 2, want_grad=True)
    args = (dev(F2)[None], dev(M2)[None], dev(P0)[None], 1.25, 2)
    for opts in ({}, {"box_uneven": 100}, {"box_uneven": 130}, {"box_uneven": 400}, {"box_cpt": 2}, {"box_cpt": 2, "box_uneven": 100},
                 {"box_yt": 16, "box_wg_target": 256}, {"box_prio": 1}, {"box_prio": 2, "box_uneven": 100}, {"box_pk": 1}, {"box_pk": 1, "box_cpt": 2}, {"box_dpp": 1}, {"box_dpp": 1, "box_pk": 1, "box_uneven": 100}, {"box_adam_role": 1}, {"box_adam_role": 1, "box_dpp": 1}):
        old = {k: L.cvx_get_option(k.encode()) for k in opts}
        for k, v in opts.items():
            assert L.cvx_set_option(k.encode(), v) == 0
        try:
            Ud, st = U.adam_run(*args, return_state=True)
        finally:
            for k, v in old.items():
                L.cvx_set_option(k.encode(), v)
        assert np.array_equal(host(Ud)[0], r["U"]) and np.array_equal(host(st["G"])[0], r["G"]), opts
        assert np.array_equal(host(st["P"])[0], r["P"]) and np.array_equal(host(st["m"])[0], r["m"]) and np.array_equal(host(st["v"])[0], r["v"]), opts


def test_adam_snapshots_and_resume(U, orc, golden):
    g = golden("adam")
    args = (dev(g["F2"])[None], dev(g["M2"])[None], dev(g["P0"])[None], float(g["lam"]))
    U8, st = U.adam_run(*args, 8, snapshot_iters=(3, 8), return_state=True)
    assert np.array_equal(host(st["snapshots"][0]), orc.adam_run(g["F2"], g["M2"], g["P0"], float(g["lam"]), 3)["U"])
    assert torch.equal(st["snapshots"][1], U8[0])
    U3, s3 = U.adam_run(*args, 3, return_state=True)
    U8b = U.adam_run(*args, 5, state=s3)
    assert torch.equal(U8, U8b)


@pytest.mark.parametrize("kw", [dict(lambda_weight=0, ic=True), dict(lambda_weight=0, ic=False), dict(lambda_weight=1.25, selected_niter=5, ic=True),
                                dict(lambda_weight=1.25, selected_niter=5, ic=False), dict(lambda_weight=1.25, selected_niter=3, selected_smooth=3, ic=True)])
def test_pipeline_vs_oracle_bit_exact(M, orc, golden, kw):
    g = golden("pipeline")
    base = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2)
    out = M.convex_adam_pt(g["fix"], g["mov"], dtype=torch.float32, device=torch.device(DEV), **base, **kw)
    ref = orc.convex_adam_pipeline(g["fix"], g["mov"], **base, **kw)
    assert out.shape == ref.shape and out.dtype == np.float64
    assert np.array_equal(out, ref), "EPE %g" % epe(out, ref)


@pytest.mark.parametrize("hw,gs", [(9, 4), (12, 4), (15, 6)])
def test_pipeline_with_search_widths_beyond_8(M, orc, hw, gs):
    """The reference puts no limit on disp_hw; since round 3 the fused correlation kernel, the coupled-convex passes and the search mesh
    take half-widths up to 15 (n = 31, 29 791 displacements; windows larger than the coarse grid): whole pipeline, bit for bit."""
    from convexadam_amd.phantom import phantom
    shape = (40, 36, 44)
    fix = phantom(shape, 1, 10)
    mov = torch.roll(phantom(shape, 1, 11), (3, -2, 1), (0, 1, 2))
    kw = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=gs, disp_hw=hw, selected_niter=3, grid_sp_adam=2, ic=True)
    out = M.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device(DEV), **kw)
    assert np.array_equal(out, orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw))


def test_batched_pairs_on_internal_streams_match_single_calls(M, golden):
    """cvx_register_pairs_f32 (pairs dealt onto internal HIP streams) returns exactly what one call per pair returns."""
    g = golden("pipeline")
    fix, mov = dev(g["fix"]), dev(g["mov"])
    kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2, lambda_weight=1.25, selected_niter=4, ic=True)
    pairs_f = [fix, mov, fix]
    pairs_m = [mov, fix, fix]
    single = [M.register_pair_device(f, m, **kw).clone() for f, m in zip(pairs_f, pairs_m)]
    for ns in (1, 2, 3):
        outs = M.register_pairs_device(pairs_f, pairs_m, n_streams=ns, **kw)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(single, outs)), ns


@pytest.mark.parametrize("shape,kw", [
    ((33, 29, 37), dict(mind_r=1, mind_d=2, grid_sp=5, disp_hw=2, grid_sp_adam=3, lambda_weight=0.7, selected_niter=4, ic=True)),
    ((26, 41, 30), dict(mind_r=2, mind_d=1, grid_sp=3, disp_hw=4, grid_sp_adam=1, lambda_weight=1.5, selected_niter=2, ic=True, selected_smooth=5)),
    ((24, 24, 50), dict(mind_r=1, mind_d=3, grid_sp=4, disp_hw=5, grid_sp_adam=2, lambda_weight=1.0, selected_niter=3, ic=False)),
    ((40, 20, 23), dict(mind_r=1, mind_d=1, grid_sp=2, disp_hw=1, grid_sp_adam=4, lambda_weight=2.0, selected_niter=2, ic=True)),
    # even D and window sizes that tile: MIND is delivered only through the fused normalise + pooling pass
    ((38, 27, 52), dict(mind_r=1, mind_d=2, grid_sp=6, disp_hw=2, grid_sp_adam=2, lambda_weight=1.25, selected_niter=3, ic=True)),
    ((25, 31, 26), dict(mind_r=1, mind_d=2, grid_sp=2, disp_hw=1, grid_sp_adam=4, lambda_weight=1.0, selected_niter=2, ic=False)),
    ((31, 26, 44), dict(mind_r=2, mind_d=2, grid_sp=6, disp_hw=2, grid_sp_adam=3, lambda_weight=0.5, selected_niter=2, ic=True)),
    ((24, 25, 30), dict(mind_r=1, mind_d=1, grid_sp=6, disp_hw=1, grid_sp_adam=6, lambda_weight=1.0, selected_niter=2, ic=True)),
    ((21, 22, 26), dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=2, grid_sp_adam=4, lambda_weight=1.0, selected_niter=2, ic=True)),
    ((17, 18, 22), dict(mind_r=1, mind_d=2, grid_sp=2, disp_hw=2, grid_sp_adam=2, lambda_weight=1.0, selected_niter=2, ic=True)),
    ((26, 27, 30), dict(mind_r=1, mind_d=2, grid_sp=6, disp_hw=2, lambda_weight=0, ic=True)),
])
def test_pipeline_ragged_shapes_vs_oracle_bit_exact(M, orc, shape, kw):
    """Extents that are not multiples of the grid spacings (floor pooling drops the remainder), every mind_r/mind_d,
    grid_sp_adam 1..4, odd widths (row padding / tail rules): the whole pipeline stays bit-identical to the oracle."""
    from convexadam_amd.phantom import phantom
    fix = phantom(shape, 11, 21)
    mov = torch.roll(phantom(shape, 11, 22), (1, -1, 2), (0, 1, 2))
    out = M.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device(DEV), **kw)
    ref = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw)
    assert out.shape == ref.shape
    assert np.array_equal(out, ref), "EPE %g" % epe(out, ref)


# ---- (2) HIP vs reference goldens -----------------------------------------------------------------------
def test_mindssc_vs_reference_golden(U, golden):
    g = golden("mind")
    for key, r, d in (("mind_r1d2", 1, 2), ("mind_r2d2", 2, 2), ("mind_r1d1", 1, 1)):
        out = host(U.MINDSSC(dev(g["img"])[None, None], r, d, device=DEV))[0]
        assert np.abs(out - g[key]).max() <= 6e-8          # 1 ulp: MKL vsExp vs the shared expf restatement


def test_convex_stage_vs_reference_golden(U, golden):
    g = golden("convex")
    H, W, D, gs, hw = [int(v) for v in g["shape"]]
    ssd, am = U.correlate(dev(g["feat_fix"])[None], dev(g["feat_mov"])[None], hw, gs, (H, W, D), 12)
    assert np.array_equal(host(ssd), g["ssd"]) and np.array_equal(host(am), g["argmin"])
    soft = U.coupled_convex(ssd, am, dev(g["mesh"])[:, :, None], gs, (H, W, D))
    assert np.array_equal(host(soft)[0], g["soft"])
    o1, o2 = U.inverse_consistency(dev(g["ic_in1"])[None], dev(g["ic_in2"])[None], iter=15)
    assert np.array_equal(host(o1)[0], g["ic_out1"]) and np.array_equal(host(o2)[0], g["ic_out2"])
    assert np.array_equal(host(U.resize_trilinear(dev(g["disp_hr"])[None], (H // 2, W // 2, D // 2)))[0], g["disp_lr"])


def test_correlate_c20_vs_reference_golden(U, golden):
    g = golden("correlate_c20")
    ssd, am = U.correlate(dev(g["fix"])[None], dev(g["mov"])[None], 1, 1, g["fix"].shape[1:], 20)
    assert np.array_equal(host(ssd), g["ssd"]) and np.array_equal(host(am), g["argmin"])


def test_adam_vs_reference_golden(U, golden):
    g = golden("adam")
    args = (dev(g["F2"])[None], dev(g["M2"])[None], dev(g["P0"])[None], float(g["lam"]))
    U1, st = U.adam_run(*args, 1, return_state=True)
    assert np.array_equal(host(U1)[0], g["U_1"]) and np.array_equal(host(st["G"])[0], g["G_1"])   # autograd-exact
    assert np.abs(host(st["P"])[0] - g["P_1"]).max() <= 2.5e-7                                    # MKL sqrt, 1 ulp
    for niter, tol in ((5, 5e-6), (20, 5e-5)):
        assert np.abs(host(U.adam_run(*args, niter))[0] - g["U_%d" % niter]).max() <= tol


@pytest.mark.parametrize("key,kw,tol", [("convex_only_ic", dict(lambda_weight=0, ic=True), 1e-5), ("convex_only_noic", dict(lambda_weight=0, ic=False), 1e-5),
                                        ("adam_1", dict(selected_niter=1), 1e-5), ("adam_5", dict(selected_niter=5), 1e-4),
                                        ("adam_20", dict(selected_niter=20), 1e-3), ("adam_5_smooth3", dict(selected_niter=5, selected_smooth=3), 1e-4),
                                        ("adam_5_noic", dict(selected_niter=5, ic=False), 1e-4)])
def test_pipeline_vs_reference_golden(M, golden, key, kw, tol):
    """North-star tolerance: mean endpoint error < 1e-3 voxel against the reference's CPU float32 field."""
    g = golden("pipeline")
    args = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2, lambda_weight=1.25, ic=True)
    args.update(kw)
    out = M.convex_adam_pt(g["fix"], g["mov"], dtype=torch.float32, device=torch.device(DEV), **args)
    assert out.shape == g[key].shape
    assert epe(out, g[key]) < tol


def test_translation_known_answers(M, golden):
    from convexadam_amd.phantom import phantom
    g = golden("translation64")
    fix = phantom((64, 64, 64), 2, 20)
    for name, sh, gs in (("roll_4_0_m8_gs4", (4, 0, -8), 4), ("roll_6_m6_0_gs6", (6, -6, 0), 6)):
        out = M.convex_adam_pt(fix, torch.roll(fix, sh, (0, 1, 2)), lambda_weight=0, grid_sp=gs, disp_hw=4, dtype=torch.float32,
                               device=torch.device(DEV))
        assert np.allclose(out[16:48, 16:48, 16:48].mean((0, 1, 2)), g[name], atol=1e-4)
        assert np.abs(out[::4, ::4, ::4] - g[name + "_sub"]).max() < 1e-3


def test_label_features_and_nnunet_path(orc, golden):
    from convexadam_amd import convex_adam_nnUNet as N
    g = golden("labels")
    lf, lm = g["lab_fix"].astype(np.float32), g["lab_mov"].astype(np.float32)
    ff, fm = N.extract_features(dev(lf), dev(lm), device=DEV)
    rf, rm, _ = orc.label_features(lf, lm, 10.0)
    assert np.array_equal(host(ff)[0], rf) and np.array_equal(host(fm)[0], rm)
    assert np.array_equal(host(ff)[0].reshape(ff.shape[1], -1).max(1), g["weights"])      # the reference's weights, bit for bit (round 3)
    out = N.convex_adam_pt(dev(lf), dev(lm), 1.25, 4, 2, 5, 0, device=DEV)
    assert out.shape == lf.shape + (3,) and np.isfinite(out).all()
    # the wrapper = label features + the feature pipeline, quantised through fp16 like the reference's `.cpu().half()` (:151-154)
    want = orc.convex_adam_pipeline(None, None, lambda_weight=1.25, grid_sp=4, disp_hw=2, selected_niter=5, features=(rf, rm))
    assert np.array_equal(out, want.astype(np.float16).astype(np.float64))


def test_nnunet_pipeline_vs_reference_golden(M, U, orc, golden, nnunet):
    """BASELINE configs[3] end to end against the reference itself (tests/golden/nnunet.npz: convex_adam_nnUNet.py:41-159 run on an
    18-label pair -- C >= 16, ATen's cascade channel sum -- at the convex stage and 1 / 5 / 20 Adam iterations).  From the reference's
    LABEL MAPS the HIP pipeline (label histogram, the restated torch.pow weights, feature expansion, registration) is bit-identical at
    the convex stage, within the sqrt-ulp sensitivity after Adam, and bit-identical at every horizon with the golden host's sqrt table."""
    from convexadam_amd import convex_adam_nnUNet as N
    g = golden("nnunet")
    gs, hw, gsa = (int(v) for v in g["cfg"])
    lf, lm, ff, fm = nnunet.features(g)
    hf, hm = N.extract_features(dev(lf), dev(lm), device=DEV)
    assert np.array_equal(host(hf)[0], ff) and np.array_equal(host(hm)[0], fm)           # the library's own features ARE the reference's
    kw = dict(feat_fixed=hf[0], feat_moving=hm[0], grid_sp=gs, disp_hw=hw, grid_sp_adam=gsa, ic=True, cost_scale=12.0)
    field = lambda t: np.moveaxis(host(t), 0, -1)
    nnunet.field_checks(g, "convex", field(M.register_pair_device(lambda_weight=0, **kw)), exact=True)
    for niter, tol in ((1, 1e-6), (5, 1e-5), (20, 1e-3)):
        out = field(M.register_pair_device(lambda_weight=1.25, selected_niter=niter, **kw))
        assert nnunet.field_checks(g, "adam_%d" % niter, out, exact=False) < tol, niter
        want = orc.convex_adam_pipeline(None, None, lambda_weight=1.25, selected_niter=niter, grid_sp=gs, disp_hw=hw, grid_sp_adam=gsa, features=(ff, fm))
        assert np.array_equal(out, want.astype(np.float32)), niter
    q = golden("mkl_vssqrt_low")
    U.set_adam_sqrt_table(q["normal"], q["denormal"], device=DEV)
    try:
        for niter in (1, 5, 20):
            nnunet.field_checks(g, "adam_%d" % niter, field(M.register_pair_device(lambda_weight=1.25, selected_niter=niter, **kw)), exact=True)
    finally:
        U.set_adam_sqrt_table(None)


def test_masked_feature_extraction(M, orc, golden):
    """use_mask=True branch (BASELINE configs[2]): erosion, gather, up-sampling and merge run as HIP kernels (only the
    EDT index search is scipy on the host, like the reference); bit-identical to the oracle, 1 ulp from the reference."""
    g = golden("masked")
    ff, fm = M.extract_features(torch.from_numpy(g["img_fix"]), torch.from_numpy(g["img_mov"]), 1, 2, True,
                                torch.from_numpy(g["mask_fix"]), torch.from_numpy(g["mask_mov"]), device=torch.device(DEV),
                                dtype=torch.float32)
    for out, img, mask, key in ((ff, g["img_fix"], g["mask_fix"], "feat_fix"), (fm, g["img_mov"], g["mask_mov"], "feat_mov")):
        filled, _ = orc.replicate_fill(img, mask)
        assert np.array_equal(host(out)[0], orc.mindssc(filled, 1, 2))
        assert np.abs(host(out)[0] - g[key]).max() <= 6e-8
    with pytest.raises(ValueError):
        M.extract_features(torch.zeros(9, 8, 8), torch.zeros(9, 8, 8), 1, 2, True, torch.ones(9, 8, 8), torch.ones(9, 8, 8),
                           device=torch.device(DEV), dtype=torch.float32)


def test_sweep_smoothers(orc, golden):
    """Row P on the GPU: forward and adjoint of every smoother of the sweep bit-identical to the oracle and to the
    reference goldens; the autograd wrapper gives the same adjoint; the fused Adam loop accepts them."""
    from convexadam_amd import convexAdam_hyper_util as HU
    from convexadam_amd import convex_adam_utils as U
    g, a = golden("smoothers"), golden("adam")
    mods = {"gauss07": HU.GaussianSmoothing(0.7), "gauss10": HU.GaussianSmoothing(1.0), "kov16": HU.kovesi_spline(1.6, 4),
            "kov19": HU.kovesi_spline(1.9, 4), "kov28": HU.kovesi_spline(2.8, 4)}
    assert np.array_equal(np.array(list(mods["gauss07"].spec.gauss_w), np.float32), g["gauss07_w"])
    for k, mod in mods.items():
        x = dev(g["x"])[None].requires_grad_(True)
        y = mod(x)
        y.backward(dev(g["go"])[None])
        assert np.array_equal(host(y)[0], g[k + "_fwd"]), k
        assert np.array_equal(host(x.grad)[0], g[k + "_bwd"]), k
    for k in ("gauss07", "kov19", "kov28"):
        sm = orc.make_smoother(gauss_w=g["gauss07_w"]) if k == "gauss07" else orc.make_smoother(mods[k].sizes)
        Ud, st = U.adam_run(dev(a["F2"])[None], dev(a["M2"])[None], dev(a["P0"])[None], 0.8, 4, smoother=mods[k], return_state=True)
        r = orc.adam_run(a["F2"], a["M2"], a["P0"], 0.8, 4, want_grad=True, smoother=sm)
        assert np.array_equal(host(Ud)[0], r["U"]) and np.array_equal(host(st["G"])[0], r["G"]) and np.array_equal(host(st["P"])[0], r["P"]), k
    # the packaged chain [3,3,3] given as a smoother takes the fused kernels and equals the default loop
    U3 = U.adam_run(dev(a["F2"])[None], dev(a["M2"])[None], dev(a["P0"])[None], 1.25, 3, smoother=HU.kovesi_spline(1.3, 4))
    assert torch.equal(U3, U.adam_run(dev(a["F2"])[None], dev(a["M2"])[None], dev(a["P0"])[None], 1.25, 3))


# ---- (3) properties at full size (BASELINE.json configs 2 and 3) ------------------------------------------
@pytest.mark.timeout(900)
def test_full_size_identity_and_determinism(M):
    """160x192x224, disp_hw 6, 80 Adam iterations (config 2): registering an image to itself gives a field
    that is exactly zero at the convex stage and stays < 0.1 voxel after Adam (reference test
    test_convex_adam_identity, atol 0.1); two runs are bit-identical (no atomics on float data)."""
    from convexadam_amd.phantom import phantom
    fix = phantom((160, 192, 224), 1, 10).to(DEV)
    conv = M.register_pair_device(fix, fix, lambda_weight=0, grid_sp=6, disp_hw=6, ic=True)
    assert float(conv.abs().max()) == 0.0
    a = M.register_pair_device(fix, fix, grid_sp=6, disp_hw=6, selected_niter=80)
    b = M.register_pair_device(fix, fix, grid_sp=6, disp_hw=6, selected_niter=80)
    assert torch.equal(a, b)
    assert float(a.abs().max()) < 0.1


@pytest.mark.timeout(900)
def test_full_size_translation_recovery(M):
    """Config 2 size, known shift: the recovered field equals the shift in the interior (reference
    test_convex_adam_translation: > 90 % of the central voxels within 1 voxel)."""
    from convexadam_amd.phantom import phantom
    fix = phantom((160, 192, 224), 1, 10).to(DEV)
    sh = (6, -4, 8)
    mov = torch.roll(fix, sh, (0, 1, 2))
    u = M.register_pair_device(fix, mov, grid_sp=6, disp_hw=6, selected_niter=80)
    c = u[:, 32:128, 38:154, 45:179]
    for a in range(3):
        assert float(((c[a] - sh[a]).abs() < 1.0).float().mean()) > 0.9
        assert abs(float(c[a].mean()) - sh[a]) < 0.25


@pytest.mark.timeout(900)
def test_large_motion_config3_argmin_consistency(U):
    """Config 3 (224x192x224, disp_hw 8, grid_sp 6 -> 4913 x 37x32x37 cost volume): the fused argmin equals
    torch-free re-evaluation of the minimum over the stored cost volume, and a rolled feature volume is
    recovered exactly (zero cost at the true displacement)."""
    rng = torch.Generator().manual_seed(0)
    f = torch.rand(1, 12, 37, 32, 37, generator=rng).to(DEV)
    m = torch.roll(f, (3, -5, 7), (2, 3, 4))
    ssd, am = U.correlate(f, m, 8, 6, (224, 192, 224), 12)
    assert ssd.shape == (4913, 37, 32, 37)
    mins = ssd.min(0)
    assert torch.equal(mins.indices, am)          # first-min tie rule == torch.argmin on device data
    k = (7 + 8) * 289 + (-5 + 8) * 17 + (3 + 8)
    inner = am[12:25, 12:20, 12:25]
    assert int((inner == k).all()) == 1
    assert float(ssd[k, 12:25, 12:20, 12:25].abs().max()) == 0.0


@pytest.mark.timeout(900)
def test_full_size_config3_pipeline(M):
    """BASELINE configs[2] geometry without the mask: 224x192x224, disp_hw 8 (4913-way search), 20 Adam iterations.
    Identity pair -> zero field at the convex stage; shifted pair -> shift recovered; memory for the 861 MB cost
    volume + 930 MB intermediate is carved from one workspace."""
    from convexadam_amd.phantom import phantom
    fix = phantom((224, 192, 224), 4, 40).to(DEV)
    conv = M.register_pair_device(fix, fix, lambda_weight=0, grid_sp=6, disp_hw=8, ic=True)
    assert float(conv.abs().max()) == 0.0
    sh = (12, -6, 18)                                   # multiples of grid_sp: exactly representable by the search mesh
    u = M.register_pair_device(fix, torch.roll(fix, sh, (0, 1, 2)), grid_sp=6, disp_hw=8, selected_niter=20)
    c = u[:, 60:160, 50:140, 60:160]
    for a in range(3):       # the reference's own criterion: within 1 voxel (test_convex_adam_translation); the coarse-grid
        assert abs(float(c[a].mean()) - sh[a]) < 1.0      # stage under-estimates by ~5 % (6 -> 5.73)


@pytest.mark.timeout(900)
def test_full_size_config4_label_features(orc):
    """BASELINE configs[3]: multi-channel (nnUNet-style) path with C = 32 label channels on a 160x192x160 volume:
    exercises the cascade channel sum (C >= 16), channel padding in the Adam gather and a 0.9 GB feature volume."""
    from convexadam_amd import convex_adam_nnUNet as N
    from convexadam_amd.phantom import label_phantom
    lab = label_phantom((160, 192, 160), 32, 3)
    assert int(lab.max()) == 31
    labm = torch.roll(lab, (4, -2, 6), (0, 1, 2))
    ff, fm = N.extract_features(lab.to(DEV), labm.to(DEV), device=DEV)
    assert ff.shape[1] == 32
    from convexadam_amd.convex_adam_MIND import register_pair_device
    u = register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], lambda_weight=1.25, grid_sp=4, disp_hw=4, selected_niter=10,
                             grid_sp_adam=2, ic=True)
    assert u.shape == (3, 160, 192, 160) and bool(torch.isfinite(u).all())
    c = u[:, 40:120, 48:144, 40:120]
    for a, s in enumerate((4, -2, 6)):
        assert abs(float(c[a].mean()) - s) < 1.0
    # small-volume bit parity of the same path against the oracle (C = 20 -> cascade + channel padding 20 -> 20)
    rng = np.random.default_rng(0)
    f = rng.random((20, 24, 20, 28), dtype=np.float32)
    m = np.roll(f, (1, -1, 2), (1, 2, 3)).copy()
    kw = dict(lambda_weight=1.25, grid_sp=4, disp_hw=2, selected_niter=3, grid_sp_adam=2, ic=True)
    out = host(register_pair_device(feat_fixed=dev(f), feat_moving=dev(m), **kw))
    ref = orc.convex_adam_pipeline(None, None, features=(f, m), **kw)
    assert np.array_equal(np.moveaxis(out, 0, -1).astype(np.float64), ref)


# ---- (4) evaluation operators of the sweep and apply_convex ------------------------------------
@pytest.fixture(scope="module")
def HU():
    from convexadam_amd import convexAdam_hyper_util
    return convexAdam_hyper_util


@pytest.fixture(scope="module")
def morc():
    from oracle import metrics_oracle
    return metrics_oracle


@pytest.mark.parametrize("shape,convert1", [((9, 10, 11), False), ((9, 10, 11), True), ((5, 6, 37), False), ((40, 33, 29), True),
                                            ((4, 9, 10), False), ((3, 5, 6), True)])      # extents <= 4: empty, like the reference's slice
def test_jacobian_determinant_vs_oracle(HU, morc, shape, convert1):
    rng = np.random.default_rng(sum(shape))
    flow = (rng.standard_normal((3,) + shape) * (0.1 if convert1 else 2.0)).astype(np.float32)
    assert np.array_equal(host(HU.jacobian_determinant_3d(dev(flow)[None], convert1)), morc.jacobian_determinant_3d(flow, convert1))


def test_metrics_vs_reference_golden(HU, morc, golden):
    g = golden("metrics")
    disp = dev(g["disp"])[None]
    jac = HU.jacobian_determinant_3d(disp, False)
    assert np.array_equal(host(jac), g["jac_vox"])
    assert np.array_equal(host(HU.jacobian_determinant_3d(dev(g["disp_norm"])[None], True)), g["jac_norm"])
    std, neg = HU.jacobian_log_std_and_folding(jac)
    assert abs(std - float(g["jac_log_std"])) <= 1e-5 * float(g["jac_log_std"])      # torch's float32 log / std are not restated
    assert abs(neg - float(g["jac_neg_frac"])) <= 1e-6
    warped = HU.warp_labels_nearest(dev(g["seg_moving"]), disp)
    assert np.array_equal(host(warped), g["seg_warped"])
    assert np.array_equal(HU.dice_coeff(dev(g["seg_fixed"]), warped, 7).numpy(), g["dice"])
    tre, samp = HU.tre_at_keypoints(disp, g["key_fixed"], g["key_moving"])
    assert np.array_equal(samp.numpy(), g["disp_sampled"])
    assert np.allclose(tre.numpy(), g["tre"], rtol=2e-7, atol=0)                       # MKL sqrt of the reference: <= 1 ulp
    assert np.array_equal(HU.sort_rank(torch.from_numpy(g["rank_in"])).numpy(), g["rank_out"])


@pytest.mark.parametrize("shape", [(22, 26, 30), (7, 5, 9)])
def test_warp_labels_and_dice_vs_oracle(HU, morc, shape):
    rng = np.random.default_rng(shape[0])
    seg = rng.integers(0, 9, shape).astype(np.float32)
    seg2 = np.roll(seg, (1, 0, -1), (0, 1, 2))
    disp = (rng.standard_normal((3,) + shape) * 2.5).astype(np.float32)                # many samples leave the volume
    w = HU.warp_labels_nearest(dev(seg), dev(disp)[None])
    assert np.array_equal(host(w), morc.warp_labels_nearest(seg, disp))
    assert np.array_equal(HU.dice_coeff(dev(seg2), w, 9).numpy(), morc.dice_coeff(seg2, host(w), 9))


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 4, 5), (9, 7, 70), (12, 33, 64), (20, 24, 130), (40, 5, 257)])
def test_edt_squared_vs_scipy(HU, shape):
    """Distance-only EDT (Meijster passes, integer arithmetic) == scipy's distances squared: random masks of several densities,
    rows / planes without a zero voxel, a single zero voxel, a lattice (ties everywhere)."""
    from scipy.ndimage import distance_transform_edt as edt
    rng = np.random.default_rng(sum(shape))
    masks = [(rng.random(shape) < pz).astype(np.float32) for pz in (0.5, 0.9, 0.995)]
    one = np.ones(shape, np.float32); one[tuple(s // 2 for s in shape)] = 0; masks.append(one)
    lat = np.ones(shape, np.float32); lat[::3, ::2, ::4] = 0; masks.append(lat)
    masks.append(np.zeros(shape, np.float32))
    for m in masks:
        if not (m == 0).any():
            continue
        got = host(HU.edt_squared(dev(m)))
        ref = np.rint(edt(m) ** 2).astype(np.int64)
        assert np.array_equal(got.astype(np.int64), ref)
    # batched: independent volumes in one set of launches
    keep = [m for m in masks if (m == 0).any()]
    got = host(HU.edt_squared(dev(np.stack(keep))))
    for i, m in enumerate(keep):
        assert np.array_equal(got[i].astype(np.int64), np.rint(edt(m) ** 2).astype(np.int64)), i


def test_hd95_vs_golden_and_oracle(HU, morc, golden):
    """cupy_hd95 on the device (feature transforms + histogram percentile) == the reference capture and the numpy/scipy oracle."""
    g = golden("hd95")
    sf, sm = torch.from_numpy(g["seg_fixed"]).to(DEV), torch.from_numpy(g["seg_moving"]).to(DEV)
    out = HU.cupy_hd95(sf.long(), sm.long(), 6)
    assert out.dtype == torch.float64 and out.device.type == "cuda"
    assert np.array_equal(host(out), g["hd95_p1"])
    assert np.array_equal(host(HU.cupy_hd95(sf, sm, 6, precision=2)), g["hd95_p2"])
    rng = np.random.default_rng(11)
    for shape in ((9, 7, 12), (16, 16, 16), (5, 30, 11)):
        a = rng.integers(0, 4, [max(1, s // 3) for s in shape])
        a = np.kron(a, np.ones((3, 3, 3), np.int64))[: shape[0], : shape[1], : shape[2]]
        a = np.pad(a, [(0, s - t) for s, t in zip(shape, a.shape)])
        b = np.roll(a, (1, -1, 2), (0, 1, 2))
        for prec in (1, 3):
            got = host(HU.cupy_hd95(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), 4, precision=prec))
            assert np.array_equal(got, morc.hd95(a, b, 4, prec)), (shape, prec)
    with pytest.raises(RuntimeError):
        HU.cupy_hd95(sf, sm, 3)                                                         # label 5 present: one_hot would fail
    # any positive scale factor of F.interpolate's nearest mode (round 4; captured from the reference like the integer ones)
    for key, prec in (("hd95_p1_5", 1.5), ("hd95_p0_5", 0.5), ("hd95_p2_5", 2.5)):
        assert np.array_equal(host(HU.cupy_hd95(sf, sm, 6, precision=prec)), g[key]), key
    a = rng.integers(0, 4, (7, 9, 11))
    b = np.roll(a, (1, 0, -1), (0, 1, 2))
    for prec in (0.7, 1.3, 2.05, 3.0, 9.0):                                             # 2.05: 20 -> 41 voxels; 9: beyond the integer kernel's range
        assert np.array_equal(host(HU.cupy_hd95(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), 4, precision=prec)), morc.hd95(a, b, 4, prec)), prec
    with pytest.raises(ValueError):
        HU.cupy_hd95(sf, sm, 6, precision=0)
    with pytest.raises(RuntimeError):
        HU.cupy_hd95(sf, sm, 6, precision=0.01)                                         # an empty resampled volume (torch: sizes should be greater than 0)


def test_file_wrapper_writes_the_field_as_nifti(M, tmp_path):
    """convex_adam (convex_adam_MIND.py:205-248): NIfTI in, disp.nii.gz out with the fixed image's affine -- through nibabel when it
    is installed, else through the built-in NIfTI-1 reader / writer; the stored field equals convex_adam_pt's return value."""
    from convexadam_amd import nifti_io as N
    from convexadam_amd.phantom import phantom
    shape = (32, 28, 36)
    fix = phantom(shape, 7, 70).numpy()
    mov = np.roll(fix, (1, -1, 2), (0, 1, 2))
    aff = np.diag([1.5, 1.5, 2.0, 1.0]); aff[:3, 3] = (-20.0, 11.0, 3.0)
    pf, pm = str(tmp_path / "fixed.nii.gz"), str(tmp_path / "moving.nii.gz")
    N.save_image(fix, aff, pf)
    N.save_image(mov, aff, pm)
    kw = dict(grid_sp=4, disp_hw=2, selected_niter=3, grid_sp_adam=2)
    M.convex_adam(pf, pm, result_path=str(tmp_path), **kw)
    got = N.load_fdata(str(tmp_path / "disp.nii.gz"))
    ref = M.convex_adam_pt(torch.from_numpy(N.load_fdata(pf)).float(), torch.from_numpy(N.load_fdata(pm)).float(), **kw)
    assert got.shape == shape + (3,) and np.array_equal(got, ref)
    assert np.allclose(N.load_affine(str(tmp_path / "disp.nii.gz")), aff)


def test_apply_convex_vs_scipy_and_golden(morc, golden):
    from scipy.ndimage import map_coordinates
    from convexadam_amd.apply_convex import apply_convex
    g = golden("metrics")
    dd = g["disp"].transpose(1, 2, 3, 0).astype(np.float64)
    out = apply_convex(dd, g["moving"])
    assert out.dtype == np.float64 and np.array_equal(out, g["warped"])                # captured from the reference
    assert np.array_equal(out, morc.apply_convex(dd, g["moving"]))
    rng = np.random.default_rng(3)
    mov = rng.random((13, 9, 17))
    d2 = rng.standard_normal((13, 9, 17, 3)) * 3.0
    idn = np.meshgrid(np.arange(13), np.arange(9), np.arange(17), indexing="ij")
    assert np.array_equal(apply_convex(d2, mov), map_coordinates(mov, d2.transpose(3, 0, 1, 2) + idn, order=1))   # scipy itself
    t32 = apply_convex(torch.from_numpy(d2), torch.from_numpy(mov.astype(np.float32)))
    assert t32.dtype == np.float32
    # integer inputs: a numpy int16 image (what a SimpleITK CT yields) is interpolated in float64 like the reference's validate_image
    # (astype(float)); an integer TENSOR keeps its dtype and is rounded the way scipy fills an integer output array
    mi = (rng.random((13, 9, 17)) * 2000 - 1000).astype(np.int16)
    wi = apply_convex(d2, mi)
    assert wi.dtype == np.float64 and np.array_equal(wi, map_coordinates(mi.astype(float), d2.transpose(3, 0, 1, 2) + idn, order=1))
    ti = apply_convex(torch.from_numpy(d2), torch.from_numpy(mi))
    assert ti.dtype == np.int16 and np.array_equal(ti, map_coordinates(mi, d2.transpose(3, 0, 1, 2) + idn, order=1))


def test_full_size_metrics_properties(HU):
    """BASELINE-size field: identity -> det 1 everywhere, no folding, Dice 1; a constant shift moves the labels by whole voxels."""
    H, W, D = 160, 192, 224
    zero = torch.zeros(1, 3, H, W, D, device=DEV)
    jac = HU.jacobian_determinant_3d(zero, False)
    assert float(jac.min()) == 1.0 and float(jac.max()) == 1.0
    std, neg = HU.jacobian_log_std_and_folding(jac)
    assert std == 0.0 and neg == 0.0
    seg = torch.randint(0, 14, (H, W, D), device=DEV).float()
    assert torch.equal(HU.warp_labels_nearest(seg, zero), seg)
    d = HU.dice_coeff(seg, seg, 14).numpy()
    assert d.shape == (13,) and np.all(d > 0.999999) and np.all(d <= 1.0)      # 2x / (1e-8 + 2x) in float32
    shift = zero.clone()
    shift[0, 0] = 3.0
    shift[0, 2] = -2.0
    w = HU.warp_labels_nearest(seg, shift)
    assert torch.equal(w[:-3, :, 2:], seg[3:, :, :-2]) and float(w[-3:].abs().max()) == 0.0


def test_sweep_driver_scores_items_on_the_device(tmp_path):
    """Single-process run of the sharded driver with --evaluate: the registration recovers the known roll, so Dice rises,
    TRE falls and no voxel folds."""
    import json
    from convexadam_amd import sweep
    out = tmp_path / "sweep.json"
    assert sweep.main(["--pairs", "1", "--settings", "2", "--shape", "48", "48", "48", "--niter", "10", "--evaluate", "--out", str(out)]) == 0
    s = json.loads(out.read_text())
    assert s["n_items"] == 2 and len(s["results"]) == 2 and len(s["ranking"]["rank"]) == 2
    for r in s["results"]:
        assert r["dice"] > r["dice_before"] + 0.1 and r["tre"] < 0.5 * r["tre_before"] and r["folding"] == 0.0


def test_bench_line_with_two_ranks_on_one_gpu():
    """bench.py's N > 1 path (one process per rank under torch.distributed.run, barrier + max over ranks, rank 0 prints): two ranks share
    this box's GPU through the test hooks (gloo instead of RCCL, which refuses two ranks on one device).  Checks the contract of the line,
    not a speed: whole-job value = 2 pairs per step over the slower rank's clock."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-batched"]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # exactly one JSON line, from rank 0
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["steps"] == 3 and b["warmup"] == 1 and b["scaling"] == "weak" and b["unit"] == "pairs/s"
    assert abs(b["value"] - 2 * 3 / (b["ms_per_step"] * 3e-3)) < 1e-6 * b["value"]
    assert "cpu_baseline" not in b and b["roofline"]["frac"] > 0 and b["vs_baseline"] is None


# ---- (5) Euclidean feature transform of the masked path ------------------------------------------------
def test_feature_transform_vs_scipy_and_oracle(M, orc):
    """Device EDT indices == scipy.ndimage.distance_transform_edt(return_indices=True) (the call of convex_adam_MIND.py:44),
    tie-breaking included: random masks of every density, a lattice whose cell centres are all ties, single sites."""
    from scipy.ndimage import distance_transform_edt as edt
    rng = np.random.default_rng(1)
    cases = []
    for _ in range(40):
        shape = tuple(int(v) for v in rng.integers(2, 20, 3))
        m = rng.random(shape) < rng.choice([0.02, 0.1, 0.3, 0.6, 0.9])
        if m.all():
            m.flat[int(rng.integers(m.size))] = False
        cases.append(m)
    for shape in ((9, 9, 9), (16, 16, 16), (40, 48, 56)):
        m = np.ones(shape, bool); m[::4, ::4, ::4] = False; cases.append(m)
        m = np.ones(shape, bool); m[0, 0, 0] = m[-1, -1, -1] = m[0, -1, 0] = False; cases.append(m)
    for m in cases:
        got = host(M.feature_transform(dev(m.astype(np.float32))))
        assert np.array_equal(got, edt(m, return_indices=True)[1]), m.shape
        assert np.array_equal(got, orc.feature_transform(m))


def test_feature_transform_half_resolution_oasis_size(M):
    """80 x 96 x 112 (the half-resolution grid of the masked path at BASELINE size), ellipsoid mask: equals scipy."""
    from scipy.ndimage import distance_transform_edt as edt
    z, y, x = np.meshgrid(np.linspace(-1, 1, 80), np.linspace(-1, 1, 96), np.linspace(-1, 1, 112), indexing="ij")
    outside = (z / 0.7) ** 2 + (y / 0.7) ** 2 + (x / 0.7) ** 2 > 1.0
    got = host(M.feature_transform(dev(outside.astype(np.float32))))
    assert np.array_equal(got, edt(outside, return_indices=True)[1])


# ---- (6) randomised configurations -------------------------------------------------------------------------------------------
def test_fuzz_pipeline_and_adam_vs_oracle_bit_exact(M, U, orc):
    """40 random small pairs (extents, MIND radius / dilation, both grid spacings, search width, lambda, iterations, ic, final
    smoothing) and 16 random control grids (row lengths around every kernel-variant boundary): bit-identical to the oracle."""
    from convexadam_amd.phantom import phantom
    rng = np.random.default_rng(20260928)
    for trial in range(40):
        gs, gsa, hw = int(rng.choice([2, 3, 4, 5, 6])), int(rng.choice([1, 2, 3, 4])), int(rng.integers(1, 9))
        shape = tuple(int(max(2 * gs, 2 * gsa, 8) + rng.integers(0, 30)) for _ in range(3))
        kw = dict(mind_r=int(rng.choice([1, 2])), mind_d=int(rng.choice([1, 2, 3])), grid_sp=gs, disp_hw=hw, grid_sp_adam=gsa,
                  lambda_weight=float(rng.choice([0.0, 0.7, 1.25])), selected_niter=int(rng.integers(1, 4)), ic=bool(rng.integers(0, 2)),
                  selected_smooth=int(rng.choice([0, 0, 3])))
        fix = phantom(shape, 100 + trial, 200 + trial)
        mov = torch.roll(phantom(shape, 100 + trial, 300 + trial), (1, -1, 1), (0, 1, 2))
        out = M.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device(DEV), **kw)
        ref = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw)
        assert out.shape == ref.shape and np.array_equal(out, ref), (shape, kw)
    for trial in range(16):
        shape = tuple(int(rng.integers(3, 40)) for _ in range(2)) + (int(rng.choice([5, 17, 29, 30, 31, 45, 61, 62, 63, 90, 125, 126, 127, 140])),)
        C = int(rng.choice([1, 4, 5, 12]))
        F2 = rng.random((C,) + shape, dtype=np.float32)
        M2 = rng.random((C,) + shape, dtype=np.float32)
        P0 = (0.7 * rng.standard_normal((3,) + shape)).astype(np.float32)
        Ud, st = U.adam_run(dev(F2)[None], dev(M2)[None], dev(P0)[None], 1.0, 2, return_state=True)
        r = orc.adam_run(F2, M2, P0, 1.0, 2, want_grad=True)
        assert np.array_equal(host(Ud)[0], r["U"]) and np.array_equal(host(st["P"])[0], r["P"]), (shape, C)
        assert np.array_equal(host(st["m"])[0], r["m"]) and np.array_equal(host(st["v"])[0], r["v"]), (shape, C)


# ---- (7) the headline configurations, numerically, at FULL size ------------------------------------------------------------------
BENCH_SHAPE = (160, 192, 224)
BENCH_CFG = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=6, disp_hw=6, selected_niter=80, selected_smooth=0, grid_sp_adam=2, ic=True)


@pytest.mark.timeout(1800)
def test_full_size_benchmark_pair_bit_identical_to_oracle(M, orc):
    """BASELINE configs[1], the exact pair bench.py times (160x192x224, hw 6, gs 6, ic, 80 Adam iterations): the HIP field equals
    the CPU oracle's bit for bit."""
    from convexadam_amd.phantom import deformed_pair
    fix, mov = deformed_pair(BENCH_SHAPE, 0, 4.0)
    out = host(M.register_pair_device(fix.to(DEV), mov.to(DEV), **BENCH_CFG))
    ref = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **BENCH_CFG)
    assert np.array_equal(np.moveaxis(out, 0, -1).astype(np.float64), ref), "EPE %g" % epe(np.moveaxis(out, 0, -1), ref)


@pytest.mark.timeout(1800)
def test_full_size_benchmark_pair_vs_reference_golden(M, U, golden):
    """The same pair against the reference itself (tests/golden/make_golden_fullsize.py): the convex stage is bit-identical; the
    north-star tolerance (mean EPE < 1e-3 voxel) holds up to 40 Adam iterations; at 80 iterations the loop has amplified MKL's
    1-ulp exp / sqrt sites and the bar is the reference's own sensitivity to a 1-ulp perturbation of its warped features."""
    from convexadam_amd.phantom import deformed_pair
    g = golden("fullsize")
    s = int(g["sub"])
    fix, mov = [t.to(DEV) for t in deformed_pair(BENCH_SHAPE, 0, 4.0)]
    kw = dict(BENCH_CFG)
    conv = M.register_pair_device(fix, mov, **dict(kw, lambda_weight=0))
    assert torch.equal(conv, U.resize_trilinear(dev(g["c1_coarse_ic"])[None], BENCH_SHAPE)[0])
    snaps = [int(v) for v in g["c1_snaps"]]
    for i, n in enumerate(snaps):
        out = host(M.register_pair_device(fix, mov, **dict(kw, selected_niter=n)))
        e = epe(np.moveaxis(out[:, ::s, ::s, ::s], 0, -1), np.moveaxis(g["c1_adam_%d_sub" % n], 0, -1))
        self_e = float(g["c1_self_perturbation_epe_sub"][i])
        print("full size, %2d Adam iterations: HIP vs reference mean EPE %.3e (reference vs its 1-ulp-perturbed self: %.3e)" % (n, e, self_e))
        if n == 1:
            assert e == 0.0
        elif n <= 40:
            assert e < 1e-3
        else:
            assert e <= self_e and e < 2e-3


@pytest.mark.timeout(1800)
def test_full_size_masked_large_motion_config3(M, U, orc, golden):
    """BASELINE configs[2]: 224x192x224 with ellipsoid masks, disp_hw 8 (4913-way search), gs 6, 20 Adam iterations.  The masked
    feature path + whole pipeline equal the oracle bit for bit; against the reference capture the convex stage is bit-identical and
    the final field is within the north-star tolerance."""
    from convexadam_amd.phantom import deformed_pair, ellipsoid_mask
    g = golden("fullsize")
    s = int(g["sub"])
    shape = (224, 192, 224)
    fix, mov = deformed_pair(shape, 3, 10.0)
    mf, mm = ellipsoid_mask(shape, 0.35), ellipsoid_mask(shape, 0.35, shift=(4, -3, 5))
    kw = dict(lambda_weight=1.25, grid_sp=6, disp_hw=8, selected_niter=20, selected_smooth=0, grid_sp_adam=2, ic=True)
    ff, fm = M.extract_features(fix, mov, 1, 2, True, mf, mm, device=torch.device(DEV), dtype=torch.float32)
    conv = M.register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], **dict(kw, lambda_weight=0))
    assert torch.equal(conv, U.resize_trilinear(dev(g["c3_coarse_ic"])[None], shape)[0])
    out = host(M.register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], **kw))
    e = epe(np.moveaxis(out[:, ::s, ::s, ::s], 0, -1), np.moveaxis(g["c3_adam_20_sub"], 0, -1))
    print("configs[2] full size, 20 Adam iterations: HIP vs reference mean EPE %.3e" % e)
    assert e < 1e-3
    # oracle: same masked features, same pipeline
    filled_f, _ = orc.replicate_fill(fix.numpy(), mf.numpy())
    filled_m, _ = orc.replicate_fill(mov.numpy(), mm.numpy())
    feats = (orc.mindssc(filled_f, 1, 2), orc.mindssc(filled_m, 1, 2))
    assert np.array_equal(host(ff)[0], feats[0]) and np.array_equal(host(fm)[0], feats[1])
    ref = orc.convex_adam_pipeline(None, None, features=feats, **kw)
    assert np.array_equal(np.moveaxis(out, 0, -1).astype(np.float64), ref)


# ---- (8) pipeline variants of the challenge scripts and fp16 storage ---------------------------------------------
@pytest.mark.parametrize("var", [dict(cost="sad", n_box=1, n_spline_pools=2), dict(n_box=1), dict(n_spline_pools=2), dict(cost="sad"),
                                 dict(storage="fp16"), dict(storage="fp16", n_spline_pools=2)])
def test_pipeline_variants_vs_oracle_bit_exact(M, orc, golden, var):
    """cost="sad" + n_box=1 + n_spline_pools=2 is the l2r_2021 task-3 configuration (task3_docker.py:54,56,191); storage="fp16" keeps the
    pooled features and the cost volume at half precision (the reference's GPU default dtype, convex_adam_MIND.py:79).  Rounding to
    half is deterministic, so even that mode is bit-identical to the oracle."""
    g = golden("pipeline")
    kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2, lambda_weight=1.25, selected_niter=4, ic=True)
    out = host(M.register_pair_device(dev(g["fix"]), dev(g["mov"]), **kw, **var))
    ref = orc.convex_adam_pipeline(g["fix"], g["mov"], **kw, **var)
    assert np.array_equal(np.moveaxis(out, 0, -1).astype(np.float64), ref), "EPE %g" % epe(np.moveaxis(out, 0, -1), ref)


@pytest.mark.parametrize("shape,hw,C", [((9, 8, 37), 3, 12), ((13, 16, 20), 4, 12), ((6, 7, 9), 2, 5), ((5, 40, 37), 2, 12), ((4, 23, 74), 1, 18)])
def test_fp16_storage_operators_vs_oracle(U, orc, shape, hw, C):
    """Real half-precision storage (convex_adam_MIND.py:79,89-91): `correlate(storage="fp16")` writes a torch.float16
    cost volume (float32 accumulation, one rounding) = the oracle's volume rounded to half, its argmin is the first minimum of the
    STORED values; `coupled_convex` solves on the half volume; `adam_run(storage="fp16")` keeps half-precision feature records."""
    rng = np.random.default_rng(sum(shape) + hw)
    f = rng.random((C,) + shape, dtype=np.float32)
    m = rng.random((C,) + shape, dtype=np.float32)
    ssd, am = U.correlate(dev(f)[None], dev(m)[None], hw, 1, shape, C, storage="fp16")
    assert ssd.dtype == torch.float16 and ssd.element_size() == 2
    ref, _ = orc.correlate(f, m, hw)
    ref_h = ref.astype(np.float16)
    assert np.array_equal(host(ssd), ref_h)
    assert np.array_equal(host(am), ref_h.astype(np.float32).reshape(ref.shape[0], -1).argmin(0).reshape(shape))
    mesh = orc.disp_mesh(hw)
    soft = U.coupled_convex(ssd, am, dev(mesh)[:, :, None], 1, shape)
    assert np.array_equal(host(soft)[0], orc.coupled_convex(ref_h.astype(np.float32), host(am), mesh, hw))
    # Adam loop on half-precision feature records
    shp2 = (shape[0] * 2, shape[1] * 2, shape[2] * 2)
    F2 = rng.random((C,) + shp2, dtype=np.float32)
    M2 = rng.random((C,) + shp2, dtype=np.float32)
    P0 = (0.5 * rng.standard_normal((3,) + shp2)).astype(np.float32)
    Ud, st = U.adam_run(dev(F2)[None], dev(M2)[None], dev(P0)[None], 1.25, 3, return_state=True, storage="fp16")
    h = lambda a: a.astype(np.float16).astype(np.float32)
    r = orc.adam_run(h(F2), h(M2), P0, 1.25, 3, want_grad=True)
    assert np.array_equal(host(Ud)[0], r["U"]) and np.array_equal(host(st["P"])[0], r["P"]) and np.array_equal(host(st["G"])[0], r["G"])


@pytest.mark.timeout(1800)
def test_full_size_fp16_storage_and_fast_mode_accuracy(M):
    """BASELINE configs[1] pair: the modes that are graded by accuracy instead of bits.  fp16 storage (features + cost volume) and the
    fast correlation mode move the final field by far less than a voxel; the fast mode leaves the convex stage untouched."""
    from convexadam_amd.phantom import deformed_pair
    fix, mov = [t.to(DEV) for t in deformed_pair(BENCH_SHAPE, 0, 4.0)]
    base = M.register_pair_device(fix, mov, **BENCH_CFG)
    conv = M.register_pair_device(fix, mov, **dict(BENCH_CFG, lambda_weight=0))
    conv_fast = M.register_pair_device(fix, mov, **dict(BENCH_CFG, lambda_weight=0), corr_mode="fast")
    flips = float((conv_fast != conv).float().mean())
    fast = M.register_pair_device(fix, mov, **BENCH_CFG, corr_mode="fast")
    h16 = M.register_pair_device(fix, mov, **BENCH_CFG, storage="fp16")
    conv16 = M.register_pair_device(fix, mov, **dict(BENCH_CFG, lambda_weight=0), storage="fp16")
    flips16 = float((conv16 != conv).any(0).float().mean())         # voxels whose convex-stage displacement changed (argmin flips show up here)
    e_conv16 = float((conv16 - conv).square().sum(0).sqrt().mean())
    e_fast = float((fast - base).square().sum(0).sqrt().mean())
    e_h16 = float((h16 - base).square().sum(0).sqrt().mean())
    print("full size: fast correlation mode: fraction of convex-stage voxels changed %.2e, final mean EPE vs exact %.3e; fp16 storage: convex-stage "
          "voxels changed %.2e (EPE %.3e), final EPE %.3e" % (flips, e_fast, flips16, e_conv16, e_h16))
    assert flips == 0.0 and e_fast < 1e-3
    assert e_h16 < 0.1 and e_conv16 < 0.1


def test_pipeline_snapshots_match_separate_runs(M, golden):
    """Row Q at pipeline level: one Adam run with snapshots after iterations 2 / 4 / 5 and final smoothings {none, 3, 5} returns the
    fields that separate runs with selected_niter = 2 / 4 / 5 and selected_smooth = 0 / 3 / 5 return (the 9-field variant of
    self_configuring/convex_adam_MIND.py:115-139)."""
    g = golden("pipeline")
    fix, mov = dev(g["fix"]), dev(g["mov"])
    kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2, lambda_weight=1.25, ic=True)
    snaps = M.register_pair_snapshots_device(fix, mov, snapshot_iters=(2, 4, 5), smooths=(0, 3, 5), **kw)
    assert snaps.shape == (3, 3) + (3,) + tuple(fix.shape)
    for i, n in enumerate((2, 4, 5)):
        for j, k in enumerate((0, 3, 5)):
            assert torch.equal(snaps[i, j], M.register_pair_device(fix, mov, selected_niter=n, selected_smooth=k, **kw)), (n, k)
    with pytest.raises(Exception):
        M.register_pair_snapshots_device(fix, mov, snapshot_iters=(4, 2), **kw)


@pytest.mark.timeout(1800)
def test_two_stage_sweep_on_the_device(tmp_path):
    """The reference's self-configuring procedure end to end on one GPU (single process): stage 1 ranks convex-only settings, stage 2 runs
    ONE 120-iteration Adam optimisation per (setting, pair) from the stage-1 field -- smoother and grid_sp_adam per setting, n_ch cost
    scale -- and scores its four snapshots x four smoothings; results are appended per item and the known roll is recovered."""
    import json
    from convexadam_amd import sweep
    out = tmp_path / "sweep.json"
    assert sweep.main(["--pairs", "1", "--shape", "48", "48", "48", "--stage1", "3", "--stage2", "2", "--out", str(out)]) == 0
    s = json.loads(out.read_text())
    assert s["stage1"]["n_items"] == 3 and s["stage2"]["n_items"] == 2 and s["stage2"]["evaluations"] == 32
    lines = [json.loads(l) for l in open(str(out) + ".rank0.jsonl")]
    assert "header" in lines[0] and s["workers_per_rank"] == 3          # three items in flight per GPU by default since round 5 (they hide each other's host side)
    lines = lines[1:]
    assert len(lines) == 5 and all("ms" in l or "evals" in l for l in lines)
    conv = [l for l in lines if l["stage"] == "convex"]
    assert max(l["dice"] for l in conv) > conv[0]["dice_before"] + 0.1
    best = max(e["dice"] for l in lines if l["stage"] == "adam" for e in l["evals"])
    assert best >= max(l["dice"] for l in conv) - 0.02 and best > 0.8


@pytest.mark.timeout(1800)
def test_full_size_sweep_extreme_settings(M):
    """BASELINE configs[4] at 160x192x224 with the two extremes of the setting grid: grid_sp 4 / disp_hw 6 (2197 x 40x48x56 = 945 MB
    cost volume per direction) and grid_sp 8 / disp_hw 3.  Both run through the same entry point, recover the known shift and stay
    bit-reproducible; their cost ratio is what the sweep's work queue orders items by."""
    import time
    from convexadam_amd import sweep
    from convexadam_amd.phantom import phantom
    shape = (160, 192, 224)
    fix = phantom(shape, 100, 200).to(DEV)
    mov = torch.roll(phantom(shape, 100, 300), sweep.SHIFT, (0, 1, 2)).to(DEV)
    ms = {}
    for name, cfg in (("gs4_hw6", dict(grid_sp=4, disp_hw=6)), ("gs8_hw3", dict(grid_sp=8, disp_hw=3))):
        kw = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp_adam=2, selected_niter=20, ic=True, **cfg)
        a = M.register_pair_device(fix, mov, **kw)
        torch.cuda.synchronize()
        t = time.time()
        b = M.register_pair_device(fix, mov, **kw)
        torch.cuda.synchronize()
        ms[name] = (time.time() - t) * 1e3
        assert torch.equal(a, b)
        c = a[:, 32:128, 38:154, 45:179]
        for ax in range(3):
            assert abs(float(c[ax].mean()) - sweep.SHIFT[ax]) < 0.5, (name, ax, float(c[ax].mean()))
    print("full-size extreme sweep settings: %s ms" % ms)
    assert sweep.item_cost(dict(grid_sp=4, disp_hw=6), shape) > sweep.item_cost(dict(grid_sp=8, disp_hw=3), shape)


def test_device_tables_equal_the_host_helpers_and_torch():
    """SURVEY 8(a) row F and the affine identity tables AS THE PIPELINE BUILDS THEM (on the device, never uploaded): cvx_disp_mesh_f32 /
    cvx_affine_base_f32 equal the host helpers bit for bit, which equal torch's affine_grid (tests/test_host_logic.py)."""
    import torch.nn.functional as Fn
    from convexadam_amd import _lib
    from convexadam_amd.convex_adam_utils import affine_base, disp_mesh
    L = _lib.lib()
    sp = _lib.stream_ptr(torch.device(DEV))
    for hw in (0, 1, 2, 3, 6, 8, 11, 15):
        n = 2 * hw + 1
        out = torch.empty((3, n ** 3), dtype=torch.float32, device=DEV)
        assert L.cvx_disp_mesh_f32(hw, _lib.ptr(out), sp) == 0
        assert np.array_equal(host(out), disp_mesh(hw)), hw
        if hw:
            m = Fn.affine_grid(hw * torch.eye(3, 4)[None], (1, 1, n, n, n), align_corners=True).permute(0, 4, 1, 2, 3).reshape(3, -1)
            assert np.array_equal(host(out), m.numpy()), hw
    for S in (1, 2, 3, 26, 37, 80, 112, 224, 1000):
        out = torch.empty((S,), dtype=torch.float32, device=DEV)
        assert L.cvx_affine_base_f32(S, _lib.ptr(out), sp) == 0
        assert np.array_equal(host(out), affine_base(S)), S
    assert L.cvx_disp_mesh_f32(99, _lib.ptr(out), sp) != 0 and L.cvx_affine_base_f32(0, _lib.ptr(out), sp) != 0


# ---- (9) every selectable kernel variant -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opt,val", [("mind_tiled", 1), ("mm_tx", 32), ("mm_tx", 64), ("mm_slots", 64), ("box_tiled", 1), ("no_prune", 1),
                                     ("corr_unfused", 1), ("prune_stream_above", 0), ("cf_census", 1), ("cf_prio", 0x9d), ("warp_flat", 1), ("box_yt", 4), ("box_wg_target", 700), ("box_xsplit", 0), ("box_cpt", 2), ("box_uneven", 100), ("box_prio", 1), ("mind_overlap", 1), ("corr_fused_all", 1), ("box_fwd_tile", 0), ("box_fwd_tile", 1000), ("box_fwd_tile", 2000), ("corr_dual", 1), ("prune_refine", 0), ("mind_records", 0), ("mind_blocked", 0), ("mind_single", 1), ("mind_single", 2), ("cf_map", 0), ("resize_up2", 0), ("box_walk", 0), ("box_bwd_tile", 0), ("box_bwd_tile", 1000), ("box_bwd_tile", 2000)])
def test_kernel_variants_agree(M, U, orc, golden, opt, val):
    """The library's run-time switches (cvx_set_option / CVX_* environment variables) select alternative kernels for the same
    operators; every one of them is bit-identical to the oracle: marching vs tiled MIND stencil and its tile shapes, marching vs tiled
    three-box kernels, branch-and-bound vs streaming coupled-convex passes (and the pruned pass forced onto its streaming fallback),
    fused vs unfused correlation."""
    from convexadam_amd import _lib
    from convexadam_amd.phantom import phantom
    L = _lib.lib()
    old = L.cvx_get_option(opt.encode())
    assert L.cvx_set_option(opt.encode(), val) == 0 and L.cvx_get_option(opt.encode()) == val
    try:
        g = golden("pipeline")
        kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2, lambda_weight=1.25, selected_niter=3, ic=True)
        out = M.convex_adam_pt(g["fix"], g["mov"], dtype=torch.float32, device=torch.device(DEV), **kw)
        assert np.array_equal(out, orc.convex_adam_pipeline(g["fix"], g["mov"], **kw))
        # a pair whose rows are multiples of 4 voxels (marching MIND stencil, fused pooling) and a wider search
        shape = (38, 27, 52)
        fix = phantom(shape, 11, 21)
        mov = torch.roll(phantom(shape, 11, 22), (1, -1, 2), (0, 1, 2))
        kw2 = dict(mind_r=1, mind_d=2, grid_sp=6, disp_hw=5, grid_sp_adam=2, lambda_weight=1.25, selected_niter=2, ic=True)
        out2 = M.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device(DEV), **kw2)
        assert np.array_equal(out2, orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw2))
        # flat cost regions through the stand-alone operator
        rng = np.random.default_rng(3)
        cshape, hw = (6, 8, 12), 3
        K = (2 * hw + 1) ** 3
        ssd = rng.random((K,) + cshape, dtype=np.float32)
        ssd[:, :, :5, :] = 0.25
        am = ssd.reshape(K, -1).argmin(0).reshape(cshape).astype(np.int64)
        mesh = orc.disp_mesh(hw)
        soft = U.coupled_convex(dev(ssd), dev(am), dev(mesh)[:, :, None], 1, cshape)
        assert np.array_equal(host(soft)[0], orc.coupled_convex(ssd, am, mesh, hw))
    finally:
        L.cvx_set_option(opt.encode(), old)
    assert L.cvx_set_option(b"no_such_switch", 1) != 0


def test_translation_wrapper_and_original_moving_warp_on_metaimage_files(tmp_path):
    """SURVEY 8(f).3 end to end without SimpleITK: MetaImage files in, convex_adam_translation_from_file registers on a 1 mm grid (HIP),
    reduces the field to a whole-voxel translation and writes the moved image; apply_convex_original_moving carries a field to an
    anisotropic moving image and warps it on the device."""
    from convexadam_amd.apply_convex import apply_convex_original_moving
    from convexadam_amd.convex_adam_translation import convex_adam_translation_from_file
    from convexadam_amd.convex_adam_utils import resample_img
    from convexadam_amd.imageio import Image, read_mha, write_mha
    from convexadam_amd.phantom import phantom
    vol = phantom((48, 40, 44), 5, 50).numpy()
    shift_zyx = (3, 0, -2)
    fixed = Image(vol, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
    moving = Image(np.roll(vol, shift_zyx, (0, 1, 2)), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
    pf, pm, po = str(tmp_path / "fixed.mha"), str(tmp_path / "moving.mha"), str(tmp_path / "moved.mha")
    write_mha(fixed, pf); write_mha(moving, pm, compress=True)
    t_xyz = convex_adam_translation_from_file(pf, pm, None, po)
    assert tuple(t_xyz) == (float(shift_zyx[2]), float(shift_zyx[1]), float(shift_zyx[0]))      # fixed(x) ~ moving(x + u): u = +shift
    moved = read_mha(po)
    assert np.array_equal(moved.array, moving.array) and np.allclose(moved.GetOrigin(), (-np.array(t_xyz)).tolist())
    # field on the 1 mm grid of an anisotropic fixed image -> warp of the original (anisotropic) moving image
    fixed_a = Image(vol[::2].copy(), (1.0, 1.0, 2.0), (0.0, 0.0, 0.0))
    moving_a = Image(np.roll(vol, shift_zyx, (0, 1, 2))[::2].copy(), (1.0, 1.0, 2.0), (0.0, 0.0, 0.0))
    fixed_1mm = resample_img(fixed_a, (1.0, 1.0, 1.0))
    field = np.zeros(fixed_1mm.array.shape + (3,))
    field[..., 0], field[..., 2] = 4.0, -2.0                                                    # z, y, x in 1 mm voxels
    warped = apply_convex_original_moving(field, moving_a, fixed_a, fixed_1mm)
    assert isinstance(warped, Image) and warped.array.dtype == np.float32 and warped.GetSpacing() == moving_a.GetSpacing()
    inner = (slice(4, -4),) * 3
    assert np.allclose(warped.array[inner], np.roll(moving_a.array, (-2, 0, 2), (0, 1, 2))[inner], atol=1e-4)   # 4 mm = 2 voxels along z


def test_adam_operator_bit_identical_to_reference_with_mkl_sqrt_table(U, orc, golden):
    """cvx_set_adam_sqrt_table: with the tabulated deviation of the reference build's sqrt (MKL vsSqrt) the HIP Adam loop reproduces
    the REFERENCE's captured control grid, gradient and disp_sample bit for bit at every horizon of the golden file (1, 2, 5, 20
    iterations) -- the loop's only non-restated site is then restated too -- and still equals the oracle run with the same table."""
    g, t = golden("adam"), golden("mkl_vssqrt_low")
    args = (dev(g["F2"])[None], dev(g["M2"])[None], dev(g["P0"])[None], float(g["lam"]))
    U.set_adam_sqrt_table(t["normal"], t["denormal"], device=DEV)
    orc.set_sqrt_table(t["normal"], t["denormal"])
    try:
        for niter in (1, 2, 5, 20):
            Ud, st = U.adam_run(*args, niter, return_state=True)
            assert np.array_equal(host(Ud)[0], g["U_%d" % niter]) and np.array_equal(host(st["G"])[0], g["G_%d" % niter]), niter
            assert np.array_equal(host(st["P"])[0], g["P_%d" % niter]), niter
        r = orc.adam_run(g["F2"], g["M2"], g["P0"], float(g["lam"]), 20, want_grad=True)
        assert np.array_equal(host(st["P"])[0], r["P"]) and np.array_equal(host(st["v"])[0], r["v"])
        # the generic-smoother path and the tiled box kernels take the same update
        from convexadam_amd import _lib
        _lib.lib().cvx_set_option(b"box_tiled", 1)
        Ut, stt = U.adam_run(*args, 5, return_state=True)
        _lib.lib().cvx_set_option(b"box_tiled", 0)
        assert np.array_equal(host(stt["P"])[0], g["P_5"])
    finally:
        U.set_adam_sqrt_table(None)
        orc.set_sqrt_table(None)
    assert not np.array_equal(host(U.adam_run(*args, 20, return_state=True)[1]["P"])[0], g["P_20"])   # default: IEEE sqrt


# ---- reference-bits mode: the reference build's exp / sqrt as tables (convexadam_amd/reference_bits.py, tests/mkl_tables.py) --------
@pytest.fixture()
def reference_bits(orc, mkl):
    """HIP library AND oracle with the tables of the host that produced the goldens."""
    from convexadam_amd import reference_bits as rb
    t = mkl.golden_tables()
    rb.set_mind_exp_table(t["exp"], t["exp_first"], t["exp_count"], device=DEV)
    rb.set_adam_sqrt_table(t["sqrt"], device=DEV)
    rb.set_mean_threads(8)                      # every golden was captured with torch.set_num_threads(8)
    orc.set_exp_table(t["exp"], t["exp_first"], t["exp_count"])
    orc.set_sqrt_table(t["sqrt"])
    orc.set_mean_threads(8)
    yield rb
    rb.disable()
    orc.set_exp_table(None)
    orc.set_sqrt_table(None)
    orc.set_mean_threads(0)


def test_reference_bits_tables_built_by_the_product_equal_the_test_infrastructure(orc, mkl):
    """reference_bits.build_exp_table tabulates (this host's torch.exp) - (the DEVICE expf) over all 310 M arguments of the domain;
    tests/mkl_tables.py does the same against the oracle's expf on the CPU.  Equal tables = the device expf equals the oracle's for
    every argument MINDSSC can produce, and the product's builder is right.  Same for the sqrt bit maps."""
    from convexadam_amd import reference_bits as rb
    t = mkl.host_tables(orc)
    assert (rb.EXP_FIRST, rb.EXP_COUNT) == (t["exp_first"], t["exp_count"])
    assert np.array_equal(host(rb.build_exp_table(DEV)), t["exp"])
    assert np.array_equal(rb.build_sqrt_table(), t["sqrt"])
    x = -torch.rand(100000, device=DEV) * 110.0
    assert np.array_equal(host(rb.device_expf(x)), orc.expf(host(x)))


def test_mindssc_bit_identical_to_reference_with_exp_table(U, orc, golden, reference_bits):
    """With the golden host's exp table MINDSSC equals the reference golden exactly (default: <= 1 ulp), for every radius / dilation of
    the golden file, through the marching, the tiled and the pooled kernels; ragged shapes against the oracle with the same table."""
    from convexadam_amd.phantom import phantom
    from convexadam_amd import _lib
    g = golden("mind")
    for key, r, d in (("mind_r1d2", 1, 2), ("mind_r2d2", 2, 2), ("mind_r1d1", 1, 1)):
        assert np.array_equal(host(U.MINDSSC(dev(g["img"])[None, None], r, d, device=DEV))[0], g[key]), key
    _lib.lib().cvx_set_option(b"mind_tiled", 1)
    assert np.array_equal(host(U.MINDSSC(dev(g["img"])[None, None], 1, 2, device=DEV))[0], g["mind_r1d2"])
    _lib.lib().cvx_set_option(b"mind_tiled", 0)
    for shape in ((20, 18, 23), (33, 40, 70), (9, 70, 8)):
        img = phantom(shape, 3, 30)
        img[: shape[0] // 2] *= 1e-3                                   # small variances: clamped voxels, large arguments of exp
        assert np.array_equal(host(U.MINDSSC(img[None, None].to(DEV), 1, 2, device=DEV))[0], orc.mindssc(img.numpy(), 1, 2)), shape


def test_two_threads_with_different_contexts_run_concurrently(M, orc, mkl):
    """the C ABI is re-entrant per stream and holds no shared mutable state.  Two Python threads register the same pair
    at the same time on two streams -- one with the default context, one inside a reference-bits context (golden host's exp / sqrt
    tables, torch's 8-thread mean, another box-kernel variant) -- and each gets exactly its own mode's field, which is the oracle's
    field for that mode.  A third context selected per call through cvx_pair_params.ctx-style binding is nested inside."""
    import threading
    from convexadam_amd import reference_bits as rb
    from convexadam_amd.context import Context
    from convexadam_amd.phantom import deformed_pair, ellipsoid_mask
    t = mkl.golden_tables()
    shape = (48, 56, 64)
    fix, mov = deformed_pair(shape, 5, 3.0)
    m = ellipsoid_mask(shape, 0.4)
    fix, mov = (fix * m).contiguous(), (mov * m).contiguous()          # flat background: clamped variances -> the mean matters
    kw = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=4, disp_hw=3, selected_niter=12, selected_smooth=0, grid_sp_adam=2, ic=True)
    want_default = np.moveaxis(orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw), -1, 0).astype(np.float32)
    orc.set_exp_table(t["exp"], t["exp_first"], t["exp_count"]); orc.set_sqrt_table(t["sqrt"]); orc.set_mean_threads(8)
    try:
        want_bits = np.moveaxis(orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw), -1, 0).astype(np.float32)
    finally:
        orc.set_exp_table(None); orc.set_sqrt_table(None); orc.set_mean_threads(0)
    assert not np.array_equal(want_default, want_bits), "the two modes must differ for this test to mean anything"
    ctx = rb.context(DEV, threads=8, exp_table=t["exp"], sqrt_table=t["sqrt"], exp_first=t["exp_first"], exp_count=t["exp_count"])
    ctx.set_option("box_tiled", 1)
    fd, md = fix.to(DEV), mov.to(DEV)
    results, errors = {}, []
    start = threading.Barrier(2)

    def worker(name, context, n):
        try:
            s = torch.cuda.Stream(DEV)
            outs = []
            start.wait()
            with torch.cuda.stream(s):
                for _ in range(n):
                    if context is None:
                        outs.append(M.register_pair_device(fd, md, **kw).clone())
                    else:
                        with context:
                            outs.append(M.register_pair_device(fd, md, **kw).clone())
                s.synchronize()
            results[name] = [host(o) for o in outs]
        except Exception as e:                                   # surfaced by the assertion below
            errors.append((name, repr(e)))

    ths = [threading.Thread(target=worker, args=("default", None, 6)), threading.Thread(target=worker, args=("bits", ctx, 6))]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    for o in results["default"]:
        assert np.array_equal(o, want_default)
    for o in results["bits"]:
        assert np.array_equal(o, want_bits)
    # the default context was never touched
    from convexadam_amd import _lib
    assert _lib.lib().cvx_get_option(b"mind_mean_threads") == 0 and _lib.lib().cvx_get_option(b"box_tiled") == 0
    assert np.array_equal(host(M.register_pair_device(fd, md, **kw)), want_default)
    ctx.close()


def test_pipeline_goldens_bit_identical_to_reference_with_mkl_tables(M, golden, reference_bits):
    """Whole pipelines of tests/golden/pipeline.npz (convex only, 1 / 5 / 20 Adam iterations, final smoothing, no ic): exact."""
    g = golden("pipeline")
    kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2)
    for key, extra in (("convex_only_ic", dict(lambda_weight=0, ic=True)), ("adam_1", dict(lambda_weight=1.25, selected_niter=1, ic=True)),
                       ("adam_5", dict(lambda_weight=1.25, selected_niter=5, ic=True)), ("adam_20", dict(lambda_weight=1.25, selected_niter=20, ic=True)),
                       ("adam_5_smooth3", dict(lambda_weight=1.25, selected_niter=5, selected_smooth=3, ic=True)),
                       ("adam_5_noic", dict(lambda_weight=1.25, selected_niter=5, ic=False))):
        out = host(M.register_pair_device(dev(g["fix"]), dev(g["mov"]), **kw, **extra))
        assert np.array_equal(np.moveaxis(out, 0, -1).astype(np.float64), g[key]), key


def test_full_size_benchmark_pair_bit_identical_to_the_reference_with_mkl_tables(M, golden, reference_bits):
    """THE headline configuration (BASELINE configs[1], the pair bench.py times: 160x192x224, hw 6, gs 6, ic, 80 Adam iterations): with
    the two tables of the golden host the HIP pipeline's field equals the field captured from the reference itself -- every 8th voxel
    per axis bit for bit at 1, 20, 40 and 80 iterations, float64 sums of the whole field and of its squares to 1e-14."""
    from convexadam_amd.phantom import deformed_pair
    g = golden("fullsize")
    s = int(g["sub"])
    shape = (160, 192, 224)
    fix, mov = deformed_pair(shape, 0, 4.0)
    kw = dict(mind_r=1, mind_d=2, grid_sp=6, disp_hw=6, grid_sp_adam=2, ic=True, lambda_weight=1.25)
    for niter in (int(v) for v in g["c1_snaps"]):
        f = M.register_pair_device(fix.to(DEV), mov.to(DEV), selected_niter=niter, **kw)
        assert np.array_equal(host(f[:, ::s, ::s, ::s]), g["c1_adam_%d_sub" % niter]), niter
        fd = f.cpu().double()
        assert np.allclose(fd.sum((1, 2, 3)).numpy(), g["c1_adam_%d_sum" % niter], rtol=1e-14, atol=0)
        assert np.allclose(fd.square().sum((1, 2, 3)).numpy(), g["c1_adam_%d_sumsq" % niter], rtol=1e-14, atol=0)


@pytest.mark.parametrize("threads", [1, 2, 8, 128])
def test_mindssc_with_torch_mean_vs_oracle(U, orc, reference_bits, threads):
    """Option mind_mean_threads: the global mean as torch sums it with T threads (two-pass reduction over chunks, 8-float vectors, 4
    interleaved cascade accumulators) -- HIP vs the oracle's restatement (itself pinned against torch.sum for 1..128 threads,
    tests/test_host_logic.py) on volumes with clamped voxels, below and above the 32768-element threshold of the parallel path."""
    from convexadam_amd.phantom import phantom
    reference_bits.set_mean_threads(threads)
    orc.set_mean_threads(threads)
    for shape in ((20, 18, 23), (33, 40, 70), (64, 72, 60)):
        img = phantom(shape, 3, 30)
        img[: shape[0] // 2] *= 1e-3
        out, mean = orc.mindssc(img.numpy(), 1, 2, return_mean=True)
        assert np.array_equal(host(U.MINDSSC(img[None, None].to(DEV), 1, 2, device=DEV))[0], out), (shape, threads)


def test_masked_goldens_bit_identical_to_reference_in_reference_bits_mode(M, golden, reference_bits):
    """extract_features(use_mask=True) of tests/golden/masked.npz: exact (flat filled regions -> clamped variances -> the mean's last bits)."""
    g = golden("masked")
    ff, fm = M.extract_features(dev(g["img_fix"]), dev(g["img_mov"]), 1, 2, True, dev(g["mask_fix"]), dev(g["mask_mov"]), device=torch.device(DEV),
                                dtype=torch.float32)
    assert np.array_equal(host(ff)[0], g["feat_fix"]) and np.array_equal(host(fm)[0], g["feat_mov"])


def test_full_size_masked_config3_bit_identical_to_the_reference(M, golden, reference_bits):
    """BASELINE configs[2] (224x192x224, masks, disp_hw 8, 20 Adam iterations) in reference-bits mode: equal to the reference capture."""
    from convexadam_amd.phantom import deformed_pair, ellipsoid_mask
    g = golden("fullsize")
    s = int(g["sub"])
    shape = (224, 192, 224)
    fix, mov = deformed_pair(shape, 3, 10.0)
    mf, mm = ellipsoid_mask(shape, 0.35), ellipsoid_mask(shape, 0.35, shift=(4, -3, 5))
    ff, fm = M.extract_features(fix, mov, 1, 2, True, mf, mm, device=torch.device(DEV), dtype=torch.float32)
    f = M.register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], lambda_weight=1.25, grid_sp=6, disp_hw=8, selected_niter=20,
                               selected_smooth=0, grid_sp_adam=2, ic=True)
    assert np.array_equal(host(f[:, ::s, ::s, ::s]), g["c3_adam_20_sub"])
    fd = f.cpu().double()
    assert np.allclose(fd.sum((1, 2, 3)).numpy(), g["c3_adam_20_sum"], rtol=1e-14, atol=0)
    assert np.allclose(fd.square().sum((1, 2, 3)).numpy(), g["c3_adam_20_sumsq"], rtol=1e-14, atol=0)


def test_reference_bits_enable_with_this_hosts_own_tables_vs_oracle(M, orc, mkl, golden):
    """reference_bits.enable() tabulates THIS host's torch.exp / torch.sqrt (on the GPU boxes an EPYC, whose MKL code path deviates from
    the IEEE root in BOTH directions and from the golden host's exp in 40 % of the deviating arguments) and installs torch's mean for
    this process's thread count.  The HIP pipeline then equals the oracle given the tables tests/mkl_tables.py builds from the same
    host -- i.e. what the reference would compute HERE -- and differs from the golden host's result."""
    from convexadam_amd import reference_bits as rb
    t = mkl.host_tables(orc)
    g = golden("pipeline")
    kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2, lambda_weight=1.25, selected_niter=20, ic=True)
    threads = torch.get_num_threads()
    rb.enable(DEV)
    orc.set_exp_table(t["exp"], t["exp_first"], t["exp_count"])
    orc.set_sqrt_table(t["sqrt"])
    orc.set_mean_threads(threads)
    try:
        out = host(M.register_pair_device(dev(g["fix"]), dev(g["mov"]), **kw))
        ref = orc.convex_adam_pipeline(g["fix"], g["mov"], **kw)
    finally:
        rb.disable()
        orc.set_exp_table(None)
        orc.set_sqrt_table(None)
        orc.set_mean_threads(0)
    assert np.array_equal(np.moveaxis(out, 0, -1).astype(np.float64), ref)
    if not t["matches_golden_host"]:
        assert not np.array_equal(ref, g["adam_20"])            # another host, another reference result


def test_even_selected_smooth_vs_oracle_and_reference(M, U, orc, golden):
    """The reference's even `selected_smooth` (convex_adam_MIND.py:184-191: three growing pools, (H+3, W+3, D+3, 3)) through the drop-in
    call: cvx_box_grow_f32 == the oracle's pool == torch's (golden), the returned field == the oracle's bit for bit and the reference's
    capture within the short-horizon tolerance; the whole-pair C entry keeps refusing (its output is [3][H][W][D])."""
    from convexadam_amd import _lib
    from convexadam_amd.phantom import phantom
    g = golden("even_smooth")
    x = dev(g["pool_in"])[None]
    for k in (2, 4, 6):
        assert np.array_equal(host(U.box_smooth(x, k, 1))[0], g["pool%d" % k]), k
    assert tuple(U.box_smooth(x, 2, 3).shape) == (1, 3) + tuple(s + 3 for s in g["pool_in"].shape[1:])
    shape = tuple(int(v) for v in g["shape"])
    fix = phantom(shape, 7, 70)
    mov = torch.roll(phantom(shape, 7, 71), (1, -1, 2), (0, 1, 2))
    kw = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=4, disp_hw=2, selected_niter=3, grid_sp_adam=2, ic=True)
    for k in (2, 4):
        out = M.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device(DEV), selected_smooth=k, **kw)
        ref = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), selected_smooth=k, **kw)
        assert out.shape == tuple(s + 3 for s in shape) + (3,) and out.dtype == np.float64
        assert np.array_equal(out, ref), k
        assert float(np.sqrt(((out - g["k%d" % k]) ** 2).sum(-1)).mean()) < 1e-5
    out0 = M.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device(DEV), lambda_weight=0, grid_sp=4, disp_hw=2, selected_smooth=2)
    assert out0.shape == shape + (3,)                                                    # never reaches the smoothing block (:155)
    with pytest.raises(_lib.CvxError):
        M.register_pairs_device([dev(fix.numpy())], [dev(mov.numpy())], selected_smooth=2, **kw)
