"""Single-pass pooled MIND-SSC (mindmarch.hip::k_mind_march_pool + mind.hip::k_mind_repair) through the C ABI (cvx_mindssc_pooled_f32):
avg_pool3d(MINDSSC(img), g, stride g) for both window sizes of a pair without the raw-SSD round trip.  Bit-exact (np.array_equal) against the
oracle's mindssc + avgpool_stride and against the two-pass kernels, on inputs chosen so that the variance clamp of convex_adam_utils.py:60-62
binds (low and high), on exact-zero backgrounds (no repair needed), on a constant image (mean 0: NaN descriptors) and with every block forced
through the repair kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def U():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from convexadam_amd import convex_adam_utils
    return convex_adam_utils


@pytest.fixture()
def single():
    """sets option mind_single for the test and restores the default"""
    from convexadam_amd import _lib
    L = _lib.lib()
    old = L.cvx_get_option(b"mind_single"), L.cvx_get_option(b"ms_zlen")

    def set_(v, zlen=0):
        assert L.cvx_set_option(b"mind_single", v) == 0 and L.cvx_set_option(b"ms_zlen", zlen) == 0
    yield set_
    L.cvx_set_option(b"mind_single", old[0]); L.cvx_set_option(b"ms_zlen", old[1])


def same(a, b):
    """equal values, NaNs in the same places (0 / 0 is 0x7fc00000 on the GPU and 0xffc00000 on an x86 host: the payload is not compared)"""
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def oracle_pooled(orc, img, g1, g2):
    m = orc.mindssc(img, 1, 2)
    return orc.avgpool_stride(m, g1), (orc.avgpool_stride(m, g2) if g2 else None)


def run(U, img, g1, g2):
    out = U.mind_pooled(dev(img)[None, None], 1, 2, g1, g2, device=DEV, return_repairs=True)
    return host(out[0])[0], (host(out[1])[0] if g2 else None), out[-1]


def textured(shape, seed):
    return (np.random.default_rng(seed).standard_normal(shape) * 10).astype(np.float32)


WINDOWS = [(6, 2), (6, 3), (6, 6), (4, 2), (4, 4), (2, 2), (6, 0), (4, 0), (2, 0), (2, 6), (2, 4)]


@pytest.mark.parametrize("g1,g2", WINDOWS)
@pytest.mark.parametrize("shape", [(24, 24, 84), (13, 19, 92), (40, 30, 172), (7, 6, 8), (50, 13, 256)])
def test_single_pass_vs_oracle(U, orc, single, shape, g1, g2):
    """Every window pair of the kernel on tiles that overhang in y and x, partial blocks, several z chunks with a cut last window, planes
    fewer than a window, the last < 32 voxels of the volume."""
    img = textured(shape, shape[0] + shape[2] + g1)
    single(1)
    o1, o2, nrep = run(U, img, g1, g2)
    r1, r2 = oracle_pooled(orc, img, g1, g2)
    assert same(o1, r1), "g1: max |diff| %g" % np.nanmax(np.abs(o1 - r1))
    assert g2 == 0 or same(o2, r2), "g2: max |diff| %g" % np.nanmax(np.abs(o2 - r2))
    assert nrep == 0                          # white noise: the clamp never binds


@pytest.mark.parametrize("r,d", [(2, 2), (1, 1), (3, 1), (1, 3)])
@pytest.mark.parametrize("g1,g2", [(6, 2), (4, 2), (2, 0)])
def test_other_radii_take_two_passes_through_the_scratch(U, orc, single, r, d, g1, g2):
    """cvx_mindssc_pooled_f32 outside the marching stencil's setting (radius 1, dilation 2): the tiled stencil + the fused normalise / pool pass through
    the caller's scratch (cvx_mindssc_pooled_scratch_bytes > 0) -- same operator, same bits as MINDSSC + avg_pool3d."""
    from convexadam_amd._lib import lib
    img = textured((18, 20, 28), 3 * r + d)
    single(1)
    assert lib().cvx_mindssc_pooled_scratch_bytes(18, 20, 28, r, d, g1, g2) >= 12 * 18 * 20 * 28 * 4
    out = U.mind_pooled(dev(img)[None, None], r, d, g1, g2, device=DEV, return_repairs=True)
    m = orc.mindssc(img, r, d)
    assert same(host(out[0])[0], orc.avgpool_stride(m, g1)) and out[-1] == 0
    if g2:
        assert same(host(out[1])[0], orc.avgpool_stride(m, g2))


@pytest.mark.parametrize("zlen", [6, 12, 18, 36])
def test_z_chunk_lengths(U, orc, single, zlen):
    img = textured((47, 20, 100), 5)
    single(1, zlen)
    o1, o2, _ = run(U, img, 6, 2)
    r1, r2 = oracle_pooled(orc, img, 6, 2)
    assert same(o1, r1) and same(o2, r2)


def clamp_cases():
    rng = np.random.default_rng(11)
    # (a) exact-zero background around a textured body: all-zero voxels need no repair, the boundary voxels with tiny variance do
    a = np.zeros((30, 36, 88), np.float32)
    a[8:22, 9:27, 20:70] = rng.normal(100, 30, (14, 18, 50)).astype(np.float32)
    # (b) nearly flat region with noise far below the texture elsewhere: variance below 0.001 x mean on whole blocks
    b = rng.normal(0, 1e-3, (30, 36, 88)).astype(np.float32)
    b[:, :, 44:] += rng.normal(0, 50, (30, 36, 44)).astype(np.float32)
    # (c) a sparse volume: a few bright voxels in noise eight orders of magnitude weaker -> variances above 1000 x mean
    c = rng.normal(0, 1e-4, (30, 36, 88)).astype(np.float32)
    c[15, 18, 40] = 500.0
    c[3, 30, 7] = -300.0
    # (d) piecewise constant: exact zeros of the distances inside the pieces, equal distances (ties in the min) at the faces
    d = np.zeros((30, 36, 88), np.float32)
    d[:, 12:, :] = 3.0
    d[:, :, 50:] += 7.0
    d[20:] += 1.0
    return {"zero_background": a, "flat_half": b, "sparse": c, "pieces": d}


@pytest.mark.parametrize("name", ["zero_background", "flat_half", "sparse", "pieces"])
@pytest.mark.parametrize("g1,g2", [(6, 2), (4, 2), (6, 3), (2, 2)])
def test_clamp_binds(U, orc, single, name, g1, g2):
    img = clamp_cases()[name]
    single(1)
    o1, o2, nrep = run(U, img, g1, g2)
    r1, r2 = oracle_pooled(orc, img, g1, g2)
    assert same(o1, r1) and same(o2, r2), "%s: %d / %d cells differ" % (name, (o1 != r1).sum(), (o2 != r2).sum())
    single(0)
    t1, t2, zero = run(U, img, g1, g2)
    assert same(t1, r1) and same(t2, r2) and zero == 0
    assert nrep > 0 or name in ("pieces", "zero_background"), "the clamp binds on this input: the repair list cannot be empty"


def test_all_zero_voxels_need_no_repair(U, orc, single):
    """An exact-zero background whose boundary is kept away from the body by more than the stencil: the body is white noise (no clamp), the
    rest has twelve zero distances -> nothing to repair, descriptors 1."""
    img = np.zeros((24, 24, 84), np.float32)
    img[:, :, :] = 0.0
    single(1)
    img2 = img.copy()
    img2[:, :, 40:] = textured((24, 24, 44), 3)
    o1, o2, nrep = run(U, img2, 6, 2)
    r1, r2 = oracle_pooled(orc, img2, 6, 2)
    assert same(o1, r1) and same(o2, r2)
    assert np.all(o2[:, :, :, :16] == 1.0)


def test_constant_image_gives_nan_like_the_reference(U, orc, single):
    """mean variance 0 -> clamp bounds 0 -> 0 / 0 on every voxel (convex_adam_utils.py:62-63): every block goes through the repair kernel."""
    img = np.full((12, 12, 84), 2.5, np.float32)
    single(1)
    o1, o2, nrep = run(U, img, 6, 2)
    r1, r2 = oracle_pooled(orc, img, 6, 2)
    assert np.isnan(r1).all() and same(o1, r1) and same(o2, r2)
    assert nrep == 2 * 2 * 14


@pytest.mark.parametrize("g1,g2", [(6, 2), (6, 3), (6, 6), (4, 2), (4, 4), (2, 2), (6, 0)])
def test_every_block_through_the_repair_kernel(U, orc, single, g1, g2):
    """mind_single = 2 forces the exact recomputation of every block: the repair kernel alone reproduces the oracle."""
    img = textured((19, 22, 92), 9)
    img[:5] = 0.0
    single(2)
    o1, o2, nrep = run(U, img, g1, g2)
    r1, r2 = oracle_pooled(orc, img, g1, g2)
    assert same(o1, r1) and (g2 == 0 or same(o2, r2))
    assert nrep == int(np.prod([-(-s // g1) for s in img.shape]))


def test_full_size_against_the_two_pass_kernels(U, single):
    """BASELINE configs[1] size: the benchmark phantom and its zero-background variant, single pass against two passes, bit for bit."""
    from convexadam_amd import phantom as ph
    shape = (160, 192, 224)
    for pair, lo, hi in ((ph.deformed_pair(shape, 0, 4.0), 0, 2), (ph.zero_background_pair(shape, 0, 4.0), 100, 3000)):
        for img in pair:
            x = img.numpy()
            single(1)
            o1, o2, nrep = run(U, x, 6, 2)
            single(0)
            t1, t2, _ = run(U, x, 6, 2)
            assert same(o1, t1) and same(o2, t2)
            assert lo <= nrep <= hi, nrep


def test_pipeline_fields_identical_with_and_without_the_single_pass(single):
    """The whole-pair pipeline (records written by the marching kernel, float32 and half precision) against the two-pass path."""
    from convexadam_amd.convex_adam_MIND import convex_adam_pt
    from convexadam_amd.phantom import zero_background_pair
    fix, mov = zero_background_pair((48, 54, 84), 0, 3.0)
    kw = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=6, disp_hw=3, selected_niter=5, grid_sp_adam=2, ic=True)
    for half in (False, True):
        outs = []
        for v in (1, 0, 2):
            single(v)
            outs.append(convex_adam_pt(fix, mov, dtype=torch.float16 if half else torch.float32, device=torch.device(DEV), **kw))
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
