"""Live cross-check of the CPU oracle against the upstream reference itself, on inputs that are NOT in the golden files.
Runs only where the reference tree is mounted (the build container: /root/reference); skipped everywhere else, and never
part of the `-m gpu` selection -- nothing that runs on the GPU box reads the reference."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
try:                                                     # the import harness itself stays in the build container (.gpurunignore)
    from _ref_import import import_reference, reference_available  # noqa: E402
except ImportError:
    import_reference, reference_available = None, (lambda: False)

pytestmark = pytest.mark.skipif(not reference_available(), reason="upstream reference not mounted")


@pytest.fixture(scope="module")
def ref():
    return import_reference()


def test_operators_on_fresh_inputs(ref, orc):
    utils, _ = ref
    g = torch.Generator().manual_seed(4242)
    # correlate -> coupled_convex -> inverse_consistency on random features (hw 3, ragged extents)
    shape, hw, C = (7, 9, 10), 3, 12
    f = torch.rand(1, C, *shape, generator=g)
    m = torch.rand(1, C, *shape, generator=g)
    ssd, am = utils.correlate(f, m, hw, 1, shape, C)
    os_, oa = orc.correlate(f[0].numpy(), m[0].numpy(), hw)
    assert np.array_equal(ssd.numpy(), os_) and np.array_equal(am.numpy(), oa)
    n = 2 * hw + 1
    mesh_t = torch.nn.functional.affine_grid(hw * torch.eye(3, 4).unsqueeze(0), (1, 1, n, n, n), align_corners=True).permute(0, 4, 1, 2, 3).reshape(3, -1, 1)
    soft = utils.coupled_convex(ssd, am, mesh_t, 1, shape)
    assert np.array_equal(soft[0].numpy(), orc.coupled_convex(os_, oa, orc.disp_mesh(hw), hw))
    a = 0.3 * torch.randn(1, 3, *shape, generator=g)
    b = 0.3 * torch.randn(1, 3, *shape, generator=g)
    r1, r2 = utils.inverse_consistency(a, b, iter=7)
    o1, o2 = orc.inverse_consistency(a[0].numpy(), b[0].numpy(), 7)
    assert np.array_equal(r1[0].numpy(), o1) and np.array_equal(r2[0].numpy(), o2)


def test_mindssc_on_fresh_input(ref, orc):
    utils, _ = ref
    img = torch.randn(1, 1, 14, 15, 33, generator=torch.Generator().manual_seed(77))
    out = utils.MINDSSC(img, 1, 2, device="cpu")[0].numpy()
    mine = orc.mindssc(img[0, 0].numpy(), 1, 2)
    assert np.abs(out - mine).max() <= 6e-8          # MKL vsExp vs the expf restatement: <= 1 ulp


def test_whole_pipeline_convex_stage_on_fresh_pair(ref, orc):
    _, mind = ref
    from convexadam_amd.phantom import phantom
    fix = phantom((36, 32, 40), 31, 41)
    mov = torch.roll(phantom((36, 32, 40), 31, 42), (2, -2, 1), (0, 1, 2))
    kw = dict(mind_r=1, mind_d=2, lambda_weight=0, grid_sp=4, disp_hw=3, ic=True)
    out = mind.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device("cpu"), **kw)
    mine = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw)
    epe = float(np.sqrt(((out - mine) ** 2).sum(-1)).mean())
    assert epe <= 1e-5, epe                          # exp differs by <= 1 ulp; everything downstream is restated exactly


@pytest.mark.parametrize("threads", [3, 8])
def test_whole_pipeline_bit_identical_in_reference_bits_mode_on_fresh_pairs(ref, orc, mkl, threads):
    """LIVE: the reference runs here on pairs that are in no golden file (a textured pair with a zero background -- flat regions whose
    variance is clamped, so the thread-count dependent mean matters -- 25 Adam iterations, final smoothing) and the oracle, given THIS
    host's MKL tables (tests/mkl_tables.py, built from torch.exp / torch.sqrt) and the run's thread count, returns the same bits."""
    _, mind = ref
    from convexadam_amd.phantom import ellipsoid_mask, phantom
    t = mkl.host_tables(orc)
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    orc.set_exp_table(t["exp"], t["exp_first"], t["exp_count"])
    orc.set_sqrt_table(t["sqrt"])
    orc.set_mean_threads(threads)
    try:
        for shape, seed, kw in (((44, 40, 48), 5, dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=4, disp_hw=3, selected_niter=25, selected_smooth=3, grid_sp_adam=2, ic=True)),
                                ((40, 48, 36), 9, dict(mind_r=2, mind_d=1, lambda_weight=0.5, grid_sp=3, disp_hw=2, selected_niter=12, selected_smooth=0, grid_sp_adam=1, ic=False))):
            m = ellipsoid_mask(shape, 0.35)
            fix = phantom(shape, seed, 41) * m
            mov = torch.roll(phantom(shape, seed, 42), (2, -1, 1), (0, 1, 2)) * m
            out = mind.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device("cpu"), **kw)
            mine = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw)
            assert np.array_equal(out, mine), (shape, float(np.abs(out - mine).max()))
    finally:
        torch.set_num_threads(old)
        orc.set_exp_table(None)
        orc.set_sqrt_table(None)
        orc.set_mean_threads(0)


def test_nan_and_inf_in_the_cost_volume(ref, orc):
    """Nulls: torch.argmin returns the FIRST NaN of a column (a NaN counts as smaller than everything) and otherwise the first minimum;
    the oracle follows that in the plain argmin and in the six coupled passes (round 3; before, NaN entries were skipped)."""
    utils, _ = ref
    rng = np.random.default_rng(0)
    shape, hw = (5, 6, 7), 2
    mesh = orc.disp_mesh(hw)
    for case in range(4):
        f = rng.random((12,) + shape, dtype=np.float32)
        m = rng.random((12,) + shape, dtype=np.float32)
        if case == 0:
            m[3, 2, 3, 4] = np.nan                                        # some displacements of the neighbouring voxels see the NaN
        elif case == 1:
            m[0, 0, 0, 0] = np.nan
            f[5, 4, 5, 6] = np.inf
        elif case == 2:
            m[7, 1, 1, 1] = np.inf
            m[2, 3, 3, 3] = -np.inf
        else:
            f[:, 2, 2, 2] = np.nan                                        # every displacement of the neighbourhood is NaN
        ssd, am = utils.correlate(torch.from_numpy(f)[None], torch.from_numpy(m)[None], hw, 1, shape, 12)
        rs, ra = orc.correlate(f, m, hw)
        assert np.array_equal(ssd.numpy(), rs, equal_nan=True) and np.array_equal(am.numpy(), ra), case
        cs = utils.coupled_convex(ssd, am, torch.from_numpy(mesh)[:, :, None], 1, shape)
        assert np.array_equal(cs.numpy()[0], orc.coupled_convex(rs, ra, mesh, hw), equal_nan=True), case
