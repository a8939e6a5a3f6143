"""Pins the CPU oracle (oracle/cvx_oracle.c) to golden vectors captured from the upstream reference
(tests/golden/make_golden.py, reference imported in the build container, torch 2.10 CPU float32).

Bit-exact (`array_equal`) everywhere except the two places where the reference calls Intel MKL VML,
whose rounding is not restated as an algorithm:
  * exp() in MINDSSC (convex_adam_utils.py:63)           -> <= 1 ulp
  * sqrt() inside torch.optim.Adam (convex_adam_MIND.py:179) -> <= 1 ulp on ~0.5 % of elements; the
    parameter trajectory then diverges chaotically (hard part 1), so multi-iteration
    Adam results are compared with tolerances that are stated next to each assert.
Both deviations are pure functions of the argument and are TABULATED (tests/mkl_tables.py); with the two tables of the host that
produced the goldens the oracle is bit-identical to the reference everywhere, including the full-size benchmark pair after 80
Adam iterations (the `*_with_mkl_tables` tests at the end).
"""
import numpy as np
import pytest


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def epe(a, b):
    """mean endpoint error between two (..., 3) or (3, ...) fields given as (H,W,D,3)."""
    return float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(-1)).mean())


@pytest.mark.parametrize("key,r,d", [("mind_r1d2", 1, 2), ("mind_r2d2", 2, 2), ("mind_r1d1", 1, 1)])
def test_mindssc(orc, golden, key, r, d):
    g = golden("mind")
    out = orc.mindssc(g["img"], r, d)
    assert out.shape == g[key].shape
    assert ulp_diff(out, g[key]).max() <= 1          # MKL vsExp vs restated Sleef-style expf
    assert np.abs(out - g[key]).max() <= 6e-8


def test_pool_and_box(orc, golden):
    g = golden("pool")
    for k, gs in (("g2", 2), ("g3", 3), ("g6", 6)):
        assert np.array_equal(orc.avgpool_stride(g["x"], gs), g[k])
    assert np.array_equal(orc.box_zero(g["x"], 3), g["box3"])
    assert np.array_equal(orc.box_zero(g["x"], 5), g["box5"])


def test_mesh_matches_reference(orc, golden):
    g = golden("convex")
    assert np.array_equal(orc.disp_mesh(int(g["shape"][4])), g["mesh"])
    # hw = 6 is NOT an integer mesh in the reference (affine_grid rounding): -1.9999999 instead of -2
    m6 = orc.disp_mesh(6)
    assert m6[0, 4] == np.float32(-1.9999998807907104)


def test_correlate_bit_exact(orc, golden):
    g = golden("convex")
    hw = int(g["shape"][4])
    ssd, am = orc.correlate(g["feat_fix"], g["feat_mov"], hw)
    assert np.array_equal(ssd, g["ssd"])
    assert np.array_equal(am, g["argmin"])
    ssd_r, am_r = orc.correlate(g["feat_mov"], g["feat_fix"], hw)
    assert np.array_equal(am_r, g["argmin_rev"])
    assert float(ssd_r.astype(np.float64).sum()) == float(g["ssd_rev_sum"])


def test_correlate_many_channels_and_ragged_tail(orc, golden):
    """C = 20 exercises ATen's cascade sum; 7*9*5*9 columns (mod 32 != 0) its interleaved tail order."""
    g = golden("correlate_c20")
    ssd, am = orc.correlate(g["fix"], g["mov"], 1)
    assert np.array_equal(ssd, g["ssd"])
    assert np.array_equal(am, g["argmin"])


def test_coupled_convex_bit_exact(orc, golden):
    g = golden("convex")
    hw = int(g["shape"][4])
    soft = orc.coupled_convex(g["ssd"], g["argmin"], g["mesh"], hw)
    assert np.array_equal(soft, g["soft"])


def test_inverse_consistency_bit_exact(orc, golden):
    g = golden("convex")
    o1, o2 = orc.inverse_consistency(g["ic_in1"], g["ic_in2"], 15)
    assert np.array_equal(o1, g["ic_out1"])
    assert np.array_equal(o2, g["ic_out2"])


def test_resize_bit_exact(orc, golden):
    g = golden("convex")
    H, W, D, gs, _ = [int(v) for v in g["shape"]]
    h, w, d = H // gs, W // gs, D // gs
    scale = (np.array([h - 1, w - 1, d - 1], np.float32) / np.float32(2)).reshape(3, 1, 1, 1)
    up_in = (g["ic_out1"][::-1] * scale) * np.float32(gs)
    hr = orc.resize_trilinear(up_in, (H, W, D))
    assert np.array_equal(hr, g["disp_hr"])
    assert np.array_equal(orc.resize_trilinear(g["disp_hr"], (H // 2, W // 2, D // 2)), g["disp_lr"])


def test_adam_first_iteration_gradient_bit_exact(orc, golden):
    """U (smoothed grid) and dL/dP of iteration 1 are bit-identical to autograd's; the parameter
    after the step differs only through MKL's sqrt (<= 1 ulp of the update, a few 1e-7)."""
    g = golden("adam")
    r = orc.adam_run(g["F2"], g["M2"], g["P0"], float(g["lam"]), 1, want_grad=True)
    assert np.array_equal(r["U"], g["U_1"])
    assert np.array_equal(r["G"], g["G_1"])
    assert np.abs(r["P"] - g["P_1"]).max() <= 2.5e-7
    assert (r["P"] != g["P_1"]).mean() < 0.02


@pytest.mark.parametrize("niter,tol_u,tol_p", [(2, 1e-6, 2e-6), (5, 5e-6, 2e-5), (20, 5e-5, 5e-4)])
def test_adam_short_horizons(orc, golden, niter, tol_u, tol_p):
    g = golden("adam")
    r = orc.adam_run(g["F2"], g["M2"], g["P0"], float(g["lam"]), niter)
    assert np.abs(r["U"] - g["U_%d" % niter]).max() <= tol_u
    assert np.abs(r["P"] - g["P_%d" % niter]).max() <= tol_p


def test_pipeline_convex_only(orc, golden):
    """Whole convex stage (MIND -> pool -> correlate -> coupled convex -> IC -> resize) against
    convex_adam_pt(lambda_weight=0).  MIND differs by <= 1 ulp (MKL exp), every later operator is
    bit-exact, so the fields agree unless an argmin flips; tolerance 1e-5 voxel mean EPE."""
    g = golden("pipeline")
    kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2)
    out = orc.convex_adam_pipeline(g["fix"], g["mov"], lambda_weight=0, ic=True, **kw)
    assert out.shape == g["convex_only_ic"].shape and out.dtype == np.float64
    assert epe(out, g["convex_only_ic"]) < 1e-5
    out = orc.convex_adam_pipeline(g["fix"], g["mov"], lambda_weight=0, ic=False, **kw)
    # reference quirk (convex_adam_MIND.py:143-144): without ic the coarse field is returned as is
    assert out.shape == g["convex_only_noic"].shape
    assert epe(out, g["convex_only_noic"]) < 1e-5


@pytest.mark.parametrize("key,niter,smooth,ic,tol", [("adam_1", 1, 0, True, 1e-5), ("adam_5", 5, 0, True, 1e-4),
                                                      ("adam_20", 20, 0, True, 1e-3), ("adam_5_smooth3", 5, 3, True, 1e-4),
                                                      ("adam_5_noic", 5, 0, False, 1e-4)])
def test_pipeline_with_adam(orc, golden, key, niter, smooth, ic, tol):
    """North-star tolerance: < 1e-3 voxel mean EPE (BASELINE.json); tighter for short horizons."""
    g = golden("pipeline")
    out = orc.convex_adam_pipeline(g["fix"], g["mov"], mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2,
                                   lambda_weight=1.25, selected_niter=niter, selected_smooth=smooth, ic=ic)
    assert out.shape == g[key].shape
    assert epe(out, g[key]) < tol


def test_translation_known_answers(orc, golden):
    """64^3 translated phantom, convex only; centre-crop mean displacement."""
    from convexadam_amd.phantom import phantom
    import torch

    g = golden("translation64")
    fix = phantom((64, 64, 64), 2, 20)
    for name, sh, gs in (("roll_4_0_m8_gs4", (4, 0, -8), 4), ("roll_6_m6_0_gs6", (6, -6, 0), 6)):
        mov = torch.roll(fix, sh, (0, 1, 2))
        out = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), lambda_weight=0, grid_sp=gs, disp_hw=4)
        assert np.allclose(out[16:48, 16:48, 16:48].mean((0, 1, 2)), g[name], atol=1e-4)
        assert np.abs(out[::4, ::4, ::4] - g[name + "_sub"]).max() < 1e-3
        # the field recovers the shift (fixed(x) ~ moving(x + u)), within a coarse-grid fraction
        assert np.allclose(out[16:48, 16:48, 16:48].mean((0, 1, 2)), sh, atol=0.35)


def test_label_features(orc, golden):
    g = golden("labels")
    ff, fm, present = orc.label_features(g["lab_fix"].astype(np.float32), g["lab_mov"].astype(np.float32), 10.0)
    assert ff.shape[0] == g["weights"].shape[0]
    w = ff.reshape(ff.shape[0], -1).max(1)
    # round 3: torch.pow (Sleef powf for the leading blocks of 32 elements, scalar double pow for the tail) and weight.mean() are restated
    assert np.array_equal(w, g["weights"])
    assert np.allclose(ff.astype(np.float64).sum((1, 2, 3)), g["feat_fix_sum"], rtol=1e-14, atol=0)
    assert np.allclose(fm.astype(np.float64).sum((1, 2, 3)), g["feat_mov_sum"], rtol=1e-14, atol=0)
    assert np.array_equal(orc.avgpool_stride(ff, 2), g["feat_fix_pool2"])


def test_nnunet_pipeline_vs_reference_golden(orc, golden, nnunet):
    """BASELINE configs[3] end to end (convex_adam_nnUNet.py:41-159 run by tests/golden/make_golden_nnunet.py, 18 channels: ATen's
    cascade channel sum): label features -> correlation, coupled convex, inverse consistency -> Adam.  FROM THE LABEL MAPS the oracle
    reproduces the reference's features (torch.pow and the mean of the weights are restated), the convex stage bit for bit, the Adam
    horizons within the sqrt-ulp sensitivity, and -- with the golden host's sqrt table -- every horizon bit for bit (no exp and no
    global mean on this path)."""
    g = golden("nnunet")
    gs, hw, gsa = (int(v) for v in g["cfg"])
    lf, lm, ff, fm = nnunet.features(g)
    of, om, _ = orc.label_features(lf, lm, 10.0)
    assert np.array_equal(of, ff) and np.array_equal(om, fm)              # the oracle's own features ARE the reference's
    kw = dict(grid_sp=gs, disp_hw=hw, grid_sp_adam=gsa, ic=True, features=(of, om))
    conv = orc.convex_adam_pipeline(None, None, lambda_weight=0, **kw)
    nnunet.field_checks(g, "convex", conv, exact=True)
    assert np.abs(conv).mean() > 0.3                                   # a real displacement, not the identity
    for niter, tol in ((1, 1e-6), (5, 1e-5), (20, 1e-3)):
        out = orc.convex_adam_pipeline(None, None, lambda_weight=1.25, selected_niter=niter, **kw)
        assert nnunet.field_checks(g, "adam_%d" % niter, out, exact=False) < tol, niter
    q = golden("mkl_vssqrt_low")
    orc.set_sqrt_table(q["normal"], q["denormal"])
    try:
        for niter in (1, 5, 20):
            out = orc.convex_adam_pipeline(None, None, lambda_weight=1.25, selected_niter=niter, **kw)
            nnunet.field_checks(g, "adam_%d" % niter, out, exact=True)
    finally:
        orc.set_sqrt_table(None)


def test_masked_feature_extraction(orc, golden):
    """extract_features(use_mask=True), convex_adam_MIND.py:36-54: eroded mask, half-resolution nearest-in-mask fill
    (scipy EDT on the host, as in the reference), x2 trilinear up-sampling, MIND-SSC of the filled image."""
    g = golden("masked")
    for img, mask, key in ((g["img_fix"], g["mask_fix"], "feat_fix"), (g["img_mov"], g["mask_mov"], "feat_mov")):
        filled, m = orc.replicate_fill(img, mask)
        assert 0.05 < m.mean() < 0.5
        assert np.array_equal(filled[m != 0], img[m != 0])
        out = orc.mindssc(filled, 1, 2)
        assert np.abs(out - g[key]).max() <= 6e-8           # 1 ulp: MKL exp only


def _smoother_specs(orc, g):
    return {"gauss07": orc.make_smoother(gauss_w=g["gauss07_w"]), "gauss10": orc.make_smoother(gauss_w=g["gauss10_w"]),
            "kov16": orc.make_smoother([3, 3, 3, 3]), "kov19": orc.make_smoother([3, 3, 3, 5]), "kov28": orc.make_smoother([5, 5, 5, 5])}


def test_sweep_smoothers_forward_and_adjoint_bit_exact(orc, golden):
    """GaussianSmoothing / kovesi_spline of self_configuring/convexAdam_hyper_util.py:454-488 and
    their autograd adjoints (oneDNN convolution / avg_pool3d_backward evaluation order)."""
    g = golden("smoothers")
    for k, sm in _smoother_specs(orc, g).items():
        assert np.array_equal(orc.smooth(g["x"], sm), g[k + "_fwd"]), k
        assert np.array_equal(orc.smooth(g["go"], sm, backward=True), g[k + "_bwd"]), k


def test_adam_with_sweep_smoothers(orc, golden):
    """adam_run_withconfig_shiftSpline.py:214-230 with avgs[avg_n] instead of the three 3^3 boxes."""
    g, a = golden("smoothers"), golden("adam")
    specs = _smoother_specs(orc, g)
    for k in ("gauss07", "kov19"):
        r = orc.adam_run(a["F2"], a["M2"], a["P0"], 0.8, 1, want_grad=True, smoother=specs[k])
        assert np.array_equal(r["U"], g[k + "_adam_U1"]) and np.array_equal(r["G"], g[k + "_adam_G1"])
        r3 = orc.adam_run(a["F2"], a["M2"], a["P0"], 0.8, 3, smoother=specs[k])
        assert np.abs(r3["U"] - g[k + "_adam_U3"]).max() < 1e-6          # MKL sqrt in the optimiser step


# ---- evaluation operators of the sweep / apply_convex: metrics_oracle vs reference goldens ----
def test_metrics_oracle_vs_reference_golden(golden):
    from oracle import metrics_oracle as mo
    g = golden("metrics")
    assert np.array_equal(mo.jacobian_determinant_3d(g["disp"], False), g["jac_vox"])
    assert np.array_equal(mo.jacobian_determinant_3d(g["disp_norm"], True), g["jac_norm"])
    std, neg = mo.jacobian_stats(g["jac_vox"])
    assert abs(std - float(g["jac_log_std"])) <= 1e-5 * float(g["jac_log_std"]) and abs(neg - float(g["jac_neg_frac"])) <= 1e-6
    warped = mo.warp_labels_nearest(g["seg_moving"], g["disp"])
    assert np.array_equal(warped, g["seg_warped"])
    assert np.array_equal(mo.dice_coeff(g["seg_fixed"], warped, 7), g["dice"])
    samp = mo.sample_field_at_points(g["disp"], g["key_fixed"])
    assert np.array_equal(samp, g["disp_sampled"])
    # the reference's sqrt is MKL's (<= 1 ulp, same non-restatable site as in torch.optim.Adam)
    assert np.allclose(mo.tre(g["key_fixed"], g["key_moving"], samp), g["tre"], rtol=2e-7, atol=0)
    assert np.array_equal(mo.sort_rank(g["rank_in"]), g["rank_out"])


def test_apply_convex_oracle_vs_reference_golden_and_scipy(golden):
    from scipy.ndimage import map_coordinates
    from oracle import metrics_oracle as mo
    g = golden("metrics")
    dd = g["disp"].transpose(1, 2, 3, 0).astype(np.float64)
    assert np.array_equal(mo.apply_convex(dd, g["moving"]), g["warped"])
    rng = np.random.default_rng(5)
    mov = rng.random((9, 8, 11))
    d2 = rng.standard_normal((9, 8, 11, 3)) * 4.0
    idn = np.meshgrid(np.arange(9), np.arange(8), np.arange(11), indexing="ij")
    assert np.array_equal(mo.apply_convex(d2, mov), map_coordinates(mov, d2.transpose(3, 0, 1, 2) + idn, order=1))


def test_torch_linspace_restatement():
    import torch
    from oracle import metrics_oracle as mo
    for n in (2, 3, 17, 64, 255):
        for a, b in ((1.0, 0.1), (-3.5, 2.25)):
            assert np.array_equal(mo.linspace(a, b, n), torch.linspace(a, b, n).numpy())


def test_hd95_oracle_vs_reference_golden(golden):
    """cupy_hd95 captured from the reference module (cupy/cupyx supplied by numpy/scipy, see make_golden.py --hd95)."""
    from oracle import metrics_oracle as mo
    g = golden("hd95")
    assert np.array_equal(mo.hd95(g["seg_fixed"], g["seg_moving"], 6), g["hd95_p1"])
    assert np.array_equal(mo.hd95(g["seg_fixed"], g["seg_moving"], 6, 2), g["hd95_p2"])
    assert g["hd95_p1"][3] == 30 and g["hd95_p1"][5] == 30 and g["hd95_p2"][3] == 15
    # non-integer scale factors (F.interpolate's nearest mode: extent (int)(n * s), index min(floor(dst * float32(1 / s)), n - 1))
    for key, prec in (("hd95_p1_5", 1.5), ("hd95_p0_5", 0.5), ("hd95_p2_5", 2.5)):
        assert np.array_equal(mo.hd95(g["seg_fixed"], g["seg_moving"], 6, prec), g[key]), key
    assert g["hd95_p1_5"][3] == 20 and g["hd95_p0_5"][3] == 60


def test_percentile_restatement_vs_numpy():
    """The product reads the percentile from a histogram: two order statistics + numpy's interpolation rule (float32)."""
    from convexadam_amd.convexAdam_hyper_util import percentile_linear_from_sorted_pair, percentile_neighbours
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 19, 20, 21, 40, 41, 1000, 99991, 1234567):
        x = np.sort(np.sqrt(rng.integers(0, 500, n).astype(np.float64)).astype(np.float32))
        for q in (95, 50, 0, 100, 30):
            k0, k1, gamma = percentile_neighbours(n, q)
            got, ref = percentile_linear_from_sorted_pair(x[k0], x[k1], gamma), np.percentile(x, q)
            assert got == ref and got.dtype == ref.dtype, (n, q)


def test_feature_transform_restatement_vs_scipy(orc):
    """The masked path's nearest-in-mask search: scipy.ndimage.distance_transform_edt(return_indices=True) (a third-party
    dependency of the reference) restated with its tie-breaking; compared with scipy itself on random and tie-heavy masks."""
    from scipy.ndimage import distance_transform_edt as edt
    rng = np.random.default_rng(0)
    for trial in range(120):
        shape = tuple(int(v) for v in rng.integers(2, 15, 3))
        m = rng.random(shape) < rng.choice([0.02, 0.1, 0.3, 0.6, 0.9])
        if m.all():
            m.flat[int(rng.integers(m.size))] = False
        assert np.array_equal(orc.feature_transform(m), edt(m, return_indices=True)[1]), (trial, shape)
    for shape in ((9, 9, 9), (8, 10, 12), (16, 16, 16), (40, 48, 56)):
        m = np.ones(shape, bool)
        m[::4, ::4, ::4] = False                                      # lattice of sites: every cell centre is a tie
        assert np.array_equal(orc.feature_transform(m), edt(m, return_indices=True)[1])
        m = np.ones(shape, bool)
        m[0, 0, 0] = m[-1, -1, -1] = m[0, -1, 0] = False
        assert np.array_equal(orc.feature_transform(m), edt(m, return_indices=True)[1])


@pytest.mark.timeout(600)
def test_full_size_benchmark_pair_vs_reference_golden(orc, golden):
    """BASELINE configs[1] at FULL size (160x192x224, hw 6, gs 6, ic, 80 Adam iterations), reference captured by
    tests/golden/make_golden_fullsize.py: the whole convex stage (MIND, both correlations, coupled convex, inverse consistency)
    is bit-identical to the reference; after 80 Adam iterations the oracle is closer to the reference than the reference is to
    a copy of itself whose warped features were perturbed by one ulp (the loop amplifies MKL's 1-ulp exp / sqrt sites)."""
    from convexadam_amd.phantom import deformed_pair
    g = golden("fullsize")
    shape = (160, 192, 224)
    fix, mov = deformed_pair(shape, 0, 4.0)
    kw = dict(mind_r=1, mind_d=2, grid_sp=6, disp_hw=6, grid_sp_adam=2, ic=True)
    conv = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), lambda_weight=0, **kw)
    assert np.array_equal(np.moveaxis(conv, -1, 0).astype(np.float32), orc.resize_trilinear(g["c1_coarse_ic"], shape))
    out = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), lambda_weight=1.25, selected_niter=80, **kw)
    s = int(g["sub"])
    e = epe(out[::s, ::s, ::s], np.moveaxis(g["c1_adam_80_sub"], 0, -1))
    self_e = float(g["c1_self_perturbation_epe_sub"][list(g["c1_snaps"]).index(80)])
    print("80 iterations, full size: oracle vs reference mean EPE %.3e; reference vs its 1-ulp-perturbed self %.3e" % (e, self_e))
    assert e <= self_e
    assert e < 2e-3


def test_correlate_variants_vs_reference_scripts(orc, golden):
    """cost="sad" / n_box=1: bit-identical to the `correlate` functions of the challenge scripts, which
    tests/golden/make_golden_variants.py lifts out of l2r_2021_convexAdam_task2/3_docker.py and executes."""
    g = golden("variants")
    for tag, cost in (("sad1", "sad"), ("ssd1", "ssd"), ("sad1_w", "sad")):
        ssd, am = orc.correlate(g[tag + "_fix"], g[tag + "_mov"], int(g[tag + "_hw"]), cost=cost, n_box=1)
        step = 7 if tag.endswith("_w") else 1
        assert np.array_equal(ssd[::step], g[tag + "_ssd"]) and np.array_equal(am, g[tag + "_argmin"]), tag
        assert float(ssd.astype(np.float64).sum()) == float(g[tag + "_ssd_sum"])


def test_adam_loop_bit_identical_with_the_mkl_sqrt_table(orc, golden):
    """The only non-restated site of the Adam loop is the square root of the reference BUILD (MKL vsSqrt).  Its deviation from the IEEE
    root is tabulated (tests/golden/mkl_vssqrt_low.npz: exhaustive over all float32 inputs); with that table the oracle reproduces
    the reference's control grid, its gradient and disp_sample BIT FOR BIT after 1, 2, 5 and 20 iterations."""
    g, t = golden("adam"), golden("mkl_vssqrt_low")
    assert [int(v) for v in t["counts"]] == [59788, 39167, 52462]
    orc.set_sqrt_table(t["normal"], t["denormal"])
    try:
        for niter in (1, 2, 5, 20):
            r = orc.adam_run(g["F2"], g["M2"], g["P0"], float(g["lam"]), niter, want_grad=True)
            assert np.array_equal(r["U"], g["U_%d" % niter]) and np.array_equal(r["G"], g["G_%d" % niter]) and np.array_equal(r["P"], g["P_%d" % niter]), niter
    finally:
        orc.set_sqrt_table(None)
    r = orc.adam_run(g["F2"], g["M2"], g["P0"], float(g["lam"]), 20)
    assert not np.array_equal(r["P"], g["P_20"])                     # the IEEE root differs, which is the default everywhere


# ---- the two MKL sites as tables: bit-identical to the reference end to end -----------------------------------------------------------
def test_mindssc_bit_identical_with_the_mkl_exp_table(orc_reference_bits, golden):
    g = golden("mind")
    for key, r, d in (("mind_r1d2", 1, 2), ("mind_r2d2", 2, 2), ("mind_r1d1", 1, 1)):
        assert np.array_equal(orc_reference_bits.mindssc(g["img"], r, d), g[key]), key


def test_every_pipeline_golden_bit_identical_with_mkl_tables(orc_reference_bits, golden):
    """The goldens that the default oracle meets within a tolerance (whole pipelines, masked features) are met EXACTLY with the tables."""
    orc, g = orc_reference_bits, golden("pipeline")
    kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2)
    assert np.array_equal(orc.convex_adam_pipeline(g["fix"], g["mov"], lambda_weight=0, ic=True, **kw), g["convex_only_ic"])
    assert np.array_equal(orc.convex_adam_pipeline(g["fix"], g["mov"], lambda_weight=0, ic=False, **kw), g["convex_only_noic"])
    for key, niter, smooth, ic in (("adam_1", 1, 0, True), ("adam_5", 5, 0, True), ("adam_20", 20, 0, True), ("adam_5_smooth3", 5, 3, True),
                                   ("adam_5_noic", 5, 0, False)):
        out = orc.convex_adam_pipeline(g["fix"], g["mov"], lambda_weight=1.25, selected_niter=niter, selected_smooth=smooth, ic=ic, **kw)
        assert np.array_equal(out, g[key]), key
    m = golden("masked")
    for img, mask, key in ((m["img_fix"], m["mask_fix"], "feat_fix"), (m["img_mov"], m["mask_mov"], "feat_mov")):
        filled, _ = orc.replicate_fill(img, mask)
        assert np.array_equal(orc.mindssc(filled, 1, 2), m[key]), key


def test_full_size_bit_identical_to_reference_with_mkl_tables(orc_reference_bits, golden):
    """BASELINE configs[1] exactly as bench.py times it, 160x192x224, 80 Adam iterations: with the exp and sqrt tables of the golden host
    the oracle's field EQUALS the field captured from the reference (every 4th voxel per axis bit for bit, float64 sums of the whole
    field and of its squares to 1e-14) -- nothing in the pipeline is approximated, the default build differs from the reference
    only through the two <= 1 ulp library calls."""
    from convexadam_amd.phantom import deformed_pair
    orc, g = orc_reference_bits, golden("fullsize")
    shape = (160, 192, 224)
    fix, mov = deformed_pair(shape, 0, 4.0)
    out = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), mind_r=1, mind_d=2, grid_sp=6, disp_hw=6, grid_sp_adam=2, ic=True,
                                   lambda_weight=1.25, selected_niter=80)
    f = np.moveaxis(out, -1, 0).astype(np.float32)
    s = int(g["sub"])
    assert np.array_equal(f[:, ::s, ::s, ::s], g["c1_adam_80_sub"])
    # whole-field float64 sums, evaluated with torch like the capture did (numpy's multi-axis reduction is not pairwise)
    import torch
    ft = torch.from_numpy(np.ascontiguousarray(f)).double()
    assert np.allclose(ft.sum((1, 2, 3)).numpy(), g["c1_adam_80_sum"], rtol=1e-14, atol=0)
    assert np.allclose(ft.square().sum((1, 2, 3)).numpy(), g["c1_adam_80_sumsq"], rtol=1e-14, atol=0)


def test_full_size_masked_config3_bit_identical_to_reference(orc_reference_bits, golden):
    """BASELINE configs[2] (224x192x224, ellipsoid masks, disp_hw 8, 20 Adam iterations): equal to the reference capture.  The masked
    images have flat filled regions whose variance is clamped to mean / 1000, so this also needs the reference's `mind_var.mean()` to
    the last bit: torch's float sum with the capture's 8 threads, restated in orc_torch_sum."""
    from convexadam_amd.phantom import deformed_pair, ellipsoid_mask
    import torch
    orc, g = orc_reference_bits, golden("fullsize")
    s = int(g["sub"])
    shape = (224, 192, 224)
    fix, mov = deformed_pair(shape, 3, 10.0)
    mf, mm = ellipsoid_mask(shape, 0.35), ellipsoid_mask(shape, 0.35, shift=(4, -3, 5))
    filled_f, _ = orc.replicate_fill(fix.numpy(), mf.numpy())
    filled_m, _ = orc.replicate_fill(mov.numpy(), mm.numpy())
    feats = (orc.mindssc(filled_f, 1, 2), orc.mindssc(filled_m, 1, 2))
    out = orc.convex_adam_pipeline(None, None, features=feats, lambda_weight=1.25, grid_sp=6, disp_hw=8, selected_niter=20, selected_smooth=0,
                                   grid_sp_adam=2, ic=True)
    f = np.moveaxis(out, -1, 0).astype(np.float32)
    assert np.array_equal(f[:, ::s, ::s, ::s], g["c3_adam_20_sub"])
    ft = torch.from_numpy(np.ascontiguousarray(f)).double()
    assert np.allclose(ft.sum((1, 2, 3)).numpy(), g["c3_adam_20_sum"], rtol=1e-14, atol=0)
    assert np.allclose(ft.square().sum((1, 2, 3)).numpy(), g["c3_adam_20_sumsq"], rtol=1e-14, atol=0)


def test_host_tables_match_the_fixtures_on_the_golden_host(orc, mkl):
    """The fixtures are nothing but torch.exp / torch.sqrt of the host that produced the goldens, tabulated: rebuilt from torch here
    they are identical.  Another CPU model makes MKL take another code path (the GPU boxes' EPYC hosts do): skipped there."""
    t = mkl.host_tables(orc)
    if not t["matches_golden_host"]:
        pytest.skip("this host's MKL code path differs from the one the goldens were generated on")
    gt = mkl.golden_tables()
    assert np.array_equal(t["exp"], gt["exp"]) and np.array_equal(t["sqrt"], gt["sqrt"])


def fullsize_case(tag):
    """Inputs of the full-size captures (tests/golden/fullsize.npz: c1; fullsize2.npz: c4-c6), regenerated from seeds: (kind, shape, a, b)."""
    from convexadam_amd import phantom as ph
    if tag == "c1":
        return ("images", (160, 192, 224)) + ph.deformed_pair((160, 192, 224), 0, 4.0)          # the benchmark pair
    if tag == "c4":
        return ("images", (160, 192, 224)) + ph.deformed_pair((160, 192, 224), 2, 6.0)          # another seed, 6-voxel warp
    if tag == "c5":
        return ("images", (160, 192, 224)) + ph.zero_background_pair((160, 192, 224), 0, 4.0)   # EXACT-zero background
    return ("labels", (160, 192, 160)) + ph.warped_label_pair((160, 192, 160), 18, 11, 0.05)     # 18 labels, nnUNet path (C >= 16)


FULLSIZE_TAGS = ("c1", "c4", "c5", "c6")
# The envelope the throughput mode is held to at 80 iterations (see the test below): no capture further than 1.6 x the reference's own
# 1-ulp self-perturbation distance, and not further than it on average over the captures.
FAST_80_ENVELOPE = 1.6


def fullsize_stages(orc, g, tag):
    """Oracle pipeline of one capture up to the start of the Adam loop; asserts the convex stage bit for bit against the reference."""
    kind, shape, a, b = fullsize_case(tag)
    kw = dict(mind_r=1, mind_d=2, grid_sp=6, disp_hw=6, grid_sp_adam=2, ic=True)
    if kind == "labels":
        ff, fm, _ = orc.label_features(a.numpy(), b.numpy(), 10.0)
        assert ff.shape[0] == int(g["c6_n_ch"]) >= 16
        _, st = orc.convex_adam_pipeline(None, None, lambda_weight=1.25, selected_niter=1, return_stages=True, features=(ff, fm), **kw)
    else:
        if tag == "c5":
            assert np.allclose([float((a == 0).float().mean()), float((b == 0).float().mean())], g["c5_zero_fraction"]) and float((a == 0).float().mean()) > 0.5
        _, st = orc.convex_adam_pipeline(a.numpy(), b.numpy(), lambda_weight=1.25, selected_niter=1, return_stages=True, **kw)
    scale = ((np.array(st["fs"].shape[1:], np.float32) - 1) / np.float32(2)).reshape(3, 1, 1, 1)
    assert np.array_equal((st["ice"][::-1] * scale) * np.float32(6), g[tag + "_coarse_ic"]), tag + ": convex stage differs from the reference"
    return shape, st


def horizons(orc, g, tag, shape, st, mode):
    """Mean EPE of one oracle Adam run against the reference capture at 1 / 20 / 40 / 80 iterations (one 80-iteration run, observed
    at every horizon through the optimiser state)."""
    s = int(g["sub"])
    res, state, done = {}, None, 0
    for n in [int(v) for v in g[tag + "_snaps"]]:
        r = orc.adam_run(st["F2"], st["M2"], st["P0"] if state is None else state["P"], 1.25, n - done, mode=mode,
                         m=None if state is None else state["m"], v=None if state is None else state["v"], step0=done)
        state, done = r, n
        f = orc.resize_trilinear(r["U"] * np.float32(2), shape)
        res[n] = epe(np.moveaxis(f[:, ::s, ::s, ::s], 0, -1), np.moveaxis(g["%s_adam_%d_sub" % (tag, n)], 0, -1))
    return res


@pytest.mark.timeout(1500)
def test_fast_adam_mode_against_four_reference_captures(orc, golden):
    """adam_mode="fast" (orc_adam_run_fast: the arithmetic the HIP kernels of convexadam_amd/csrc/adamfast.hip follow bit for bit) at FULL
    size against fields captured from the reference itself: the benchmark pair (c1) and the three captures VERDICT round 4 asked for
    (c4 another seed and a 6-voxel warp, c5 an EXACT-zero background, c6 18-label maps through convex_adam_nnUNet: C >= 16).

    What holds on every capture and is asserted: convex stage bit-identical; mean EPE 0 after one iteration; < 1e-3 after 20 and 40.
    At 80 iterations every arithmetic is 1-3e-3 voxel from the reference -- the reference itself after a 1-ulp perturbation of its
    warped features (1.6 / 1.8 / 2.1 / 2.7e-3 on c1 / c4 / c5 / c6), the exact-order restatement with this library's libm (1.2 / 2.2 /
    1.3 / 0.6e-3), the fast mode (1.4 / 2.7 / 1.2 / 1.3e-3).  The criteria registered in round 3 for ONE pair (<= the self-perturbation
    distance AND <= 1.15 x the exact mode's) do NOT survive the wider set: the fast mode misses the first on c4 (as does the exact mode)
    and the second on c1 / c4 / c6 -- which is why "exact" is the package default again (convex_adam_MIND.py) and the fast mode is
    opt-in.  Asserted here as a regression envelope, not as an acceptance: no capture beyond FAST_80_ENVELOPE x its self-perturbation
    distance, and on average over the captures not beyond it.  adam_mode="fast_all" (separable forward boxes too) is further away on
    every capture at 20 iterations and on average at 80."""
    g1, g2 = golden("fullsize"), golden("fullsize2")
    table = {}
    for tag in FULLSIZE_TAGS:
        g = g1 if tag == "c1" else g2
        shape, st = fullsize_stages(orc, g, tag)
        fast = horizons(orc, g, tag, shape, st, "fast")
        self_e = [float(v) for v in g[tag + "_self_perturbation_epe_sub"]]
        table[tag] = (fast, self_e)
        print(tag, "fast mode vs reference capture:", {n: "%.3e" % v for n, v in fast.items()}, "| reference vs its 1-ulp-perturbed self:", ["%.3e" % v for v in self_e], flush=True)
        assert fast[1] <= 1e-6 and fast[20] < 1e-3 and fast[40] < 1e-3, tag
        assert fast[80] <= FAST_80_ENVELOPE * self_e[3], tag
        if tag == "c1":                                   # (one capture is enough to keep the other modes' relation on record)
            fa = horizons(orc, g, tag, shape, st, "fast_all")
            print("   fast_all:", {n: "%.3e" % v for n, v in fa.items()})
            assert fa[20] > 3 * fast[20] and fa[80] > fast[80] and fa[80] < 3e-3
    assert np.mean([table[t][0][80] for t in FULLSIZE_TAGS]) <= np.mean([table[t][1][3] for t in FULLSIZE_TAGS])


def test_fast_box_is_the_three_chained_boxes(orc):
    """orc_fast_box3x3 (separable sums, one scale) equals box3(box3(box3(.))) with per-stage zero padding to rounding, borders included."""
    rng = np.random.default_rng(5)
    for shape in ((3, 4, 5), (2, 2, 2), (9, 7, 11), (1, 6, 3)):
        x = rng.standard_normal((3,) + shape).astype(np.float32)
        ref = orc.box_zero(orc.box_zero(orc.box_zero(x, 3), 3), 3)
        assert np.abs(orc.fast_box3x3(x) - ref).max() <= 4e-6 * max(1.0, float(np.abs(ref).max())), shape


def test_even_selected_smooth_grows_the_field_like_the_reference(orc, golden):
    """convex_adam_MIND.py:184-191 with an EVEN selected_smooth: the reference overwrites its own "+1" (:189), so each of the three
    avg_pool3d(k, stride 1, padding k//2) makes every axis one voxel longer and convex_adam_pt returns (H+3, W+3, D+3, 3).  The growing
    pool is bit-exact against torch's; the whole call against the reference's capture within the short-horizon tolerance (3 iterations)."""
    import torch
    from convexadam_amd.phantom import phantom

    g = golden("even_smooth")
    for k in (2, 4, 6):
        assert np.array_equal(orc.box_grow(g["pool_in"], k), g["pool%d" % k]), k
    shape = tuple(int(v) for v in g["shape"])
    fix = phantom(shape, 7, 70)
    mov = torch.roll(phantom(shape, 7, 71), (1, -1, 2), (0, 1, 2))
    for k in (2, 4):
        out = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=4, disp_hw=2, selected_niter=3,
                                       selected_smooth=k, grid_sp_adam=2, ic=True)
        assert out.shape == tuple(s + 3 for s in shape) + (3,) == g["k%d" % k].shape
        assert epe(out, g["k%d" % k]) < 1e-5, k
    # lambda_weight <= 0 never reaches the smoothing block (:155): the even kernel is ignored
    a = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), lambda_weight=0, grid_sp=4, disp_hw=2, selected_smooth=2)
    assert a.shape == shape + (3,)
