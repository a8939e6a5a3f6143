#!/usr/bin/env python
"""Long randomised parity campaign on the GPU box (not collected by pytest; the suite's own fuzz test is the short version):
    python tests/fuzz_campaign.py [--minutes 10] [--seed 1]
Random pairs through the whole pipeline (extents up to ~100 voxels, MIND radius / dilation, both grid spacings, search half-widths up
to 10, the challenge-script variants and fp16 storage, masked pairs, label-map pairs) and random Adam control grids, each compared
with the CPU oracle bit for bit.  Prints one line per failure with the configuration that reproduces it, and a summary."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from convexadam_amd import convex_adam_MIND as M           # noqa: E402
from convexadam_amd import convex_adam_utils as U          # noqa: E402
from convexadam_amd.phantom import ellipsoid_mask, phantom  # noqa: E402
from oracle import oracle as orc                           # noqa: E402  (the checker)

# trials that do not name a mode compare with the oracle's restatement of the REFERENCE's evaluation order (as tests/conftest.py does for
# the suite); the throughput arithmetic of the Adam loop is drawn explicitly (adam_mode="fast" on both sides)
M.set_default_adam_mode("exact")

DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def field(t):
    return np.moveaxis(host(t), 0, -1).astype(np.float64)


LARGE = False          # --large: extents of 96 .. 224 voxels (x tiles, uneven z chunks, y-tiled correlation, streaming fallbacks)


def _draw_options(rng):
    """Bit-identical kernel variants added in rounds 5 and 6, drawn per trial (process default context)."""
    from convexadam_amd import _lib
    L = _lib.lib()
    L.cvx_set_option(b"box_fwd_tile", int(rng.choice([-1, 0, 1000, 2000, 1834, 2274, 1222])))
    L.cvx_set_option(b"box_bwd_tile", int(rng.choice([-1, 0, 1000, 2000, 1834, 2274, 1222])))
    L.cvx_set_option(b"box_walk", int(rng.integers(0, 2)))
    L.cvx_set_option(b"corr_dual", int(rng.integers(0, 2)))
    L.cvx_set_option(b"prune_refine", int(rng.integers(0, 4) > 0))
    L.cvx_set_option(b"mind_records", int(rng.integers(0, 4) > 0))
    L.cvx_set_option(b"mind_blocked", int(rng.integers(0, 4) > 0))
    L.cvx_set_option(b"mind_single", int(rng.choice([0, 0, 1, 1, 2])))          # two passes / single pass + repair / every block through the repair kernel
    L.cvx_set_option(b"resize_up2", int(rng.integers(0, 4) > 0))
    # round 6: the certified-fast correlation path (1 = role kernel in fast arithmetic, 2 = staged kernel) or the exact volumes (0)
    L.cvx_set_option(b"corr_cert", int(rng.choice([1, 1, 2, 2, 0])))
    L.cvx_set_option(b"cf_map", int(rng.integers(0, 2)))
    L.cvx_set_option(b"cert_unfused", int(rng.choice([1, 2, 2, 0])))          # C >= 16: two-kernel certified-fast path by its default rule / always / never


def trial_pipeline(rng, t):
    gs, gsa, hw = int(rng.choice([2, 3, 4, 5, 6, 7])), int(rng.choice([1, 2, 3, 4])), int(rng.integers(1, 11))
    big = rng.random() < 0.3
    if LARGE:
        gs, gsa = int(rng.choice([3, 4, 5, 6, 8])), int(rng.choice([2, 2, 3, 4]))
        shape = tuple(int(rng.integers(96, 225)) for _ in range(3))
    else:
        shape = tuple(int(max(2 * gs, 2 * gsa, 8) + rng.integers(0, 90 if big else 36)) for _ in range(3))
    coarse = np.prod([s // gs for s in shape])
    while (2 * hw + 1) ** 3 * coarse > (4e8 if LARGE else 6e7):     # keep the oracle's cost volume below ~240 MB (1.6 GB with --large)
        hw -= 1
    kw = dict(mind_r=int(rng.choice([1, 2, 3])), mind_d=int(rng.choice([1, 2, 3, 4])), grid_sp=gs, disp_hw=hw, grid_sp_adam=gsa,
              lambda_weight=float(rng.choice([0.0, 0.7, 1.25])), selected_niter=int(rng.integers(1, 6)), ic=bool(rng.integers(0, 2)),
              selected_smooth=int(rng.choice([0, 0, 3, 5])))
    var = {}
    r = rng.random()
    if r < 0.4:
        var = [dict(cost="sad"), dict(n_box=1), dict(n_spline_pools=2), dict(cost="sad", n_box=1, n_spline_pools=2), dict(storage="fp16"),
               dict(storage="fp16", n_spline_pools=2)][int(rng.integers(0, 6))]
    elif r < 0.7:
        var = dict(adam_mode="fast")           # round 4: the throughput arithmetic of the Adam loop against its own oracle restatement
        if rng.random() < 0.3:
            var["storage"] = "fp16"            # round 5: 8-byte half-precision feature records in the throughput loop
    _draw_options(rng)                         # round 5: tile forward boxes, z-walking box filters, both directions in one correlation launch
    fix = phantom(shape, 1000 + t, 2000 + t)
    mov = torch.roll(phantom(shape, 1000 + t, 3000 + t), (1, -1, 2), (0, 1, 2))
    if rng.random() < 0.2:                     # exact-zero background around the body (flat MIND -> all-zero cost columns at the coarse grid)
        body = ellipsoid_mask(shape, float(rng.uniform(0.2, 0.4)))
        fix, mov = fix * body, mov * body
        var = dict(var, zero_background=True)
    run = {k: v for k, v in var.items() if k != "zero_background"}
    out = field(M.register_pair_device(fix.to(DEV), mov.to(DEV), **kw, **run))
    ref = orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw, **run)
    return np.array_equal(out, ref), ("pipeline", shape, kw, var)


def trial_masked(rng, t):
    gs, hw = int(rng.choice([3, 4, 6])), int(rng.integers(1, 6))
    shape = tuple(int(2 * (max(gs, 6) + rng.integers(0, 24))) for _ in range(3))
    kw = dict(mind_r=1, mind_d=int(rng.choice([1, 2])), grid_sp=gs, disp_hw=hw, grid_sp_adam=2, lambda_weight=1.25,
              selected_niter=int(rng.integers(1, 4)), ic=True, selected_smooth=0)
    fix = phantom(shape, 4000 + t, 5000 + t)
    mov = torch.roll(phantom(shape, 4000 + t, 6000 + t), (1, 0, -1), (0, 1, 2))
    mf = ellipsoid_mask(shape, float(rng.uniform(0.25, 0.45))).float()
    mm = torch.roll(mf, (1, 0, -1), (0, 1, 2))
    gf, gm = M.extract_features(fix, mov, kw["mind_r"], kw["mind_d"], True, mf, mm, device=torch.device(DEV), dtype=torch.float32)
    rest = {k: v for k, v in kw.items() if k not in ("mind_r", "mind_d")}
    out = field(M.register_pair_device(feat_fixed=gf[0], feat_moving=gm[0], **rest))
    ff, _ = orc.replicate_fill(fix.numpy(), mf.numpy())
    fm, _ = orc.replicate_fill(mov.numpy(), mm.numpy())
    ref = orc.convex_adam_pipeline(ff, fm, **kw)
    return np.array_equal(out, ref), ("masked", shape, kw)


def trial_labels(rng, t):
    from convexadam_amd import convex_adam_nnUNet as N
    gs, hw = int(rng.choice([2, 3, 4])), int(rng.integers(1, 5))
    shape = tuple(int(max(2 * gs, 8) + rng.integers(0, 30)) for _ in range(3))
    nlab = int(rng.choice([3, 9, 15, 16, 17, 33, 40]))
    blocks = rng.integers(0, nlab, [max(1, s // 5) for s in shape])
    lab = np.kron(blocks, np.ones((5, 5, 5), np.int64))
    lab = np.pad(lab, [(0, max(0, s - l)) for s, l in zip(shape, lab.shape)], mode="edge")[:shape[0], :shape[1], :shape[2]].astype(np.float32)
    lab2 = np.roll(lab, (1, -1, 1), (0, 1, 2))
    kw = dict(lambda_weight=1.25, grid_sp=gs, disp_hw=hw, selected_niter=int(rng.integers(1, 4)), selected_smooth=int(rng.choice([0, 3, 5])),
              grid_sp_adam=int(rng.choice([1, 2])), ic=bool(rng.integers(0, 2)))
    ff, fm = N.extract_features(torch.from_numpy(lab), torch.from_numpy(lab2), device=DEV)
    of, om, _ = orc.label_features(lab, lab2)
    ok = np.array_equal(host(ff)[0], of) and np.array_equal(host(fm)[0], om)
    out = field(M.register_pair_device(feat_fixed=ff[0], feat_moving=fm[0], cost_scale=12.0, **kw))
    ref = orc.convex_adam_pipeline(None, None, features=(of, om), **kw)
    return ok and np.array_equal(out, ref), ("labels", shape, nlab, kw)


def trial_adam(rng, t):
    from convexadam_amd import convexAdam_hyper_util as HU
    shape = tuple(int(rng.integers(3, 60)) for _ in range(2)) + (int(rng.integers(3, 150)),)
    C = int(rng.choice([1, 3, 4, 5, 12, 13, 16, 24, 33]))
    F2 = rng.random((C,) + shape, dtype=np.float32)
    M2 = rng.random((C,) + shape, dtype=np.float32)
    P0 = (float(rng.choice([0.3, 1.0, 3.0])) * rng.standard_normal((3,) + shape)).astype(np.float32)
    lam, nit = float(rng.choice([0.5, 1.0, 1.25])), int(rng.integers(1, 5))
    mod, sm, storage = None, None, "fp32"
    _draw_options(rng)
    r = rng.random()
    if r > 0.7:                                 # round 4: adam_mode "fast" (orc_adam_run_fast), also with resumed state; round 5: half-precision records
        st16 = "fp16" if rng.random() < 0.3 else "fp32"
        h16 = (lambda a: a.astype(np.float16).astype(np.float32)) if st16 == "fp16" else (lambda a: a)
        Ud, st = U.adam_run(dev(F2)[None], dev(M2)[None], dev(P0)[None], lam, nit, return_state=True, mode="fast", storage=st16)
        ref = orc.adam_run(h16(F2), h16(M2), P0, lam, nit, want_grad=True, mode="fast")
        ok = all(np.array_equal(host(st[k])[0], ref[k]) for k in ("P", "m", "v", "G")) and np.array_equal(host(Ud)[0], ref["U"])
        return ok, ("adam-fast", shape, C, lam, nit)
    if r < 0.2:
        mod, sm = HU.GaussianSmoothing(0.7), None
        sm = orc.make_smoother(gauss_w=np.array(list(mod.spec.gauss_w), np.float32))
    elif r < 0.4:
        mod = HU.kovesi_spline(float(rng.choice([1.6, 1.9, 2.8])), 4)
        sm = orc.make_smoother(mod.sizes)
    elif r < 0.55:
        storage = "fp16"
    Ud, st = U.adam_run(dev(F2)[None], dev(M2)[None], dev(P0)[None], lam, nit, return_state=True, smoother=mod, storage=storage)
    h = (lambda a: a.astype(np.float16).astype(np.float32)) if storage == "fp16" else (lambda a: a)
    ref = orc.adam_run(h(F2), h(M2), P0, lam, nit, want_grad=True, smoother=sm)
    ok = all(np.array_equal(host(st[k])[0], ref[k]) for k in ("P", "m", "v")) and np.array_equal(host(Ud)[0], ref["U"])
    return ok, ("adam", shape, C, lam, nit, storage, None if mod is None else type(mod).__name__)


def trial_mind(rng, t):
    shape = tuple(int(rng.integers(4, 50)) for _ in range(2)) + (int(rng.choice([5, 31, 32, 33, 64, 70, 96, 100, 130])),)
    r, d = int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 3, 4]))
    img = phantom(shape, 7000 + t, 8000 + t)
    got = host(U.MINDSSC(img.to(DEV)[None, None], r, d, device=torch.device(DEV)))[0]
    return np.array_equal(got, orc.mindssc(img.numpy(), r, d)), ("mind", shape, r, d)


def trial_convex_ops(rng, t):
    hw = int(rng.integers(1, 9))
    shape = tuple(int(rng.integers(2, 30)) for _ in range(2)) + (int(rng.integers(2, 60)),)
    C = int(rng.choice([1, 5, 12, 16, 20, 33]))
    while (2 * hw + 1) ** 3 * np.prod(shape) > 4e7:
        hw -= 1
    f = rng.random((C,) + shape, dtype=np.float32)
    m = rng.random((C,) + shape, dtype=np.float32)
    if rng.random() < 0.3:                                   # nulls: torch.argmin returns the first NaN of a column
        f.flat[rng.integers(0, f.size, 3)] = np.nan
    cost, n_box = str(rng.choice(["ssd", "ssd", "sad"])), int(rng.choice([2, 2, 1]))
    ssd, am = U.correlate(dev(f)[None], dev(m)[None], hw, 1, shape, C, cost=cost, n_box=n_box)
    rs, ra = orc.correlate(f, m, hw, cost=cost, n_box=n_box)
    ok = np.array_equal(host(ssd), rs, equal_nan=True) and np.array_equal(host(am), ra)
    mesh = orc.disp_mesh(hw)
    soft = U.coupled_convex(ssd, am, dev(mesh)[:, :, None], 1, shape)
    ok = ok and np.array_equal(host(soft)[0], orc.coupled_convex(rs, ra, mesh, hw), equal_nan=True)
    return ok, ("convex_ops", shape, C, hw, cost, n_box)


def trial_operators(rng, t):
    """inverse consistency, trilinear resize, grid_sample, box smoothing, stride pooling, feature transform"""
    from scipy.ndimage import distance_transform_edt as edt
    sh = tuple(int(rng.integers(2, 28)) for _ in range(3))
    a = (0.3 * rng.standard_normal((3,) + sh)).astype(np.float32)
    b = (0.3 * rng.standard_normal((3,) + sh)).astype(np.float32)
    it = int(rng.choice([1, 3, 15]))
    o1, o2 = U.inverse_consistency(dev(a)[None], dev(b)[None], iter=it)
    r1, r2 = orc.inverse_consistency(a, b, it)
    ok = np.array_equal(host(o1)[0], r1) and np.array_equal(host(o2)[0], r2)
    C = int(rng.choice([1, 2, 3, 5]))
    dst = tuple(int(rng.integers(1, 60)) for _ in range(2)) + (int(rng.integers(1, 200)),)
    x = rng.standard_normal((C,) + sh).astype(np.float32)
    ok = ok and np.array_equal(host(U.resize_trilinear(dev(x)[None], dst))[0], orc.resize_trilinear(x, dst))
    grid = (rng.random(dst[:2] + (min(dst[2], 40), 3), dtype=np.float32) * 2.8 - 1.4).astype(np.float32)
    ok = ok and np.array_equal(host(U.grid_sample(dev(x)[None], dev(grid)[None]))[0], orc.grid_sample(x, grid))
    k, passes = int(rng.choice([3, 5, 7])), int(rng.integers(1, 4))
    r = x
    for _ in range(passes):
        r = orc.box_zero(r, k)
    ok = ok and np.array_equal(host(U.box_smooth(dev(x)[None], k, passes))[0], r)
    g = int(rng.integers(1, 7))
    big = rng.standard_normal((C,) + tuple(s + g for s in sh)).astype(np.float32)
    ok = ok and np.array_equal(host(U.avg_pool(dev(big)[None], g))[0], orc.avgpool_stride(big, g))
    m = rng.random(sh) < float(rng.choice([0.3, 0.8, 0.98]))
    if m.any() and not m.all():
        got = host(M.feature_transform(dev((~m).astype(np.float32))))
        ok = ok and np.array_equal(got, edt(~m, return_indices=True)[1])
    return ok, ("operators", sh, C, dst, it, k, passes, g)


def trial_metrics(rng, t):
    from convexadam_amd import convexAdam_hyper_util as HU
    from oracle import metrics_oracle as morc
    from scipy.ndimage import distance_transform_edt as edt
    sh = tuple(int(rng.integers(3, 36)) for _ in range(3))
    conv = bool(rng.integers(0, 2))
    flow = (rng.standard_normal((3,) + sh) * (0.1 if conv else 2.0)).astype(np.float32)
    ok = np.array_equal(host(HU.jacobian_determinant_3d(dev(flow)[None], conv)), morc.jacobian_determinant_3d(flow, conv))
    nl = int(rng.integers(2, 12))
    seg = rng.integers(0, nl, sh).astype(np.float32)
    seg2 = np.roll(seg, (1, 0, -1), (0, 1, 2))
    disp = (rng.standard_normal((3,) + sh) * 2.5).astype(np.float32)
    w = HU.warp_labels_nearest(dev(seg), dev(disp)[None])
    ok = ok and np.array_equal(host(w), morc.warp_labels_nearest(seg, disp))
    ok = ok and np.array_equal(HU.dice_coeff(dev(seg2), w, nl).numpy(), morc.dice_coeff(seg2, host(w), nl))
    m = (rng.random(sh) < float(rng.choice([0.5, 0.9, 0.99]))).astype(np.float32)
    if (m == 0).any():
        ok = ok and np.array_equal(host(HU.edt_squared(dev(m))).astype(np.int64), np.rint(edt(m) ** 2).astype(np.int64))
    return ok, ("metrics", sh, conv, nl)


def trial_hd95(rng, t):
    """cupy_hd95: the surface-only path (bit planes + cube / ring searches) against the transforms and the scipy restatement, on blob and
    noise label maps with rows of 1 .. 3+ words, labels missing from either map, and a small search radius now and then (hand-over)."""
    from convexadam_amd import convexAdam_hyper_util as HU
    from oracle import metrics_oracle as morc
    sh = (int(rng.integers(3, 40)), int(rng.integers(3, 40)), int(rng.choice([rng.integers(3, 40), rng.integers(60, 70), rng.integers(120, 200)])))
    nl = int(rng.integers(1, 16))
    if rng.integers(0, 3):
        cell = int(rng.integers(2, 7))
        a = rng.integers(0, nl + 1, [max(1, -(-s // cell)) for s in sh])
        a = np.kron(a, np.ones((cell,) * 3, np.int64))[: sh[0], : sh[1], : sh[2]]
    else:
        a = rng.integers(0, nl + 1, sh)
    b = np.roll(a, tuple(int(v) for v in rng.integers(-3, 4, 3)), (0, 1, 2))
    if rng.integers(0, 2):
        b = np.where(rng.random(sh) < 0.02, rng.integers(0, nl + 1, sh), b)
    if nl > 1 and rng.integers(0, 2):
        b[b == int(rng.integers(1, nl + 1))] = 0
    fa, fb = dev(a.astype(np.float32)), dev(b.astype(np.float32))
    old = HU.HD95_SURFACE_MAX_RADIUS
    HU.HD95_SURFACE_MAX_RADIUS = int(rng.choice([old, old, 2, 6]))
    try:
        res, errs = [], []
        for call in (lambda: host(HU.cupy_hd95(fa, fb, nl)), lambda: host(HU.cupy_hd95(fa, fb, nl, method="edt")), lambda: morc.hd95(a, b, nl, 1)):
            try:
                res.append(call()); errs.append(False)
            except (RuntimeError, ValueError):           # a label filling a whole map: no outside voxel (scipy: undefined feature transform)
                res.append(None); errs.append(True)
        # a label that fills a whole map has no outside voxel: scipy's (and cupyx's) transform of a volume without background is undefined,
        # so the restatement is no witness there (seed 20260929, trial 6071: label 1 fills BOTH 17 x 5 x 3 maps -- no surface voxel at all,
        # both device paths return the percentile of an empty set, NaN, the scipy form 1.0); the two device paths must still agree
        full = any((a == q).all() or (b == q).all() for q in range(1, nl + 1))
        if any(errs[:2]):
            ok = errs[0] == errs[1]
        else:
            ok = np.array_equal(res[0], res[1], equal_nan=True) and (errs[2] or full or np.array_equal(res[0], res[2], equal_nan=True))
        radius = HU.HD95_SURFACE_MAX_RADIUS
        if not ok:                                           # keep the case (gpurun_out/ travels back from the GPU box)
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fuzz_hd95_fail_%d.npz" % t), a=a, b=b, nl=nl, radius=radius,
                                    surface=np.asarray(res[0] if res[0] is not None else []), edt=np.asarray(res[1] if res[1] is not None else []),
                                    scipy=np.asarray(res[2] if res[2] is not None else []), errs=np.asarray(errs))
            except Exception:
                pass
    finally:
        HU.HD95_SURFACE_MAX_RADIUS = old
    return ok, ("hd95", sh, nl, radius)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--large", action="store_true", help="pipeline trials on extents of 96 .. 224 voxels")
    ap.add_argument("--reference-bits", action="store_true",
                    help="library AND oracle in reference-bits mode: the golden host's MKL exp / sqrt tables, torch's thread-count dependent mean")
    a = ap.parse_args()
    global LARGE
    LARGE = a.large
    orc.build()
    if a.reference_bits:
        import mkl_tables
        from convexadam_amd import reference_bits as rb
        t = mkl_tables.golden_tables()
        rb.set_mind_exp_table(t["exp"], t["exp_first"], t["exp_count"], device=DEV)
        rb.set_adam_sqrt_table(t["sqrt"], device=DEV)
        orc.set_exp_table(t["exp"], t["exp_first"], t["exp_count"])
        orc.set_sqrt_table(t["sqrt"])
    rng = np.random.default_rng(a.seed)
    kinds = [trial_pipeline, trial_pipeline, trial_masked, trial_labels, trial_adam, trial_adam, trial_convex_ops, trial_mind, trial_operators, trial_metrics, trial_hd95]
    if a.only:
        kinds = [k for k in kinds if any(tok in k.__name__ for tok in a.only.replace(',', ' ').split())]
    t0, n, bad, count = time.time(), 0, [], {}
    while time.time() - t0 < a.minutes * 60:
        k = kinds[n % len(kinds)]
        if a.reference_bits:                                 # the mean of MINDSSC as torch computes it with T threads
            T = int(rng.choice([1, 2, 3, 8, 16]))
            rb.set_mean_threads(T)
            orc.set_mean_threads(T)
        try:
            ok, what = k(rng, n)
        except Exception as e:                               # an explicit error for a supported configuration is a finding too
            ok, what = False, (k.__name__, "EXCEPTION", repr(e)[:300])
        count[k.__name__] = count.get(k.__name__, 0) + 1
        if not ok:
            bad.append(what)
            print("MISMATCH seed=%d trial=%d: %r" % (a.seed, n, what), flush=True)
        n += 1
    kindsum = {}
    for w in bad:
        key = (w[0], w[1]) if w[1] == "EXCEPTION" else (w[0],)
        kindsum[key + ((w[2][:120],) if w[1] == "EXCEPTION" else ())] = kindsum.get(key + ((w[2][:120],) if w[1] == "EXCEPTION" else ()), 0) + 1
    for k, v in kindsum.items():
        print("  %4d x %r" % (v, k), flush=True)
    print("fuzz campaign: %d trials in %.1f min (%s), %d mismatches" % (n, (time.time() - t0) / 60, count, len(bad)), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
