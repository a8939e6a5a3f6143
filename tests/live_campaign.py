#!/usr/bin/env python
"""Long randomised cross-check of the CPU oracle against the upstream reference ITSELF (build container only: reads /root/reference
through tests/golden/_ref_import.py; listed in .gpurunignore, never part of the suite):
    python tests/live_campaign.py [--minutes 15] [--seed 1]
Random small pairs -- extents, MIND radius / dilation, both grid spacings, search half-width, lambda, iterations, inverse consistency,
final smoothing, zero backgrounds, masked pairs, thread counts -- through the reference's convex_adam_pt on the CPU and through the
oracle given THIS host's MKL exp / sqrt tables and the run's thread count: the two fields must be equal bit for bit."""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
from _ref_import import import_reference          # noqa: E402
import mkl_tables                                 # noqa: E402
from convexadam_amd.phantom import ellipsoid_mask, phantom   # noqa: E402  (host-side synthetic images only)
from oracle import oracle as orc                  # noqa: E402


class NNUNetHarness:
    """The reference's label-map pipeline (convex_adam_nnUNet.py:41-159) executed unmodified with its CUDA / fp16 / nibabel calls
    neutralised from the outside, as tests/golden/make_golden_nnunet.py does: Tensor.cuda / Module.cuda -> identity, Tensor.half ->
    float32, nibabel -> an in-memory stand-in that hands over the label arrays and captures the field."""

    def __init__(self):
        import importlib
        nib = sys.modules["nibabel"]
        self.store, self.captured = {}, {}
        store, captured = self.store, self.captured

        class _Img:
            def __init__(self, arr):
                self._a = arr
                self.affine = np.eye(4)

            def get_fdata(self):
                return self._a

        nib.load = lambda path: _Img(store[path])

        def _nifti(arr, affine):
            captured["disp"] = np.array(arr)
            return object()
        nib.Nifti1Image = _nifti
        nib.save = lambda img, path: None
        torch.Tensor.cuda = lambda s, *a, **k: s
        torch.Tensor.half = lambda s, *a, **k: s.float()
        torch.cuda.synchronize = lambda *a, **k: None
        torch.nn.Module.cuda = lambda s, *a, **k: s
        self.N = importlib.import_module("convexAdam.convex_adam_nnUNet")

    def run(self, lab, lab2, lam, gs, hw, niter, smooth, gsa, ic):
        self.store["fix"], self.store["mov"] = lab.astype(np.float64), lab2.astype(np.float64)
        self.captured.clear()
        self.N.convex_adam("fix", "mov", lam, gs, hw, niter, smooth, gsa, ic, "/tmp")
        return self.captured["disp"]


def trial_labels(rng, n, harness):
    gs, hw, gsa = int(rng.choice([2, 3, 4])), int(rng.integers(1, 4)), int(rng.choice([1, 2]))
    shape = tuple(int(max(3 * gs, 3 * gsa, 8) + rng.integers(0, 26)) for _ in range(3))
    nlab = int(rng.choice([3, 9, 15, 16, 17, 33]))
    blocks = rng.integers(0, nlab, [max(1, s // 5) for s in shape])
    lab = np.kron(blocks, np.ones((5, 5, 5), np.int64))
    lab = np.pad(lab, [(0, max(0, s - l)) for s, l in zip(shape, lab.shape)], mode="edge")[:shape[0], :shape[1], :shape[2]].astype(np.float32)
    lab2 = np.roll(lab, (1, -1, 1), (0, 1, 2))
    lab[0, 0, 0] = nlab - 1; lab2[-1, -1, -1] = nlab - 1               # the reference needs equal max labels in both maps
    lam = float(rng.choice([0.0, 1.25]))
    niter, smooth, ic = int(rng.integers(1, 7)), int(rng.choice([0, 3, 5])), bool(rng.integers(0, 2))
    out = harness.run(lab, lab2, lam, gs, hw, niter, smooth, gsa, ic)
    of, om, _ = orc.label_features(lab, lab2)
    ref = orc.convex_adam_pipeline(None, None, features=(of, om), lambda_weight=lam, grid_sp=gs, disp_hw=hw, selected_niter=niter,
                                   selected_smooth=smooth, grid_sp_adam=gsa, ic=ic)
    return np.array_equal(out, ref), ("labels", shape, nlab, lam, gs, hw, niter, smooth, gsa, ic)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=15.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--labels", action="store_true", help="the label-map (nnUNet) pipeline instead of the MIND one")
    a = ap.parse_args()
    orc.build()
    utils, mind = import_reference()
    t = mkl_tables.host_tables(orc)
    orc.set_exp_table(t["exp"], t["exp_first"], t["exp_count"])
    orc.set_sqrt_table(t["sqrt"])
    rng = np.random.default_rng(a.seed)
    harness = NNUNetHarness() if a.labels else None
    t0, n, bad, kinds = time.time(), 0, 0, {}
    while time.time() - t0 < a.minutes * 60:
        threads = int(rng.choice([1, 2, 3, 8]))
        torch.set_num_threads(threads)
        orc.set_mean_threads(threads)
        if a.labels:
            try:
                ok, what = trial_labels(rng, n, harness)
            except Exception as e:                         # noqa: BLE001
                ok, what = False, ("labels", "EXCEPTION", repr(e)[:300])
            kinds["labels"] = kinds.get("labels", 0) + 1
            if not ok:
                bad += 1
                print("MISMATCH seed=%d trial=%d threads=%d %r" % (a.seed, n, threads, what), flush=True)
            n += 1
            continue
        gs, gsa, hw = int(rng.choice([2, 3, 4, 5, 6])), int(rng.choice([1, 2, 3])), int(rng.integers(1, 6))
        masked = rng.random() < 0.25
        if masked:
            shape = tuple(int(2 * (max(2 * gs, 2 * gsa, 5) + rng.integers(0, 14))) for _ in range(3))
        else:                                              # (coarse extents below 3 voxels: torch's avg_pool3d refuses them in the reference)
            shape = tuple(int(max(3 * gs, 3 * gsa, 8) + rng.integers(0, 34)) for _ in range(3))
        kw = dict(mind_r=int(rng.choice([1, 2, 3])), mind_d=int(rng.choice([1, 2, 3])), grid_sp=gs, disp_hw=hw, grid_sp_adam=gsa,
                  lambda_weight=float(rng.choice([0.0, 0.7, 1.25])), selected_niter=int(rng.integers(1, 12)), ic=bool(rng.integers(0, 2)),
                  selected_smooth=int(rng.choice([0, 0, 3, 5])))
        fix = phantom(shape, 10 + n, 1000 + n)
        mov = torch.roll(phantom(shape, 10 + n, 2000 + n), (1, -1, 2), (0, 1, 2))
        kind = "masked" if masked else "plain"
        try:
            if masked:
                mf = ellipsoid_mask(shape, float(rng.uniform(0.28, 0.45)))
                mm = torch.roll(mf, (1, 0, -1), (0, 1, 2))
                ff, fm = mind.extract_features(fix, mov, kw["mind_r"], kw["mind_d"], True, mf, mm, device=torch.device("cpu"), dtype=torch.float32)
                of, _ = orc.replicate_fill(fix.numpy(), mf.numpy())
                om, _ = orc.replicate_fill(mov.numpy(), mm.numpy())
                ok = np.array_equal(ff[0].numpy(), orc.mindssc(of, kw["mind_r"], kw["mind_d"])) and np.array_equal(fm[0].numpy(), orc.mindssc(om, kw["mind_r"], kw["mind_d"]))
            else:
                if rng.random() < 0.3:                     # exact-zero background: flat regions, clamped variances, ties in every argmin
                    m = ellipsoid_mask(shape, 0.38)
                    fix, mov, kind = fix * m, mov * m, "zero background"
                out = mind.convex_adam_pt(fix, mov, dtype=torch.float32, device=torch.device("cpu"), **kw)
                ok = np.array_equal(out, orc.convex_adam_pipeline(fix.numpy(), mov.numpy(), **kw))
        except Exception as e:                             # noqa: BLE001
            ok = False
            print("EXCEPTION %r" % (repr(e)[:300],), flush=True)
        kinds[kind] = kinds.get(kind, 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH seed=%d trial=%d threads=%d %s shape=%s %r" % (a.seed, n, threads, kind, shape, kw), flush=True)
        n += 1
    print("live campaign (oracle vs the reference on this host, reference-bits mode): %d trials in %.1f min %s, %d mismatches" % (
        n, (time.time() - t0) / 60, kinds, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
