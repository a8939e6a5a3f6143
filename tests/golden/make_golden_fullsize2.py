"""More full-size reference captures at 80 Adam iterations: the acceptance of adam_mode="fast" rested on ONE
pair.  Run ONLY in the build container:

    python tests/golden/make_golden_fullsize2.py [a] [b] [c]      # -> tests/golden/fullsize2.npz  (about 15 minutes on 8 cores)

  a  another seed and a larger warp amplitude           phantom.deformed_pair((160,192,224), 2, 6.0)
  b  the benchmark pair with an EXACT-zero background   phantom.zero_background_pair((160,192,224), 0, 4.0)
  c  18-label maps through the reference's nnUNet path  phantom.warped_label_pair((160,192,160), 18): convex_adam_nnUNet.convex_adam
     (src/convexAdam/convex_adam_nnUNet.py:41-159; C = 18 >= 16 channels: ATen's cascade channel sum)

Each case: the reference's own function, unmodified, observed through the wrappers of make_golden_fullsize.py (F.interpolate,
Adam.step, F.grid_sample); a second run multiplies the warped features by 1 + 6e-8 N(0,1) -- the reference's distance from a
1-ulp-perturbed copy of ITSELF.  Stored per case <t>: <t>_coarse_ic, <t>_adam_<n>_sub / _sum / _sumsq at n = 1, 20, 40, 80,
<t>_self_perturbation_epe(_sub), <t>_snaps.  No reference source text is reproduced; inputs are regenerated from seeds by
convexadam_amd/phantom.py.  Listed in .gpurunignore (the reference never travels to the GPU box)."""
import importlib
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden_fullsize as G1  # noqa: E402  (Capture, describe, epe; imports the reference)
from convexadam_amd.phantom import deformed_pair, warped_label_pair, zero_background_pair  # noqa: E402

M = G1.M
CPU = torch.device("cpu")
SUB = G1.SUB
SNAPS = (1, 20, 40, 80)


class Capture(G1.Capture):
    """make_golden_fullsize.Capture with the perturbation on every feature warp (any channel count > 3; the inverse-consistency step
    warps 3-channel fields through grid_sample as well and stays untouched)."""

    def __enter__(self):
        super().__enter__()
        cap = self

        def gs(inp, grid, *a, **k):
            out = cap._gs(inp, grid, *a, **k)
            if cap.gen is not None and inp.shape[1] > 3:
                out = out * (1.0 + 6e-8 * torch.randn(out.shape, generator=cap.gen))
            return out
        F.grid_sample = gs
        return self


def run_case(out, tag, call, shape, t0):
    """call() runs the reference once and returns its (3,H,W,D) float32 field (or None when the caller's output is quantised)."""
    with Capture(SNAPS) as c:
        final = call()
    print("%s reference run: %.0f s" % (tag, time.time() - t0), flush=True)
    out[tag + "_coarse_ic"] = c.coarse.numpy()
    fields = {n: c.field(n, 2, shape) for n in SNAPS}
    if final is not None:
        assert torch.equal(fields[80], final), "capture does not reproduce the returned field"
    for n in SNAPS:
        G1.describe(out, "%s_adam_%d" % (tag, n), fields[n])
    with Capture(SNAPS, perturb_seed=99) as cp:
        call()
    assert torch.equal(cp.coarse, c.coarse)
    out[tag + "_self_perturbation_epe"] = np.array([G1.epe(cp.field(n, 2, shape), fields[n]) for n in SNAPS])
    out[tag + "_self_perturbation_epe_sub"] = np.array([G1.epe(cp.field(n, 2, shape)[:, ::SUB, ::SUB, ::SUB], fields[n][:, ::SUB, ::SUB, ::SUB]) for n in SNAPS])
    out[tag + "_snaps"] = np.array(SNAPS)
    out[tag + "_mean_abs"] = np.array([float(fields[n].abs().mean()) for n in SNAPS])
    print("%s perturbed run done: %.0f s; self-perturbation EPE %s; mean |u| %s" % (tag, time.time() - t0, out[tag + "_self_perturbation_epe"], out[tag + "_mean_abs"]), flush=True)
    return fields


def main():
    torch.set_num_threads(8)
    which = set(sys.argv[1:]) or {"a", "b", "c"}
    path = os.path.join(HERE, "fullsize2.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    out["sub"] = np.int64(SUB)
    t0 = time.time()
    shape = (160, 192, 224)
    kw = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=6, disp_hw=6, selected_niter=80, selected_smooth=0, grid_sp_adam=2,
              ic=True, dtype=torch.float32, device=CPU)
    if "a" in which:
        fix, mov = deformed_pair(shape, 2, 6.0)
        run_case(out, "c4", lambda: torch.from_numpy(M.convex_adam_pt(fix, mov, **kw)).permute(3, 0, 1, 2).float(), shape, t0)
        np.savez_compressed(path, **out)
    if "b" in which:
        fixz, movz = zero_background_pair(shape, 0, 4.0)
        out["c5_zero_fraction"] = np.array([float((fixz == 0).float().mean()), float((movz == 0).float().mean())])
        run_case(out, "c5", lambda: torch.from_numpy(M.convex_adam_pt(fixz, movz, **kw)).permute(3, 0, 1, 2).float(), shape, t0)
        np.savez_compressed(path, **out)
    if "c" in which:
        shape_l = (160, 192, 160)
        lab, labm = warped_label_pair(shape_l, 18, 11, 0.05)
        nib = sys.modules["nibabel"]
        store = {"fix": lab.double().numpy(), "mov": labm.double().numpy()}
        nib.load = lambda p: type("Img", (), {"get_fdata": staticmethod(lambda: store[p]), "affine": np.eye(4)})()
        nib.Nifti1Image = lambda arr, affine: object()
        nib.save = lambda img, p: None
        saved = (torch.Tensor.cuda, torch.Tensor.half, torch.cuda.synchronize, torch.nn.Module.cuda)
        torch.Tensor.cuda = lambda s, *a, **k: s
        torch.Tensor.half = lambda s, *a, **k: s.float()
        torch.cuda.synchronize = lambda *a, **k: None
        torch.nn.Module.cuda = lambda s, *a, **k: s
        try:
            N = importlib.import_module("convexAdam.convex_adam_nnUNet")
            ff, _ = N.extract_features(lab, labm)
            out["c6_n_ch"] = np.int32(ff.shape[1])
            del ff

            def call():
                N.convex_adam("fix", "mov", 1.25, 6, 6, 80, 0, 2, True, "/tmp")
                return None
            run_case(out, "c6", call, shape_l, t0)
        finally:
            torch.Tensor.cuda, torch.Tensor.half, torch.cuda.synchronize, torch.nn.Module.cuda = saved
        np.savez_compressed(path, **out)
    print("wrote fullsize2.npz %.1f KB: %s" % (os.path.getsize(path) / 1024, sorted(k for k in out if k.endswith("_snaps"))))


if __name__ == "__main__":
    main()
