"""Golden vectors for BASELINE configs[3] end to end: the reference's multi-channel pipeline itself
(/root/reference/src/convexAdam/convex_adam_nnUNet.py:41-159 `convex_adam`, label maps in, displacement field out) run in the build
container on a synthetic label pair with 18 labels (C >= 16: ATen's cascade channel sum is exercised), captured at four horizons:
convex stage only (lambda_weight = 0) and 1 / 5 / 20 Adam iterations.

    python tests/golden/make_golden_nnunet.py        ->  tests/golden/nnunet.npz   (inputs + reference outputs, no source text)

The reference function is CUDA / fp16 / nibabel bound; it is executed unmodified with those calls neutralised from the outside
(SURVEY.md appendix A): Tensor.cuda / Module.cuda -> identity, Tensor.half -> float32, torch.cuda.synchronize -> no-op, nibabel
replaced by an in-memory stand-in that hands over the label arrays and captures the array given to Nifti1Image.  So the capture is the
reference's float32 CPU evaluation of its own code path (fp16 storage is the separate `storage="fp16"` mode here).
Only usable where /root/reference exists; listed in .gpurunignore."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import import_reference  # noqa: E402

CFG = dict(grid_sp=4, disp_hw=3, selected_smooth=0, grid_sp_adam=2, ic=True)
SHAPE = (40, 32, 48)


def label_pair():
    """18-label maps: argmax of smooth random fields; the moving map is the fixed one pulled through a smooth warp (nearest)."""
    g = torch.Generator().manual_seed(11)
    f = F.interpolate(torch.randn(1, 18, 6, 5, 7, generator=g), size=SHAPE, mode="trilinear", align_corners=False)
    lab = torch.argmax(f, 1)[0].float()
    base = F.affine_grid(torch.eye(3, 4)[None], (1, 1) + SHAPE, align_corners=False)
    warp = F.interpolate(torch.randn(1, 3, 4, 4, 4, generator=g) * 0.3, size=SHAPE, mode="trilinear", align_corners=False)
    labm = F.grid_sample(lab[None, None], base + warp.permute(0, 2, 3, 4, 1), mode="nearest", padding_mode="border", align_corners=False)[0, 0]
    lab[0, 0, 0] = 17.0          # the reference needs equal max labels in both maps (bincount / one_hot sizes)
    labm[-1, -1, -1] = 17.0
    return lab.contiguous(), labm.contiguous()


def main():
    torch.set_num_threads(8)
    import_reference()
    nib = sys.modules["nibabel"]
    store, captured = {}, {}

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

    saved = (torch.Tensor.cuda, torch.Tensor.half, torch.cuda.synchronize, torch.nn.Module.cuda)
    torch.Tensor.cuda = lambda s, *a, **k: s
    torch.Tensor.half = lambda s, *a, **k: s.float()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.nn.Module.cuda = lambda s, *a, **k: s
    try:
        import importlib
        N = importlib.import_module("convexAdam.convex_adam_nnUNet")
        lab, labm = label_pair()
        store["fix"], store["mov"] = lab.double().numpy(), labm.double().numpy()
        f_fix, f_mov = N.extract_features(lab, labm)
        out = dict(lab_fix=lab.numpy().astype(np.int16), lab_mov=labm.numpy().astype(np.int16),
                   weights=(f_fix[0].amax((1, 2, 3)) / 10).numpy(),      # not used for parity (10 * w is rounded): see feat_max
                   feat_max=f_fix[0].amax((1, 2, 3)).numpy(), n_ch=np.int32(f_fix.shape[1]),
                   feat_fix_sum=f_fix[0].double().sum((1, 2, 3)).numpy(), feat_mov_sum=f_mov[0].double().sum((1, 2, 3)).numpy(),
                   cfg=np.array([CFG["grid_sp"], CFG["disp_hw"], CFG["grid_sp_adam"]], np.int32))
        for name, lam, niter in (("convex", 0.0, 0), ("adam_1", 1.25, 1), ("adam_5", 1.25, 5), ("adam_20", 1.25, 20)):
            captured.clear()
            N.convex_adam("fix", "mov", lam, CFG["grid_sp"], CFG["disp_hw"], niter, CFG["selected_smooth"], CFG["grid_sp_adam"], CFG["ic"], "/tmp")
            d = captured["disp"]                                   # (H, W, D, 3) float64 holding float32 values
            assert d.shape == SHAPE + (3,) and np.array_equal(d, d.astype(np.float32).astype(np.float64))
            # full fields at the first and the last horizon, every second voxel per axis in between (fixture size)
            out[name] = d.astype(np.float32) if name in ("convex", "adam_20") else d[::2, ::2, ::2].astype(np.float32)
            out[name + "_sum"] = d.sum((0, 1, 2))                  # float64 sums of the whole field
            out[name + "_sumsq"] = (d * d).sum((0, 1, 2))
            print(name, "mean |u| %.4f" % np.abs(d).mean(), flush=True)
        np.savez_compressed(os.path.join(HERE, "nnunet.npz"), **out)
        print("wrote nnunet.npz: C =", int(out["n_ch"]), {k: v.shape for k, v in out.items() if k.startswith(("convex", "adam"))})
    finally:
        torch.Tensor.cuda, torch.Tensor.half, torch.cuda.synchronize, torch.nn.Module.cuda = saved


if __name__ == "__main__":
    main()
