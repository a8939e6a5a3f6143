"""Golden for the reference's EVEN `selected_smooth` (convex_adam_MIND.py:184-191), captured from the reference itself (run ONLY in the
build container):

    python tests/golden/make_golden_even_smooth.py        # -> tests/golden/even_smooth.npz

The reference announces "+1" for an even kernel and then overwrites its own fix (:189), so the three avg_pool3d(k, stride 1, padding k//2)
each GROW the field by one voxel per axis: the function returns (H+3, W+3, D+3, 3).  Inputs are regenerated from seeds by the tests
(convexadam_amd.phantom); only the reference's outputs are stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402
from convexadam_amd.phantom import phantom  # noqa: E402


def main():
    _, mind = import_reference()
    torch.set_num_threads(4)
    out = {}
    shape = (28, 24, 36)
    fix = phantom(shape, 7, 70)
    mov = torch.roll(phantom(shape, 7, 71), (1, -1, 2), (0, 1, 2))
    for k in (2, 4):
        kw = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=4, disp_hw=2, selected_niter=3, selected_smooth=k, grid_sp_adam=2, ic=True)
        r = mind.convex_adam_pt(fix.clone(), mov.clone(), dtype=torch.float32, device=torch.device("cpu"), verbose=False, **kw)
        assert r.shape == tuple(s + 3 for s in shape) + (3,), r.shape
        out["k%d" % k] = r.astype(np.float32)
        assert np.array_equal(out["k%d" % k].astype(np.float64), r)
    # the building block alone: one growing pool of a random field
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 3, 6, 9, 7, generator=g)
    for k in (2, 4, 6):
        out["pool%d" % k] = torch.nn.functional.avg_pool3d(x, k, padding=k // 2, stride=1)[0].numpy()
    out["pool_in"] = x[0].numpy()
    out["shape"] = np.array(shape)
    path = os.path.join(HERE, "even_smooth.npz")
    np.savez_compressed(path, **out)
    print("wrote even_smooth.npz %.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
