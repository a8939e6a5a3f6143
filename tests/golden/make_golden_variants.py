"""Goldens for the operator variants of the challenge scripts, captured by EXECUTING the reference's own
function definitions (run ONLY in the build container):

    python tests/golden/make_golden_variants.py        # -> tests/golden/variants.npz

The scripts l2r_2021_convexAdam_task2_docker.py / task3_docker.py run a whole challenge case at import time, so they cannot be
imported; instead their `correlate` FunctionDef nodes are lifted out of the parsed file with `ast`, compiled as they stand and
called with this script's globals for the names they expect (H, W, D, torch, F, time, gpu_usage).  No reference text is stored.
  task3 :41-66   SAD cost (`.abs().sum(0)`) + ONE avg_pool3d
  task2 :47-72   SSD cost + ONE avg_pool3d
"""
import ast
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CONVEXADAM_REFERENCE", "/root/reference")


def lift(script, name, **globs):
    tree = ast.parse(open(os.path.join(REF, script)).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = dict(torch=torch, F=F, time=time, gpu_usage=lambda: None, print=lambda *a, **k: None, **globs)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), script, "exec"), ns)
    return ns[name], ns


def main():
    torch.cuda.synchronize = lambda *a, **k: None
    g = torch.Generator().manual_seed(2021)
    out = {}
    for tag, script, shape, hw in (("sad1", "l2r_2021_convexAdam_task3_docker.py", (5, 6, 7), 2), ("ssd1", "l2r_2021_convexAdam_task2_docker.py", (6, 5, 9), 2),
                                   ("sad1_w", "l2r_2021_convexAdam_task3_docker.py", (3, 4, 37), 4)):
        f = torch.rand(1, 12, *shape, generator=g)
        m = torch.rand(1, 12, *shape, generator=g)
        fn, ns = lift(script, "correlate")
        ns["H"], ns["W"], ns["D"] = shape
        ssd, am = fn(f, m, hw, 1)
        out.update({tag + "_fix": f[0].numpy(), tag + "_mov": m[0].numpy(), tag + "_hw": np.int64(hw), tag + "_ssd": ssd.numpy()[::(7 if tag.endswith("_w") else 1)].copy(), tag + "_argmin": am.numpy(),
                    tag + "_ssd_sum": np.float64(ssd.double().sum().item())})
    path = os.path.join(HERE, "variants.npz")
    np.savez_compressed(path, **out)
    print("wrote variants.npz %.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
