"""Per-stage / per-kernel time of the reference-bits mode (the mode that meets the literal 1e-3 tolerance) on the benchmark pair."""
import lzma, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from convexadam_amd import reference_bits as rb
from convexadam_amd.convex_adam_MIND import last_profile, register_pair_device, set_profiling
from convexadam_amd.convex_adam_utils import sqrt_codes_from_low_bitmaps
from convexadam_amd.phantom import deformed_pair
dev = torch.device("cuda:0")
fix, mov = deformed_pair((160, 192, 224), 0, 4.0); fix, mov = fix.to(dev), mov.to(dev)
gd = os.path.join(ROOT, "tests", "golden")
exp_tbl = np.frombuffer(lzma.decompress(open(os.path.join(gd, "mkl_vsexp_codes.xz"), "rb").read()), np.uint8)
q = np.load(os.path.join(gd, "mkl_vssqrt_low.npz"))
rb.set_mind_exp_table(exp_tbl, device=dev); rb.set_adam_sqrt_table(sqrt_codes_from_low_bitmaps(q["normal"], q["denormal"]), device=dev); rb.set_mean_threads(8)
CFG = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=6, disp_hw=6, selected_niter=80, selected_smooth=0, grid_sp_adam=2, ic=True, adam_mode="exact")
for _ in range(2): register_pair_device(fix, mov, **CFG)
torch.cuda.synchronize(); set_profiling(2)
t0 = time.perf_counter()
for _ in range(5): register_pair_device(fix, mov, **CFG)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
st = {}
for n, t in last_profile(): st.setdefault(n, []).append(t)
print("reference-bits mode: %.2f ms per pair;" % ms, {k: round(sum(v) / len(v), 3) for k, v in st.items()})
rb.disable()
