import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
from oracle import oracle as orc
from convexadam_amd.phantom import deformed_pair
cache = "/tmp/cvx_coarse_feats.npz"
if os.path.exists(cache):
    c = np.load(cache); fs, ms = c["fs"], c["ms"]
else:
    fix, mov = deformed_pair((160, 192, 224), 0, 4.0)
    fs = orc.avgpool_stride(orc.mindssc(fix.numpy(), 1, 2), 6); ms = orc.avgpool_stride(orc.mindssc(mov.numpy(), 1, 2), 6)
    np.savez(cache, fs=fs, ms=ms)
t = time.time(); ssd, am = orc.correlate(fs, ms, 6); print("correlate", time.time() - t)
K = ssd.shape[0]; S = ssd.reshape(K, -1)
part = np.partition(S, 1, axis=0)[:2]
best, sec = part[0], part[1]
rel = (sec - best) / np.maximum(best, 1e-30)
for thr in (1e-6, 4e-6, 1e-5, 4e-5, 1e-4, 1e-3):
    print("plain argmin: voxels with second within %.0e rel of best: %d of %d" % (thr, int((rel <= thr).sum()), rel.size))
print("best range", best.min(), best.max(), "zeros", int((best == 0).sum()))

