"""How often does the variance clamp of MIND-SSC (convex_adam_utils.py:60-62) bind, and how many 6^3 blocks hold such a voxel?
(sizing of the single-pass MIND kernel's repair list; torch on the CPU, reference arithmetic order is irrelevant for a census)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from convexadam_amd import phantom as ph

OFF1 = [(0,0,-1),(0,-1,0),(0,-1,0),(0,0,1),(0,0,1),(1,0,0),(1,0,0),(1,0,0),(0,1,0),(0,1,0),(0,1,0),(0,1,0)]
OFF2 = [(-1,0,0),(-1,0,0),(0,0,-1),(-1,0,0),(0,-1,0),(0,0,-1),(0,-1,0),(0,0,1),(-1,0,0),(0,0,-1),(0,0,1),(1,0,0)]

def census(img, name, d=2, r=1):
    H, W, D = img.shape
    p = F.pad(img[None, None], (d,) * 6, mode="replicate")[0, 0]
    def sh(o):
        return p[d + d * o[0]: d + d * o[0] + H, d + d * o[1]: d + d * o[1] + W, d + d * o[2]: d + d * o[2] + D]
    ssd = []
    for a, b in zip(OFF1, OFF2):
        q = (sh(a) - sh(b)) ** 2
        ssd.append(F.avg_pool3d(F.pad(q[None, None], (r,) * 6, mode="replicate"), 2 * r + 1, stride=1)[0, 0])
    ssd = torch.stack(ssd)
    m = ssd - ssd.min(0)[0]
    var = m.mean(0)
    mu = var.mean().item()
    lo, hi = mu * 0.001, mu * 1000
    allzero = (m == 0).all(0)
    low = (var < lo) & ~allzero
    high = var > hi
    bad = low | high
    nb = F.max_pool3d(bad[None, None].float(), 6, stride=6, ceil_mode=True).sum().item()
    tot = ((H + 5) // 6) * ((W + 5) // 6) * ((D + 5) // 6)
    print("%-28s V %9d  mu %.4g  allzero %.3f  low-clamped %d (%.4f)  high-clamped %d  blocks %d / %d" %
          (name, var.numel(), mu, allzero.float().mean().item(), int(low.sum()), low.float().mean().item(), int(high.sum()), int(nb), tot))

if __name__ == "__main__":
    shape = (160, 192, 224)
    for tag, f in (("c1 deformed_pair idx0", lambda: ph.deformed_pair(shape, 0, 4.0)), ("c4 deformed_pair idx2", lambda: ph.deformed_pair(shape, 2, 6.0)),
                   ("c5 zero_background", lambda: ph.zero_background_pair(shape, 0, 4.0))):
        a, b = f()
        census(a, tag + " fixed"); census(b, tag + " moving")
