"""Where does convex_adam_pt_many spend its time?  Host-side pieces timed alone on the GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from convexadam_amd.convex_adam_MIND import convex_adam_pt, convex_adam_pt_many, register_pair_device
from convexadam_amd.convex_adam_utils import validate_image
from convexadam_amd.phantom import deformed_pair
dev = torch.device("cuda:0")
SHAPE = (160, 192, 224)
fix, mov = deformed_pair(SHAPE, 0, 4.0)
CFG = dict(mind_r=1, mind_d=2, lambda_weight=1.25, grid_sp=6, disp_hw=6, selected_niter=80, selected_smooth=0, grid_sp_adam=2, ic=True, adam_mode="fast")
print("torch threads", torch.get_num_threads(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
def T(f, n=10):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
pin = torch.empty(SHAPE, dtype=torch.float32, pin_memory=True)
print("validate_image + float + contiguous: %.2f ms" % T(lambda: validate_image(fix).float().contiguous()))
print("np.copyto 27.5 MB into pinned (1 thread): %.2f ms" % T(lambda: np.copyto(pin.numpy(), fix.numpy())))
print("torch copy_ 27.5 MB into pinned: %.2f ms" % T(lambda: pin.copy_(fix)))
import threading
def par_copy(dst, src, k=4):
    d, s = dst.reshape(-1), src.reshape(-1); n = d.size; ts = []
    for i in range(k):
        t = threading.Thread(target=np.copyto, args=(d[i * n // k:(i + 1) * n // k], s[i * n // k:(i + 1) * n // k])); t.start(); ts.append(t)
    for t in ts: t.join()
for k in (2, 4, 8):
    print("np.copyto in %d python threads: %.2f ms" % (k, T(lambda: par_copy(pin.numpy(), fix.numpy(), k))))
fd, md = fix.to(dev), mov.to(dev)
torch.cuda.synchronize()
def reg():
    register_pair_device(fd, md, **CFG); torch.cuda.synchronize()
print("engine: %.2f ms" % T(reg, 5))
def up():
    pin.to(dev, non_blocking=True); torch.cuda.synchronize()
print("H2D 27.5 MB pinned: %.2f ms" % T(up))
def up2():
    fix.to(dev); torch.cuda.synchronize()
print("H2D 27.5 MB pageable: %.2f ms" % T(up2))
big = torch.empty(SHAPE + (3,), dtype=torch.float64, device=dev); pbig = torch.empty(SHAPE + (3,), dtype=torch.float64, pin_memory=True)
def down():
    pbig.copy_(big, non_blocking=True); torch.cuda.synchronize()
print("D2H 165 MB pinned: %.2f ms" % T(down))
for m in (8, 8, 16, 16, 32):
    t0 = time.perf_counter()
    for _ in convex_adam_pt_many([(fix, mov)] * m, device=dev, **CFG): pass
    print("convex_adam_pt_many, %d pairs: %.2f ms per pair" % (m, (time.perf_counter() - t0) / m * 1e3))
print("convex_adam_pt: %.2f ms" % T(lambda: convex_adam_pt(fix, mov, device=dev, **CFG), 8))
for rep in range(2):
    ts = [time.perf_counter()]
    for _ in convex_adam_pt_many([(fix, mov)] * 10, device=dev, **CFG):
        ts.append(time.perf_counter())
    print("yield gaps (ms):", " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(ts, ts[1:])))
