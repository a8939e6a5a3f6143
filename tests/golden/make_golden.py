"""Regenerates tests/golden/*.npz by importing and running the upstream reference (CPU, float32).

Run ONLY in the build container (needs /root/reference):   python tests/golden/make_golden.py
The fixtures are data: seeded inputs (or the seeds to rebuild them) plus the reference's outputs.
No reference source text is stored.  torch 2.10.0 CPU (AVX512, MKL VML) produced the committed files.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import import_reference  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from convexadam_amd.phantom import phantom, smooth_warp  # noqa: E402  (shared synthetic-data recipe)

U, M = import_reference()
CPU = torch.device("cpu")


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def main():
    torch.manual_seed(1234)
    torch.set_num_threads(8)

    # ---- A: MINDSSC (convex_adam_utils.py:24-68) -------------------------------------------------
    img = phantom((20, 18, 23), 3, 30)            # V = 8280, V mod 32 = 24 -> exercises the tail rule
    save("mind", img=img.numpy(),
         mind_r1d2=U.MINDSSC(img[None, None], 1, 2, device="cpu")[0].numpy(),
         mind_r2d2=U.MINDSSC(img[None, None], 2, 2, device="cpu")[0].numpy(),
         mind_r1d1=U.MINDSSC(img[None, None], 1, 1, device="cpu")[0].numpy())

    # ---- B: pooling + correlate + coupled_convex + IC + resize on one small pair -----------------
    H, W, D, gs, hw = 36, 30, 42, 3, 2
    fix = phantom((H, W, D), 2, 20)
    mov = torch.roll(phantom((H, W, D), 2, 21), (2, -1, 3), (0, 1, 2))
    ff = U.MINDSSC(fix[None, None], 1, 2, device="cpu")
    fm = U.MINDSSC(mov[None, None], 1, 2, device="cpu")
    fs = F.avg_pool3d(ff, gs, stride=gs)
    ms = F.avg_pool3d(fm, gs, stride=gs)
    ssd, am = U.correlate(fs, ms, hw, gs, (H, W, D), 12)
    n = 2 * hw + 1
    mesh = F.affine_grid(hw * torch.eye(3, 4).unsqueeze(0), (1, 1, n, n, n), align_corners=True).permute(0, 4, 1, 2, 3).reshape(3, -1, 1)
    soft = U.coupled_convex(ssd, am, mesh, gs, (H, W, D))
    ssd_, am_ = U.correlate(ms, fs, hw, gs, (H, W, D), 12)
    soft_ = U.coupled_convex(ssd_, am_, mesh, gs, (H, W, D))
    h, w, d = H // gs, W // gs, D // gs
    scale = torch.tensor([h - 1, w - 1, d - 1]).view(1, 3, 1, 1, 1).float() / 2
    in1, in2 = (soft / scale).flip(1), (soft_ / scale).flip(1)
    i1, i2 = U.inverse_consistency(in1, in2, iter=15)
    hr = F.interpolate(i1.flip(1) * scale * gs, size=(H, W, D), mode="trilinear", align_corners=False)
    lr = F.interpolate(hr, size=(H // 2, W // 2, D // 2), mode="trilinear", align_corners=False)
    save("convex", shape=np.array([H, W, D, gs, hw]), feat_fix=fs[0].numpy(), feat_mov=ms[0].numpy(), ssd=ssd.numpy(), argmin=am.numpy(),
         mesh=mesh[:, :, 0].numpy(), soft=soft[0].numpy(), ssd_rev_sum=np.float64(ssd_.double().sum().item()),
         argmin_rev=am_.numpy(), soft_rev=soft_[0].numpy(), ic_in1=in1[0].numpy(), ic_in2=in2[0].numpy(),
         ic_out1=i1[0].numpy(), ic_out2=i2[0].numpy(), disp_hr=hr[0].numpy(), disp_lr=lr[0].numpy())

    # avg_pool3d(g, stride=g) with remainders (convex_adam_MIND.py:118-119,149-150)
    x = torch.randn(1, 5, 13, 14, 15)
    save("pool", x=x[0].numpy(), g2=F.avg_pool3d(x, 2, stride=2)[0].numpy(), g3=F.avg_pool3d(x, 3, stride=3)[0].numpy(),
         g6=F.avg_pool3d(x, 6, stride=6)[0].numpy(), box3=F.avg_pool3d(x, 3, stride=1, padding=1)[0].numpy(),
         box5=F.avg_pool3d(x, 5, stride=1, padding=2)[0].numpy())

    # multi-channel correlate with C >= 16 (cascade sum) and ragged inner size (tail rule)
    f20 = torch.rand(1, 20, 7, 5, 9)
    m20 = torch.rand(1, 20, 7, 5, 9)
    s20, a20 = U.correlate(f20, m20, 1, 1, (7, 5, 9), 20)
    save("correlate_c20", fix=f20[0].numpy(), mov=m20[0].numpy(), ssd=s20.numpy(), argmin=a20.numpy())

    # ---- C: Adam instance optimisation (convex_adam_MIND.py:147-182): the reference's own loop runs inside convex_adam_pt; its inputs,
    #         per-iteration control grids and gradients are observed through wrappers around F.avg_pool3d and torch.optim.Adam.step
    g, lam = 2, 1.25
    rec = dict(pooled=[], P={}, G={})
    _pool, _step = F.avg_pool3d, torch.optim.Adam.step

    def pool(x, *a, **k):
        y = _pool(x, *a, **k)
        if x.shape[1] == 12 and (a[0] if a else k.get("kernel_size")) == g and k.get("stride") == g:
            rec["pooled"].append(y.detach()[0].clone())            # patch_features_fix, then patch_features_mov (:149-150)
        return y

    def step(opt, *a, **k):
        n = len(rec["P"]) + 1
        w = opt.param_groups[0]["params"][0]
        rec["P"][n] = w.detach()[0].clone()                        # control grid of forward pass n (before its update)
        rec["G"][n] = w.grad.detach()[0].clone()
        r = _step(opt, *a, **k)
        rec["after"] = w.detach()[0].clone()
        return r

    F.avg_pool3d, torch.optim.Adam.step = pool, step
    try:
        M.convex_adam_pt(fix, mov, mind_r=1, mind_d=2, lambda_weight=lam, grid_sp=gs, disp_hw=hw, selected_niter=20, grid_sp_adam=g, ic=True,
                         dtype=torch.float32, device=CPU)
    finally:
        F.avg_pool3d, torch.optim.Adam.step = _pool, _step
    pf, pm = rec["pooled"][-2][None], rec["pooled"][-1][None]
    P0 = rec["P"][1].numpy().copy()

    def disp_sample(P):                                            # (:166) three zero-padded 3^3 mean filters
        u = P[None]
        for _ in range(3):
            u = F.avg_pool3d(u, 3, stride=1, padding=1)
        return u[0].numpy().copy()

    out = {}
    for niter in (1, 2, 5, 20):
        out["U_%d" % niter] = disp_sample(rec["P"][niter])
        out["G_%d" % niter] = rec["G"][niter].numpy().copy()
        out["P_%d" % niter] = (rec["P"][niter + 1] if niter + 1 in rec["P"] else rec["after"]).numpy().copy()
    save("adam", F2=pf[0].numpy(), M2=pm[0].numpy(), P0=P0, lam=np.float32(lam), **out)

    # ---- D: whole pipeline convex_adam_pt (convex_adam_MIND.py:64-202) ---------------------------
    H, W, D = 32, 28, 36
    fix = phantom((H, W, D), 1, 10)
    u = smooth_warp((H, W, D), 5, amp=2.0)
    mov = F.grid_sample(phantom((H, W, D), 1, 11)[None, None], u, mode="bilinear", padding_mode="border", align_corners=False)[0, 0]
    pipe = dict(fix=fix.numpy(), mov=mov.numpy())
    kw = dict(mind_r=1, mind_d=2, grid_sp=4, disp_hw=3, grid_sp_adam=2, dtype=torch.float32, device=CPU)
    pipe["convex_only_ic"] = M.convex_adam_pt(fix, mov, lambda_weight=0, ic=True, **kw).astype(np.float32)
    pipe["convex_only_noic"] = M.convex_adam_pt(fix, mov, lambda_weight=0, ic=False, **kw).astype(np.float32)
    for niter in (1, 5, 20):
        pipe["adam_%d" % niter] = M.convex_adam_pt(fix, mov, lambda_weight=1.25, selected_niter=niter, ic=True, **kw).astype(np.float32)
    pipe["adam_5_smooth3"] = M.convex_adam_pt(fix, mov, lambda_weight=1.25, selected_niter=5, selected_smooth=3, ic=True, **kw).astype(np.float32)
    pipe["adam_5_noic"] = M.convex_adam_pt(fix, mov, lambda_weight=1.25, selected_niter=5, ic=False, **kw).astype(np.float32)
    save("pipeline", **pipe)

    # ---- E: known answers of SURVEY appendix A (64^3 translated pair, convex only) ---------------
    fix = phantom((64, 64, 64), 2, 20)
    ka = {}
    for name, sh, gs_ in (("roll_4_0_m8_gs4", (4, 0, -8), 4), ("roll_6_m6_0_gs6", (6, -6, 0), 6)):
        mov = torch.roll(fix, sh, (0, 1, 2))
        dsp = M.convex_adam_pt(fix, mov, lambda_weight=0, grid_sp=gs_, disp_hw=4, dtype=torch.float32, device=CPU)
        ka[name] = dsp[16:48, 16:48, 16:48].mean((0, 1, 2))
        ka[name + "_sub"] = dsp[::4, ::4, ::4].astype(np.float32)
    save("translation64", **ka)

    # ---- F: nnUNet label features (convex_adam_nnUNet.py:19-38), CUDA/half calls neutralised -----
    import importlib
    nib = sys.modules["nibabel"]
    _cuda, _half = torch.Tensor.cuda, torch.Tensor.half
    torch.Tensor.cuda = lambda s, *a, **k: s
    torch.Tensor.half = lambda s, *a, **k: s.float()
    try:
        N = importlib.import_module("convexAdam.convex_adam_nnUNet")
        gl = torch.Generator().manual_seed(7)
        lab = torch.argmax(F.interpolate(torch.randn(1, 9, 5, 5, 5, generator=gl), size=(24, 20, 28), mode="trilinear"), 1)[0].float()
        lab[lab == 4] = 6   # leave a gap in the label set
        labm = torch.roll(lab, (2, -1, 1), (0, 1, 2))
        labm[:3] = 10       # label only present in moving
        lab[0, 0, 0] = 11   # the reference needs equal max labels in both maps (bincount / one_hot sizes)
        labm[-1, -1, -1] = 11
        f_fix, f_mov = N.extract_features(lab, labm)
        save("labels", lab_fix=lab.numpy().astype(np.int16), lab_mov=labm.numpy().astype(np.int16),
             feat_fix_sum=f_fix[0].double().sum((1, 2, 3)).numpy(), feat_mov_sum=f_mov[0].double().sum((1, 2, 3)).numpy(),
             weights=f_fix[0].amax((1, 2, 3)).numpy(), feat_fix_pool2=F.avg_pool3d(f_fix, 2, stride=2)[0].numpy())
    finally:
        torch.Tensor.cuda, torch.Tensor.half = _cuda, _half


if __name__ == "__main__" and "--smoothers" not in sys.argv and "--metrics" not in sys.argv and "--hd95" not in sys.argv:
    main()
    smoother_goldens_later = True


def smoother_goldens():
    """Row P: GaussianSmoothing / kovesi_spline of self_configuring/convexAdam_hyper_util.py inside the sweep's Adam loop
    (adam_run_withconfig_shiftSpline.py:214-230).  cupy is stubbed (only the metrics of that module need it)."""
    import types
    for m in ("cupy", "cupyx", "cupyx.scipy", "cupyx.scipy.ndimage"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, os.path.join(os.environ.get("CONVEXADAM_REFERENCE", "/root/reference"), "self_configuring"))
    import convexAdam_hyper_util as HU
    torch.manual_seed(77)
    x = torch.randn(1, 3, 11, 12, 13)
    go = torch.randn(1, 3, 11, 12, 13)
    out = dict(x=x[0].numpy(), go=go[0].numpy())
    specs = {"gauss07": HU.GaussianSmoothing(0.7), "gauss10": HU.GaussianSmoothing(1.0), "kov16": HU.kovesi_spline(1.6, 4),
             "kov19": HU.kovesi_spline(1.9, 4), "kov28": HU.kovesi_spline(2.8, 4)}
    for k, sm in specs.items():
        xr = x.clone().requires_grad_(True)
        y = sm(xr)
        y.backward(go)
        out[k + "_fwd"] = y.detach()[0].numpy()
        out[k + "_bwd"] = xr.grad[0].numpy()
    out["gauss07_w"] = specs["gauss07"].weight.numpy()
    out["gauss10_w"] = specs["gauss10"].weight.numpy()
    # a few Adam iterations with two of the smoothers, n_ch cost scale (adam_run_withconfig_shiftSpline.py:227)
    a = dict(np.load(os.path.join(HERE, "adam.npz")))
    pf, pm = torch.from_numpy(a["F2"])[None], torch.from_numpy(a["M2"])[None]
    h, w, d = pf.shape[2:]
    lam = 0.8
    for k in ("gauss07", "kov19"):
        net = nn.Sequential(nn.Conv3d(3, 1, (h, w, d), bias=False))
        net[0].weight.data[:] = torch.from_numpy(a["P0"])[None]
        opt = torch.optim.Adam(net.parameters(), lr=1)
        grid0 = F.affine_grid(torch.eye(3, 4).unsqueeze(0), (1, 1, h, w, d), align_corners=False)
        for it in range(3):
            opt.zero_grad()
            ds = specs[k](net[0].weight).permute(0, 2, 3, 4, 1)
            reg = lam * ((ds[0, :, 1:, :] - ds[0, :, :-1, :]) ** 2).mean() + lam * ((ds[0, 1:, :, :] - ds[0, :-1, :, :]) ** 2).mean() + lam * ((ds[0, :, :, 1:] - ds[0, :, :, :-1]) ** 2).mean()
            sc = torch.tensor([(h - 1) / 2, (w - 1) / 2, (d - 1) / 2]).unsqueeze(0)
            gd = grid0.view(-1, 3).float() + ((ds.view(-1, 3)) / sc).flip(1).float()
            pms = F.grid_sample(pm.float(), gd.view(1, h, w, d, 3), align_corners=False, mode="bilinear")
            loss = ((pms - pf).pow(2).mean(1) * 12).mean()
            (loss + reg).backward()
            if it == 0:
                out[k + "_adam_U1"] = ds.detach().permute(0, 4, 1, 2, 3)[0].numpy().copy()
                out[k + "_adam_G1"] = net[0].weight.grad[0].numpy().copy()
            opt.step()
        out[k + "_adam_U3"] = ds.detach().permute(0, 4, 1, 2, 3)[0].numpy().copy()
    save("smoothers", **out)


def metrics_goldens():
    """evaluation operators of the sweep scripts (convex_run_withconfig.py:136-150,
    convex_run_paired_mind.py:167-173) and apply_convex (apply_convex.py:13-24), captured from the reference on CPU."""
    import types
    for m in ("cupy", "cupyx", "cupyx.scipy", "cupyx.scipy.ndimage"):
        sys.modules.setdefault(m, types.ModuleType(m))
    from _ref_import import import_reference
    import_reference()
    import importlib
    AC = importlib.import_module("convexAdam.apply_convex")
    sys.path.insert(0, os.path.join(os.environ.get("CONVEXADAM_REFERENCE", "/root/reference"), "self_configuring"))
    import convexAdam_hyper_util as HU
    g = torch.Generator().manual_seed(123)
    H, W, D = 22, 26, 30
    # smooth displacement field in voxels with a few folds, channel order (H, W, D) like disp_hr
    disp = F.interpolate(torch.randn(1, 3, 4, 5, 6, generator=g) * 3.0, size=(H, W, D), mode="trilinear", align_corners=False)
    out = dict(disp=disp[0].numpy())
    out["jac_vox"] = HU.jacobian_determinant_3d(disp, False).numpy()                       # convex_run_withconfig.py:137
    norm = disp / (torch.tensor([H - 1, W - 1, D - 1]) / 2).view(1, 3, 1, 1, 1)
    out["disp_norm"] = norm[0].numpy()
    out["jac_norm"] = HU.jacobian_determinant_3d(norm, True).numpy()
    jd = torch.from_numpy(out["jac_vox"])
    jl = jd.add(3).clamp_(0.000000001, 1000000000).log()                                   # :148
    out["jac_log_std"] = np.float32(jl.std().item())
    out["jac_neg_frac"] = np.float32((jd < 0).float().mean().item())
    # label maps, nearest-neighbour warp, Dice                                              (:96, :141-142)
    lab = torch.randn(1, 7, 5, 6, 7, generator=g)
    seg_m = F.interpolate(lab, size=(H, W, D), mode="trilinear", align_corners=False).argmax(1)[0].float()
    seg_f = torch.roll(seg_m, (1, -2, 1), (0, 1, 2))
    grid0 = F.affine_grid(torch.eye(3, 4).unsqueeze(0), (1, 1, H, W, D), align_corners=False)
    scale1 = torch.tensor([D - 1, W - 1, H - 1]) / 2
    seg_w = F.grid_sample(seg_m.view(1, 1, H, W, D), grid0 + disp.permute(0, 2, 3, 4, 1).flip(-1).div(scale1), mode="nearest").squeeze()
    out.update(seg_moving=seg_m.numpy(), seg_fixed=seg_f.numpy(), seg_warped=seg_w.numpy())
    out["dice"] = HU.dice_coeff(seg_f, seg_w, 7).numpy()
    # key-point TRE                                                                          (convex_run_paired_mind.py:167-173)
    key_f = torch.rand(40, 3, generator=g) * torch.tensor([H - 1.0, W - 1.0, D - 1.0])
    key_f[:4] = torch.tensor([[0.0, 0.0, 0.0], [H - 1.0, W - 1.0, D - 1.0], [0.5, 25.0, 29.0], [21.0, 0.0, 14.5]])
    key_m = key_f + torch.randn(40, 3, generator=g)
    lms = (key_f.flip(1) / scale1 - 1).view(1, -1, 1, 1, 3)
    samp = F.grid_sample(disp.float(), lms).squeeze().t()
    out.update(key_fixed=key_f.numpy(), key_moving=key_m.numpy(), disp_sampled=samp.numpy(),
               tre=(key_f - key_m + samp).square().sum(-1).sqrt().numpy())
    # rank aggregation helper                                                                (hyper_util:28-31)
    vals = torch.rand(17, generator=g)
    out.update(rank_in=vals.numpy(), rank_out=HU.sort_rank(vals).numpy())
    # apply_convex: scipy map_coordinates, order 1                                           (apply_convex.py:13-24)
    moving = torch.rand(H, W, D, generator=g).numpy().astype(np.float32)
    dd = disp[0].permute(1, 2, 3, 0).numpy().astype(np.float64)                              # (H, W, D, 3) float64 like convex_adam_pt's result
    out.update(moving=moving, warped=AC.apply_convex(dd, moving))
    save("metrics", **out)


def hd95_goldens():
    """cupy_hd95 (self_configuring/convexAdam_hyper_util.py:32-51) run from the reference module itself.  cupy and
    cupyx are absent from this image (and the module's own cupyx import is commented out, :23), so the two names the function
    needs are supplied with their numpy/scipy equivalents: cupy.asarray/zeros -> numpy, distance_transform_edt(x,
    float64_distances=False) -> scipy's EDT cast to float32.  The goldens therefore pin the function's logic (masks, surfaces,
    the two percentiles, the 30 default), not cupy's rounding."""
    import types
    from scipy.ndimage import distance_transform_edt as sedt
    cp = types.ModuleType("cupy")
    cp.asarray = lambda x: np.asarray(x)
    cp.zeros = lambda n: np.zeros(n)
    sys.modules["cupy"] = cp
    for m in ("cupyx", "cupyx.scipy", "cupyx.scipy.ndimage"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, os.path.join(os.environ.get("CONVEXADAM_REFERENCE", "/root/reference"), "self_configuring"))
    sys.modules.pop("convexAdam_hyper_util", None)
    import convexAdam_hyper_util as HU
    HU.cupy = cp
    HU.distance_transform_edt = lambda x, float64_distances=False: sedt(np.asarray(x)).astype(np.float32)
    g = torch.Generator().manual_seed(321)
    H, W, D = 20, 24, 28
    lab = torch.randn(1, 6, 4, 5, 6, generator=g)
    seg_f = F.interpolate(lab, size=(H, W, D), mode="trilinear", align_corners=False).argmax(1)[0]
    seg_m = torch.roll(seg_f, (2, -1, 3), (0, 1, 2)).clone()
    seg_m[seg_m == 4] = 0                                                     # label 4 absent from one map -> 30
    out = dict(seg_fixed=seg_f.numpy(), seg_moving=seg_m.numpy())
    out["hd95_p1"] = HU.cupy_hd95(seg_f.long(), seg_m.long(), 6).numpy()       # label 6 absent from both
    out["hd95_p2"] = HU.cupy_hd95(seg_f.long(), seg_m.long(), 6, precision=2).numpy()
    # non-integer precisions (round 4): F.interpolate(.., scale_factor=p) in nearest mode -- output extent floor(n * p), source index
    # min(floor(dst * float32(1 / p)), n - 1)
    out["hd95_p1_5"] = HU.cupy_hd95(seg_f.long(), seg_m.long(), 6, precision=1.5).numpy()
    out["hd95_p0_5"] = HU.cupy_hd95(seg_f.long(), seg_m.long(), 6, precision=0.5).numpy()
    out["hd95_p2_5"] = HU.cupy_hd95(seg_f.long(), seg_m.long(), 6, precision=2.5).numpy()
    save("hd95", **out)


if __name__ == "__main__":
    if "--hd95" in sys.argv:
        hd95_goldens()
    elif "--metrics" in sys.argv:
        metrics_goldens()
    else:
        smoother_goldens()
