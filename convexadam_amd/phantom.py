"""Synthetic volumes for tests and benchmarks.

The reference ships no data that can be used here (its only test subject is a missing large blob,
tests/input/10000/*_t2w.mha), so tests and bench.py use a seeded multi-octave phantom: single-scale
noise finer than `grid_sp` makes the convex stage return ~0, a sum of three octaves does not.
Pure torch-CPU generators so the same seeds give the same volumes in the build container and on the
GPU box.
"""
import torch
import torch.nn.functional as F


def phantom(shape, seed, noise_seed, noise=0.02):
    """(H,W,D) float32 volume: sum of trilinearly up-sampled Gaussian octaves + white noise."""
    g = torch.Generator().manual_seed(seed)
    x = 0
    for div, amp in ((16, 1.0), (8, 0.5), (4, 0.25)):
        x = x + amp * F.interpolate(torch.randn(1, 1, *[max(2, s // div) for s in shape], generator=g),
                                    size=tuple(shape), mode="trilinear", align_corners=False)
    x = x + noise * torch.randn(1, 1, *shape, generator=torch.Generator().manual_seed(noise_seed))
    return x[0, 0].contiguous()


def smooth_warp(shape, seed, amp=4.0):
    """Normalised sampling grid (1,H,W,D,3) for F.grid_sample: identity + smooth random displacement
    of about `amp` voxels (pull-back warp used to make a deformed 'moving' image)."""
    H, W, D = shape
    g = torch.Generator().manual_seed(seed)
    u = amp * F.interpolate(torch.randn(1, 3, 3, 3, 4, generator=g), size=(H, W, D), mode="trilinear", align_corners=False)
    ident = F.affine_grid(torch.eye(3, 4).unsqueeze(0), (1, 1, H, W, D), align_corners=False)
    scale = torch.tensor([2.0 / D, 2.0 / W, 2.0 / H]).view(1, 1, 1, 1, 3)
    return ident + u.permute(0, 2, 3, 4, 1).flip(-1) * scale


def label_phantom(shape, n_labels, seed):
    """(H,W,D) integer label map with `n_labels` blobs."""
    g = torch.Generator().manual_seed(seed)
    z = F.interpolate(torch.randn(1, n_labels, 8, 8, 8, generator=g), size=tuple(shape), mode="trilinear", align_corners=False)
    return torch.argmax(z, 1)[0].float().contiguous()


def deformed_pair(shape, idx=0, amp=4.0):
    """Synthetic pair of SURVEY 8(d) config 2 (and bench.py): fixed = multi-octave phantom, moving = another noise
    realisation of the same phantom pulled back through a smooth random warp of about `amp` voxels.  CPU tensors."""
    fix = phantom(shape, 1 + idx, 10 + idx)
    grid = smooth_warp(shape, 5 + idx, amp=amp)
    mov = F.grid_sample(phantom(shape, 1 + idx, 110 + idx)[None, None], grid, mode="bilinear", padding_mode="border",
                        align_corners=False)[0, 0]
    return fix.contiguous(), mov.contiguous()


def ellipsoid_mask(shape, semi=0.35, shift=(0, 0, 0)):
    """(H,W,D) float32 0/1 mask: ellipsoid with semi-axes `semi` x extent around the (shifted) volume centre
    (the lung-mask stand-in)."""
    ax = [(torch.arange(s, dtype=torch.float32) - (s - 1) / 2.0 - sh) / (semi * s) for s, sh in zip(shape, shift)]
    r2 = ax[0].view(-1, 1, 1) ** 2 + ax[1].view(1, -1, 1) ** 2 + ax[2].view(1, 1, -1) ** 2
    return (r2 <= 1.0).float().contiguous()


def zero_background_pair(shape, idx=0, amp=4.0, semi=0.42):
    """deformed_pair with an EXACT-zero background, the way skull-stripped brain MRI is: both images are multiplied by 0/1 ellipsoid
    masks (the moving one shifted by a few voxels), so every voxel outside is bit-zero and MIND's variance clamp / the flat cost
    columns of the search are exercised at full size."""
    fix, mov = deformed_pair(shape, idx, amp)
    return (fix * ellipsoid_mask(shape, semi)).contiguous(), (mov * ellipsoid_mask(shape, semi, shift=(2, -1, 3))).contiguous()


def warped_label_pair(shape, n_labels=18, seed=11, amp=0.05):
    """(fixed, moving) float32 label maps with exactly `n_labels` labels 0 .. n_labels-1 present in both: argmax of smooth random
    fields; the moving map is the fixed one pulled through a smooth random warp (nearest neighbour), `amp` in normalised units
    (0.05 ~ 4-5 voxels at 160-224 voxels per axis).  The largest label is planted in a corner of both maps: the reference's nnUNet
    feature extraction (convex_adam_nnUNet.py:19-38) needs equal maxima (bincount / one_hot sizes)."""
    g = torch.Generator().manual_seed(seed)
    f = F.interpolate(torch.randn(1, n_labels, 6, 5, 7, generator=g), size=tuple(shape), mode="trilinear", align_corners=False)
    lab = torch.argmax(f, 1)[0].float()
    base = F.affine_grid(torch.eye(3, 4)[None], (1, 1) + tuple(shape), align_corners=False)
    warp = F.interpolate(torch.randn(1, 3, 4, 4, 4, generator=g) * amp, size=tuple(shape), mode="trilinear", align_corners=False)
    labm = F.grid_sample(lab[None, None], base + warp.permute(0, 2, 3, 4, 1), mode="nearest", padding_mode="border", align_corners=False)[0, 0]
    lab[0, 0, 0] = float(n_labels - 1)
    labm[-1, -1, -1] = float(n_labels - 1)
    return lab.contiguous(), labm.contiguous()
