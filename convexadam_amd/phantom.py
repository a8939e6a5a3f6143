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
