"""GPU tests of the surface-only HD95 path (csrc/surfdist.hip): bit planes of a label map, exact squared distances at the surface voxels,
and cupy_hd95(method="surface") == method="edt" == the numpy/scipy restatement of hyper_util.py:32-51 (oracle/metrics_oracle.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def HU():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from convexadam_amd import convexAdam_hyper_util
    return convexAdam_hyper_util


@pytest.fixture(scope="module")
def morc():
    from oracle import metrics_oracle
    return metrics_oracle


def blobs(shape, nl, seed, shift=(0, 0, 0)):
    """Label map of nl smooth blobs (arg-max of low-pass noise), integers 0 .. nl."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    f = np.stack([gaussian_filter(rng.standard_normal(shape), 2.0 + 0.3 * i) for i in range(nl + 1)])
    f[0] *= 1.5
    return np.roll(f.argmax(0), shift, (0, 1, 2)).astype(np.int64)


def surface_hist_numpy(seg_b, seg_a, nl, active, nbins):
    """dist_a[surf_b] per label, squared, as histograms: the definition of hyper_util.py:39-48 with scipy's transform."""
    from scipy.ndimage import distance_transform_edt as edt
    hist = np.zeros((nl, nbins), np.int64)
    over = np.zeros(nl, np.int32)
    for lab in active:
        mb = seg_b == lab
        if not mb.any():
            continue
        surf = np.rint(edt(mb) ** 2) == 1 if not mb.all() else np.zeros_like(mb)
        ma = seg_a == lab
        if ma.all() or not ma.any():                 # one of the two transforms has no zero voxel
            inside = ma[surf]
            need_missing = inside if ma.all() else ~inside
            if need_missing.any():
                over[lab - 1] = 1
            continue
        d2 = np.rint(edt(ma).astype(np.float64) ** 2 + edt(~ma).astype(np.float64) ** 2).astype(np.int64)
        np.add.at(hist[lab - 1], d2[surf], 1)
    return hist, over


KERNELS = ("voxels", "bits")            # cvx_surface_distance_hist_i64 (one lane per voxel of map b) / _bits_i64 (word arithmetic on both maps' planes, round 5)


def run_surface_hist(seg_b, seg_a, nl, active, max_radius=0, kernel="voxels"):
    from convexadam_amd._lib import check, lib, ptr, stream_ptr
    L = lib()
    H, W, D = seg_a.shape
    nbins = (H - 1) ** 2 + (W - 1) ** 2 + (D - 1) ** 2 + 2
    a, b = dev(seg_a.astype(np.float32)), dev(seg_b.astype(np.float32))
    bits = torch.empty(int(L.cvx_label_bits_bytes(H, W, D, nl)) // 8, dtype=torch.int64, device=DEV)
    sp = stream_ptr(torch.device(DEV))
    check(L.cvx_label_bits_u64(ptr(a), H, W, D, nl, ptr(bits), sp))
    hist = torch.zeros((nl, nbins), dtype=torch.int64, device=DEV)
    over = torch.zeros(nl, dtype=torch.int32, device=DEV)
    act = [0, 0, 0, 0]
    for lab in active:
        act[lab >> 6] |= 1 << (lab & 63)
    act4 = (C.c_uint64 * 4)(*act)
    if kernel == "bits":
        bits_b = torch.empty_like(bits)
        check(L.cvx_label_bits_u64(ptr(b), H, W, D, nl, ptr(bits_b), sp))
        nws = int(L.cvx_surface_distance_hist_bits_workspace_bytes(H, W, D, nl))
        ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
        check(L.cvx_surface_distance_hist_bits_i64(ptr(bits_b), ptr(bits), H, W, D, nl, C.cast(act4, C.c_void_p), nbins, ptr(hist), nbins, ptr(over), 1,
                                                   max_radius, ptr(ws), nws, sp))
    else:
        check(L.cvx_surface_distance_hist_i64(ptr(b), ptr(bits), H, W, D, nl, C.cast(act4, C.c_void_p), nbins, ptr(hist), nbins, ptr(over), 1, max_radius, sp))
    return host(bits), host(hist), host(over), nbins


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("shape", [(9, 7, 12), (12, 10, 64), (6, 9, 65), (5, 4, 200), (30, 26, 70), (3, 40, 129), (4, 5, 330)])
def test_label_bits_and_surface_distances_vs_scipy(shape, kernel):
    """cvx_label_bits_u64 == numpy's packed masks, and cvx_surface_distance_hist_i64 == the histogram of (edt(a==l) + edt(a!=l))**2 over
    the voxels of b whose inside distance is 1: rows of less / exactly / more than one 64-voxel word, labels touching the border, a
    label missing from one map (inactive: skipped), one-voxel labels, non-integer and out-of-range values (no label)."""
    H, W, D = shape
    nl = 5
    a = blobs(shape, nl, sum(shape))
    b = blobs(shape, nl, sum(shape), shift=(1, -1, 2))
    b[b == 4] = 0                                           # label 4 only in a
    a[0, 0, 0], b[-1, -1, -1] = 3, 3                        # corner voxels
    a[H // 2, W // 2, D // 2] = 5                           # make sure label 5 exists in both (possibly as single voxels)
    b[H // 2, W // 2, min(D - 1, D // 2 + 1)] = 5
    active = [lab for lab in range(1, nl + 1) if (a == lab).any() and (b == lab).any()]
    assert 4 not in active and len(active) >= 3
    bits, hist, over, nbins = run_surface_hist(b, a, nl, active, kernel=kernel)
    nseg = (D + 63) // 64
    want_bits = np.zeros((nl, H * W, nseg * 64), np.uint8)
    for lab in range(1, nl + 1):
        want_bits[lab - 1, :, :D] = (a == lab).reshape(H * W, D)
    packed = np.packbits(want_bits.reshape(nl, H * W, nseg, 64), axis=-1, bitorder="little").view(np.uint64).reshape(-1)
    assert np.array_equal(bits.view(np.uint64), packed)
    want_hist, want_over = surface_hist_numpy(b, a, nl, active, nbins)
    assert np.array_equal(over, want_over) and not over.any()
    assert np.array_equal(hist, want_hist)
    assert hist.sum() > 0
    # values that are no label: fractional and out of range (bit planes and surface test both ignore them)
    af, bf = a.astype(np.float32), b.astype(np.float32)
    af[1, 1, 1], bf[2, 2, 2] = 1.5, 77.0
    bits2, hist2, over2, _ = run_surface_hist(bf, af, nl, active, kernel=kernel)
    a2, b2 = a.copy(), b.copy()
    a2[1, 1, 1], b2[2, 2, 2] = 0, 0
    # the voxel of b holding 77 differs from every neighbour (it makes ITS neighbours surface voxels, like any other foreign value)
    want2, _ = surface_hist_numpy(np.where(bf == 77.0, -1, b2), a2, nl, active, nbins)
    assert np.array_equal(hist2, want2)


@pytest.mark.parametrize("kernel", KERNELS)
def test_surface_distance_far_targets_and_missing_targets(kernel):
    """The ring search runs to the far corner when the only voxel of the label in map a is there (exact at any distance), and reports
    overflow when map a holds no voxel of the wanted kind (the label fills all of a: its outside transform has no zero voxel)."""
    shape = (20, 33, 70)
    a = np.zeros(shape, np.int64)
    a[19, 32, 69] = 1
    b = np.zeros(shape, np.int64)
    b[0:2, 0:2, 0:3] = 1
    bits, hist, over, nbins = run_surface_hist(b, a, 2, [1], kernel=kernel)
    want, wover = surface_hist_numpy(b, a, 2, [1], nbins)
    assert np.array_equal(hist, want) and not over.any() and hist[0].sum() > 0
    assert hist[0, 19 ** 2 + 32 ** 2 + 67 ** 2] == 1                                    # from the surface voxel (0, 0, 2)
    # a bounded search gives up on those voxels (flag 2: the caller switches to the transforms) but still counts what it reaches
    bits, hist, over, nbins = run_surface_hist(b, a, 2, [1], max_radius=5, kernel=kernel)
    assert over[0] == 2 and hist.sum() == 0
    a[1, 3, 0] = 1
    bits, hist, over, nbins = run_surface_hist(b, a, 2, [1], max_radius=5, kernel=kernel)
    assert over[0] == 0 and np.array_equal(hist, surface_hist_numpy(b, a, 2, [1], nbins)[0])
    a[:] = 1
    bits, hist, over, nbins = run_surface_hist(b, a, 2, [1], kernel=kernel)
    assert over[0] == 1 and over[1] == 0 and hist.sum() == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_surface_distance_kernels_agree_on_mid_range_distances_and_many_labels(seed):
    """The two kernels count the same squared distances where the word arithmetic's stages hand over to each other: a label map against
    itself shifted by 3 .. 9 voxels (squared distances around 8, 24 and 63: the table's levels, the early hand-over of sparse wavefronts, the
    ring search), 70 labels (labels above 64 count through global atomics), rows of two words with a ragged tail."""
    rng = np.random.default_rng(seed)
    shape, nl = (26, 30, 100), 70
    base = blobs(shape, 12, 100 + seed)
    a = np.where(base > 0, base * 5 + 8, 0)                 # labels 13, 18, .. 68: some beyond the 64 that count in LDS
    shift = tuple(int(v) for v in rng.integers(2, 7, 3))
    b = np.roll(a, shift, (0, 1, 2))
    active = [lab for lab in range(1, nl + 1) if (a == lab).any() and (b == lab).any()]
    assert len(active) >= 6 and max(active) > 64
    r0 = run_surface_hist(b, a, nl, active, kernel="voxels")
    r1 = run_surface_hist(b, a, nl, active, kernel="bits")
    assert np.array_equal(r0[1], r1[1]) and np.array_equal(r0[2], r1[2]) and r1[1].sum() > 0
    assert r1[1][:, 9:64].sum() > 0 and r1[1][:, 64:].sum() > 0, "the example must reach the table's upper levels and the ring search"
    want, _ = surface_hist_numpy(b, a, nl, active, r1[3])
    assert np.array_equal(r1[1], want)


@pytest.mark.parametrize("shape,nl", [((24, 30, 40), 6), ((40, 36, 70), 13), ((17, 21, 130), 3)])
def test_hd95_surface_method_equals_edt_method_and_oracle(HU, morc, shape, nl):
    """cupy_hd95: method="surface" (default at precision 1) == method="edt" == the scipy restatement, bit for bit, with and without the
    per-fixed-map cache; labels missing from one map score 30."""
    a = blobs(shape, nl, 5 + nl)
    for shift in ((1, -1, 2), (0, 3, -4)):
        b = blobs(shape, nl, 5 + nl, shift=shift)
        b[b == 2] = 0
        fa, fb = dev(a.astype(np.float32)), dev(b.astype(np.float32))
        want = morc.hd95(a, b, nl, 1)
        s = host(HU.cupy_hd95(fa, fb, nl))
        e = host(HU.cupy_hd95(fa, fb, nl, method="edt"))
        assert np.array_equal(s, want) and np.array_equal(e, want), (shape, shift)
        assert want[1] == 30
        cache = {}
        for _ in range(2):
            assert np.array_equal(host(HU.cupy_hd95(fa, fb, nl, fixed_cache=cache)), want)
        assert ("bits", nl) in cache
        old = HU.HD95_SURFACE_MAX_RADIUS
        HU.HD95_SURFACE_MAX_RADIUS = 1                                                   # force the hand-over to the transforms
        try:
            assert np.array_equal(host(HU.cupy_hd95(fa, fb, nl, fixed_cache=cache)), want)
        finally:
            HU.HD95_SURFACE_MAX_RADIUS = old
    with pytest.raises(NotImplementedError):
        HU.cupy_hd95(fa, fb, nl, precision=2, method="surface")
    with pytest.raises(ValueError):
        HU.cupy_hd95(fa, fb, nl, method="sorted")
    bad = a.copy()
    bad[0, 0, 0] = nl + 1
    with pytest.raises(RuntimeError):
        HU.cupy_hd95(dev(bad.astype(np.float32)), fb, nl)                               # F.one_hot would fail on the value nl + 1
    full = torch.ones_like(fa)
    for method in ("surface", "edt"):
        with pytest.raises(RuntimeError):
            HU.cupy_hd95(full, fb, nl, method=method)                                   # label 1 fills the fixed map: no outside voxel


def test_hd95_surface_equals_transforms_at_full_size(HU):
    """BASELINE's full extent (160 x 192 x 224, 13 labels, the sweep's synthetic anatomy): the surface-only path == the whole-volume
    transforms on a warped label map (rough surfaces, wrap-around slabs far from their label), with the fixed side cached."""
    from convexadam_amd import sweep
    shape = (160, 192, 224)
    seg_f, seg_m, _, _, nl = sweep._make_labels(shape, 1, torch.device(DEV))
    g = torch.Generator().manual_seed(9)
    disp = torch.zeros((1, 3) + shape)
    disp[0, 0], disp[0, 1], disp[0, 2] = 2.0, -1.0, 3.0
    disp += 0.4 * torch.randn(disp.shape, generator=g)
    warped = HU.warp_labels_nearest(seg_m, disp.to(DEV))
    cache_s, cache_e = {}, {}
    s = host(HU.cupy_hd95(seg_f, warped, nl, fixed_cache=cache_s))
    e = host(HU.cupy_hd95(seg_f, warped, nl, fixed_cache=cache_e, method="edt"))
    assert np.array_equal(s, e) and np.isfinite(s).all() and (s > 0).all()
    counts = HU.label_overlap_counts(seg_f, warped, nl + 1)
    assert np.array_equal(host(HU.cupy_hd95(seg_f, warped, nl, fixed_cache=cache_s, counts=counts)), s)
    assert np.array_equal(HU.dice_coeff(seg_f, warped, nl + 1, counts=counts).numpy(), HU.dice_coeff(seg_f, warped, nl + 1).numpy())
    with pytest.raises(ValueError):
        HU.cupy_hd95(seg_f, warped, nl, counts=counts[:, :-1])
