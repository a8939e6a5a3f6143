"""CPU checks of the drop-in boundary: the C-ABI library builds, loads without a GPU and exports every
symbol include/convexadam_hip.h declares; the ctypes table matches the header; argument validation and
the no-fallback rule hold.  No kernels are launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "convexadam_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(cvx_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def L():
    from convexadam_amd.csrc import build
    build.build()
    from convexadam_amd import _lib
    return _lib.lib()


def test_header_declares_expected_entry_points():
    names = header_functions()
    for must in ("cvx_mindssc_f32", "cvx_avgpool_f32", "cvx_correlate_f32", "cvx_coupled_convex_f32",
                 "cvx_inverse_consistency_f32", "cvx_resize_trilinear_f32", "cvx_adam_run_f32", "cvx_register_pair_f32"):
        assert must in names


def test_library_exports_every_declared_symbol(L):
    from convexadam_amd import _lib
    for name in header_functions():
        assert hasattr(L, name), "libconvexadam_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "ctypes table lacks %s" % name
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_pair_params_layout_matches_the_header_and_the_version_is_checked(L):
    """cvx_pair_params grew twice (ctx in round 3, adam_fast + reserved in round 4): the ctypes mirror must have the header's fields in
    the header's order, the library must report the header's CVX_ABI_VERSION, and the reserved tail must be refused when non-zero
    (what a struct laid out by an older header would look like to the library)."""
    import ctypes as C
    from convexadam_amd import _lib
    from convexadam_amd._lib import PairParams
    hdr = open(HEADER).read()
    ver = int(re.search(r"#define\s+CVX_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert L.cvx_version() == ver == _lib.ABI_VERSION
    body = re.search(r"typedef struct cvx_pair_params \{(.*?)\} cvx_pair_params;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for piece in decl.split(","):
            names.append(re.sub(r"\[.*\]", "", piece.strip().split()[-1].lstrip("*")))
    assert names == [f[0] for f in PairParams._fields_], (names, [f[0] for f in PairParams._fields_])
    assert C.sizeof(PairParams) == 19 * 4 + 4 + 8 + 4 * 4          # 19 ints/floats, padding, ctx pointer, adam_fast + reserved_[3]
    p = PairParams(64, 64, 64, 1, 2, 1.25, 4, 3, 5, 0, 2, 1, 0, 12.0)
    assert L.cvx_register_pair_workspace_bytes(C.byref(p)) > 0
    p.reserved_[1] = 7
    assert L.cvx_register_pair_workspace_bytes(C.byref(p)) == 0 and b"older header" in L.cvx_last_error()
    p.reserved_[1] = 0
    p.adam_fast = 1
    assert L.cvx_register_pair_workspace_bytes(C.byref(p)) > 0
    p.n_spline_pools = 2
    assert L.cvx_register_pair_workspace_bytes(C.byref(p)) == 0            # fast mode: packaged smoother only


def test_version_and_device_count(L):
    assert L.cvx_version() >= 1
    n = L.cvx_device_count()
    assert n >= 0
    if not torch.cuda.is_available():
        assert n == 0


def test_argument_validation_without_gpu(L):
    # invalid arguments are rejected before anything touches a device
    assert L.cvx_mindssc_f32(None, 8, 8, 8, 1, 2, None, None, 0, None) == -1
    assert b"null" in L.cvx_last_error()
    assert L.cvx_correlate_f32(None, None, 12, 4, 4, 4, 2, None, None, None, 0, None) == -1
    dummy = C.c_void_p(256)
    assert L.cvx_correlate_f32(dummy, dummy, 12, 4, 4, 4, 16, dummy, None, dummy, 1 << 40, None) == -4  # hw > 15 unsupported
    assert L.cvx_correlate_f32(dummy, dummy, 12, 4, 4, 4, 2, dummy, None, dummy, 16, None) == -2         # workspace too small
    assert L.cvx_box_smooth_f32(dummy, 3, 4, 4, 4, 4, 1, C.c_void_p(512), None, 0, None) == -1           # even kernel
    assert b"odd" in L.cvx_last_error()
    assert L.cvx_box_grow_f32(dummy, 3, 4, 4, 4, 3, C.c_void_p(512), None) == -1 and b"even" in L.cvx_last_error()   # odd kernel: the other entry
    assert L.cvx_box_grow_f32(dummy, 3, 4, 4, 4, 2, dummy, None) == -1                                               # in place
    assert L.cvx_box_grow_f32(dummy, 3, 4, 4, 1, 4, C.c_void_p(512), None) == -1 and b"padding" in L.cvx_last_error()
    # pooled descriptor (round 6): null pointers, bad windows, windows that do not tile, missing scratch for the two-pass settings
    assert L.cvx_mindssc_pooled_f32(None, 8, 8, 8, 1, 2, 2, None, 0, None, None, 0, None, 0, None, None) == -1 and b"null" in L.cvx_last_error()
    assert L.cvx_mindssc_pooled_f32(dummy, 8, 8, 8, 1, 2, 0, dummy, 0, None, None, 0, dummy, 1 << 30, None, None) == -1 and b"windows" in L.cvx_last_error()
    assert L.cvx_mindssc_pooled_f32(dummy, 8, 8, 8, 1, 2, 2, dummy, 2, None, None, 0, dummy, 1 << 30, None, None) == -1 and b"second output" in L.cvx_last_error()
    assert L.cvx_mindssc_pooled_f32(dummy, 20, 20, 20, 1, 2, 5, dummy, 0, None, None, 0, dummy, 1 << 30, None, None) == -4        # 5^3 windows: use the two operators
    assert L.cvx_mindssc_pooled_scratch_bytes(24, 24, 24, 2, 2, 6, 2) == 12 * 24 * 24 * 24 * 4       # radius 2: two passes through the raw distances
    assert L.cvx_mindssc_pooled_f32(dummy, 24, 24, 24, 2, 2, 6, dummy, 2, dummy, None, 0, dummy, 1 << 30, None, None) == -2 and b"scratch" in L.cvx_last_error()
    assert L.cvx_mindssc_pooled_scratch_bytes(24, 24, 24, 1, 2, 6, 2) in (0, 12 * 24 * 24 * 24 * 4)   # (0 with option mind_single, the single pass)
    # evaluation operators: null pointers and degenerate extents
    assert L.cvx_jacobian_det_f32(None, 8, 8, 8, 0, None, None) == -1
    assert L.cvx_jacobian_det_f32(dummy, 4, 8, 8, 0, dummy, None) == -1 and b"crop" in L.cvx_last_error()
    assert L.cvx_jacobian_stats_f64(dummy, 0, dummy, None) == -1
    assert L.cvx_warp_labels_nearest_f32(dummy, dummy, 8, 8, 8, dummy, dummy, dummy, dummy, None) == -1   # in place
    assert L.cvx_label_overlap_i64(dummy, dummy, 10, 0, dummy, None) == -1 and b"num_labels" in L.cvx_last_error()
    assert L.cvx_map_coordinates_linear_f64(dummy, dummy, 4, 4, 0, C.c_void_p(512), None) == -1
    # Hausdorff-95 building blocks
    assert L.cvx_label_mask_f32(dummy, 4, 4, 4, 1, 0, dummy, dummy, dummy, None) == -1 and b"precision" in L.cvx_last_error()
    assert L.cvx_label_mask_f32(dummy, 4, 4, 4, 1, 1, None, dummy, dummy, None) == -1          # count alone may be NULL
    assert L.cvx_label_mask_scaled_f32(dummy, 4, 4, 4, 1, 6, 6, 0, C.c_float(1 / 1.5), dummy, dummy, None, None) == -1 and b"extent" in L.cvx_last_error()
    assert L.cvx_label_mask_scaled_f32(dummy, 4, 4, 4, 1, 6, 6, 6, C.c_float(0.0), dummy, dummy, None, None) == -1 and b"scale_inv" in L.cvx_last_error()
    assert L.cvx_edt_sqdist_i32(dummy, dummy, 40000, 40000, 4, dummy, None) == -1 and b"int32" in L.cvx_last_error()
    assert L.cvx_surface_hist_i64(dummy, dummy, dummy, 0, 8, dummy, dummy, None) == -1
    assert L.cvx_hist_order_stats_i64(dummy, 0, 0, 0, dummy, None) == -1
    assert L.cvx_hist_percentile_neighbours_i64(dummy, 8, C.c_float(1.5), dummy, None) == -1
    assert L.cvx_edt_squared_i32(dummy, 1, 40000, 40000, 4, dummy, dummy, 1 << 40, None) == -1 and b"int32" in L.cvx_last_error()
    assert L.cvx_edt_squared_i32(dummy, 2, 4, 4, 4, dummy, dummy, 16, None) == -2
    assert L.cvx_edt_squared_i32(dummy, 0, 4, 4, 4, dummy, dummy, 1 << 20, None) == -1
    # surface-only HD95 (surfdist.hip): one bit per (label, voxel), 64 voxels along D per word
    assert L.cvx_label_bits_bytes(160, 192, 224, 13) == 13 * 160 * 192 * 4 * 8 and L.cvx_label_bits_bytes(4, 4, 65, 2) == 2 * 16 * 2 * 8
    assert L.cvx_label_bits_bytes(4, 4, 0, 2) == 0
    assert L.cvx_label_bits_u64(dummy, 4, 4, 4, 256, dummy, None) == -1 and b"255" in L.cvx_last_error()
    assert L.cvx_label_bits_u64(None, 4, 4, 4, 3, dummy, None) == -1
    act = (C.c_uint64 * 4)(2, 0, 0, 0)
    assert L.cvx_surface_distance_hist_i64(dummy, dummy, 4, 4, 4, 3, C.cast(act, C.c_void_p), 8, dummy, 4, dummy, 1, 0, None) == -1   # stride < nbins
    assert L.cvx_surface_distance_hist_i64(dummy, dummy, 4, 4, 4, 3, None, 8, dummy, 8, dummy, 1, 0, None) == -1                       # no label mask
    assert L.cvx_surface_distance_hist_i64(dummy, dummy, 4096, 4, 4, 3, C.cast(act, C.c_void_p), 8, dummy, 8, dummy, 1, 0, None) == -1 and b"2047" in L.cvx_last_error()
    assert L.cvx_surface_distance_hist_i64(dummy, dummy, 4, 4, 4, 3, C.cast(act, C.c_void_p), 8, dummy, 8, dummy, 1, -1, None) == -1   # negative radius


def test_workspace_queries(L):
    from convexadam_amd._lib import PairParams
    # OASIS-size pair: raw (K*h*w*dp) + ssd (K*v) dominate
    p = PairParams(160, 192, 224, 1, 2, 1.25, 6, 6, 80, 0, 2, 1, 0, 12.0)
    n = L.cvx_register_pair_workspace_bytes(C.byref(p))
    K, v = 13 ** 3, 26 * 32 * 37
    assert n > 4 * K * v * 2 + 2 * 12 * 160 * 192 * 224 * 4          # two cost volumes + two descriptors
    assert n < 3 * 1024 ** 3
    bad = PairParams(160, 192, 224, 1, 2, 1.25, 6, 6, 0, 0, 2, 1, 0, 12.0)       # niter 0 with lambda > 0
    assert L.cvx_register_pair_workspace_bytes(C.byref(bad)) == 0
    # fused correlation kernel: padded feature copies only, no raw SSD intermediate -- also for C >= 16 (cascade channel sum) and for the
    # tall planes of the sweep's fine grids (y tiles); the round-1 kernels with their raw intermediate remain behind `corr_unfused`
    assert L.cvx_correlate_workspace_bytes(12, 26, 32, 37, 6) < 16 * 1024 ** 2
    assert L.cvx_correlate_workspace_bytes(32, 26, 32, 37, 6) >= 4 * K * 26 * 32 * 40       # C >= 16: the faster round-1 kernels by default
    L.cvx_set_option(b"corr_fused_all", 1)
    assert L.cvx_correlate_workspace_bytes(32, 26, 32, 37, 6) < 32 * 1024 ** 2
    L.cvx_set_option(b"corr_fused_all", 0)
    for gs in (2, 3, 4, 5):                                        # every stage-1 grid of the sweep at 160 x 192 x 224
        h, w, d = 160 // gs, 192 // gs, 224 // gs
        # (+ 24 bytes per voxel: keys, runner-up, winners, minima and work list of the certified argmin, cvx_corr_opts.fast = 2)
        assert L.cvx_correlate_workspace_bytes(12, h, w, d, 5) < 2 * 12 * (h + 10) * (w + 10) * (d + 32) * 4 + 24 * h * w * d + (1 << 20), gs
    old = L.cvx_get_option(b"corr_unfused")
    L.cvx_set_option(b"corr_unfused", 1)
    assert L.cvx_correlate_workspace_bytes(20, 26, 32, 37, 6) >= 4 * K * 26 * 32 * 40
    L.cvx_set_option(b"corr_unfused", old)


def test_no_cpu_fallback():
    from convexadam_amd import convex_adam_utils as U, convex_adam_MIND as M
    x = torch.zeros(1, 1, 8, 8, 8)
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            U.MINDSSC(x, 1, 2, device="cuda")
    with pytest.raises(RuntimeError, match="no CPU"):
        U.correlate(torch.zeros(1, 12, 4, 4, 4), torch.zeros(1, 12, 4, 4, 4), 2, 2, (8, 8, 8), 12)
    with pytest.raises(RuntimeError, match="no CPU|HIP device"):
        M.convex_adam_pt(np.zeros((16, 16, 16), np.float32), np.zeros((16, 16, 16), np.float32), device=torch.device("cpu"))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "convexadam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libcvx_oracle" not in txt, f


def test_option_table(L):
    """Run-time variant switches: known names round-trip, unknown names are rejected (no GPU needed)."""
    for name in (b"mind_tiled", b"mm_tx", b"box_tiled", b"no_prune", b"corr_unfused", b"prune_stream_above"):
        old = L.cvx_get_option(name)
        assert L.cvx_set_option(name, 7) == 0 and L.cvx_get_option(name) == 7
        assert L.cvx_set_option(name, old) == 0
    assert L.cvx_set_option(b"bogus", 1) != 0 and L.cvx_get_option(b"bogus") == -1


def test_contexts_are_per_caller(L):
    """cvx_context_*: switches live in a context; a context bound to one thread is invisible to another thread and to the default
    context; cvx_pair_params.ctx names one per call (no GPU needed: workspace queries depend on mind_mean_threads)."""
    import threading
    from convexadam_amd._lib import PairParams
    from convexadam_amd.context import Context
    base = L.cvx_mindssc_workspace_bytes(32, 32, 32, 1, 2)
    ctx = Context(mind_mean_threads=8, box_yt=4)
    assert ctx.get_option("mind_mean_threads") == 8 and L.cvx_get_option(b"mind_mean_threads") == 0
    assert L.cvx_context_set_option(ctx.handle, b"bogus", 1) != 0 and L.cvx_context_get_option(ctx.handle, b"bogus") == -1
    seen = {}

    def other_thread():
        seen["other"] = L.cvx_mindssc_workspace_bytes(32, 32, 32, 1, 2)

    with ctx:
        big = L.cvx_mindssc_workspace_bytes(32, 32, 32, 1, 2)
        th = threading.Thread(target=other_thread)
        th.start(); th.join()
        with Context(mind_mean_threads=0):                       # nesting restores the outer binding
            assert L.cvx_mindssc_workspace_bytes(32, 32, 32, 1, 2) == base
        assert L.cvx_mindssc_workspace_bytes(32, 32, 32, 1, 2) == big
    assert big > base and seen["other"] == base and L.cvx_mindssc_workspace_bytes(32, 32, 32, 1, 2) == base
    # per call through cvx_pair_params.ctx
    p = PairParams(64, 64, 64, 1, 2, 1.25, 4, 3, 5, 0, 2, 1, 0, 12.0)
    n0 = L.cvx_register_pair_workspace_bytes(C.byref(p))
    p.ctx = ctx.handle
    assert L.cvx_register_pair_workspace_bytes(C.byref(p)) > n0
    assert L.cvx_mindssc_workspace_bytes(32, 32, 32, 1, 2) == base   # the scope ended with the call
    # tables: NULL restores the default without touching a device
    assert L.cvx_context_set_adam_sqrt_table(ctx.handle, None, None) == 0
    assert L.cvx_context_set_mind_exp_table(ctx.handle, None, 0, 0, None) == 0
    ctx.close()


def test_every_option_has_its_environment_variable():
    """The default context reads CVX_<NAME> for every switch (positional initialiser in api.hip): each name must map to its own field."""
    import subprocess
    import sys
    names = ["mind_tiled", "mind_overlap", "mm_tx", "mm_slots", "box_tiled", "no_prune", "corr_unfused", "corr_fused_all", "prune_stream_above",
             "cf_census", "cf_prio", "warp_flat", "box_yt", "box_wg_target", "box_xsplit", "box_cpt", "box_uneven", "box_adam_role", "box_dpp", "box_pk",
             "box_prio", "label_pow_block", "mind_mean_threads"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    for i, n in enumerate(names):
        env["CVX_" + n.upper()] = str(11 + i)
    code = ("from convexadam_amd import _lib; L = _lib.lib(); import sys; names = sys.argv[1].split(',');"
            "print(','.join(str(L.cvx_get_option(n.encode())) for n in names + ['census_ptr']))")
    out = subprocess.run([sys.executable, "-c", code, ",".join(names)], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
    assert out == ",".join(str(11 + i) for i in range(len(names))) + ",0", out
