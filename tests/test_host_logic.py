"""CPU tests of the host-side logic: table helpers against torch itself, the synthetic-data recipe, the
API mirror's error behaviour, and the sharded multi-GPU driver over gloo (world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_affine_base_matches_torch_affine_grid():
    from convexadam_amd.convex_adam_utils import affine_base
    for S in list(range(2, 130)) + [160, 192, 224, 255, 399]:
        g = F.affine_grid(torch.eye(3, 4).unsqueeze(0), (1, 1, 2, 2, S), align_corners=False)[0, 0, 0, :, 0].numpy()
        assert np.array_equal(affine_base(S), g), S


def test_disp_mesh_matches_torch_affine_grid():
    from convexadam_amd.convex_adam_utils import disp_mesh
    for hw in range(1, 9):
        n = 2 * hw + 1
        m = F.affine_grid(hw * torch.eye(3, 4).unsqueeze(0), (1, 1, n, n, n), align_corners=True).permute(0, 4, 1, 2, 3).reshape(3, -1)
        assert np.array_equal(disp_mesh(hw), m.numpy()), hw
    # k -> (dH, dW, dD) with the D-shift slowest
    m = disp_mesh(2)
    assert m[:, 3 * 25 + 1 * 5 + 4].tolist() == [2.0, -1.0, 1.0]


def test_oracle_tables_agree_with_library(orc):
    from convexadam_amd.convex_adam_utils import affine_base, disp_mesh
    assert all(np.array_equal(affine_base(S), orc.affine_base(S)) for S in range(1, 300))
    assert all(np.array_equal(disp_mesh(h), orc.disp_mesh(h)) for h in range(0, 9))


def test_outer_sum_tail_rule_matches_torch(orc):
    """ATen sums an outer dimension in blocks of 32 columns; the last (ncols mod 32) columns use a 4-way
    interleaved order.  The oracle (and the HIP kernels) restate that rule."""
    import ctypes as C
    lib = orc.lib()
    lib.orc_outer_sum_rows.restype = C.c_float
    lib.orc_outer_sum_rows.argtypes = [np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), C.c_int64, C.c_int]
    g = torch.Generator().manual_seed(0)
    for Cn, N in ((12, 1000), (20, 333), (40, 97), (12, 64)):
        x = torch.rand(Cn, N, generator=g)
        t = x.sum(0).numpy()
        xn = x.numpy()
        tail = (N // 32) * 32
        mine = np.array([lib.orc_outer_sum_rows(np.ascontiguousarray(xn[:, j]), Cn, int(j >= tail)) for j in range(N)], np.float32)
        assert np.array_equal(mine, t), (Cn, N)


def test_phantom_is_deterministic():
    from convexadam_amd.phantom import phantom, smooth_warp, label_phantom
    a, b = phantom((16, 12, 20), 3, 7), phantom((16, 12, 20), 3, 7)
    assert torch.equal(a, b) and a.shape == (16, 12, 20) and a.dtype == torch.float32
    assert not torch.equal(a, phantom((16, 12, 20), 4, 7))
    assert smooth_warp((8, 8, 8), 1).shape == (1, 8, 8, 8, 3)
    assert label_phantom((8, 8, 8), 5, 1).max() <= 4


def test_validate_image_mirrors_reference_errors():
    from convexadam_amd.convex_adam_utils import validate_image
    assert isinstance(validate_image(np.zeros((2, 2, 2), np.float32)), torch.Tensor)
    t = torch.zeros(2, 2, 2)
    assert validate_image(t) is t
    with pytest.raises(ValueError):
        validate_image("not an image")


def test_shard_assignment():
    from convexadam_amd.sweep import shard_items
    items = list(range(10))
    assert shard_items(items, 0, 1) == items
    parts = [shard_items(items, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == items
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_items([], 1, 2) == []


def test_sweep_grid_enumeration():
    from convexadam_amd.sweep import sweep_settings
    s = sweep_settings()
    assert len(s) == 256 and len({tuple(sorted(d.items())) for d in s}) == 256
    assert all(d["disp_hw"] <= 8 and d["grid_sp"] >= 2 for d in s)


def _run_sweep(tmp_path, port, extra, nproc=2):
    out = tmp_path / "res.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "convexadam_amd", "sweep.py"), "--dry-run", "--out", str(out)] + extra
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:]
    import json
    return json.loads(out.read_text()), out


@pytest.mark.timeout(300)
def test_sharded_driver_gloo_world_size_2(tmp_path):
    """The N>1 path: one process per rank, items drawn from the shared queue (or dealt round-robin with --static), no data-path
    collective, results gathered on rank 0.  Runs on CPU with the gloo backend and the driver's --dry-run mode (no kernels)."""
    res, _ = _run_sweep(tmp_path, 29613, ["--pairs", "5", "--settings", "3"])
    assert res["world_size"] == 2 and res["n_items"] == 15
    assert sorted(res["items_done"]) == list(range(15))
    parts = list(res["per_rank"].values())
    assert sorted(sum(parts, [])) == list(range(15))                       # every item exactly once, whoever drew it
    res, _ = _run_sweep(tmp_path, 29615, ["--pairs", "5", "--settings", "3", "--static"])
    assert sorted(len(v) for v in res["per_rank"].values()) == [7, 8]
    # two worker threads per rank (two registrations in flight per GPU): still every item exactly once
    res, _ = _run_sweep(tmp_path, 29621, ["--pairs", "5", "--settings", "3", "--workers", "2"])
    assert res["workers_per_rank"] == 2 and sorted(res["items_done"]) == list(range(15))
    assert sorted(sum(res["per_rank"].values(), [])) == list(range(15))


@pytest.mark.timeout(300)
def test_two_stage_sweep_queue_and_resume_gloo(tmp_path):
    """Two-stage mode of the reference's self-configuring scripts (convex ranking, then one Adam run per item scored at 4 snapshots x
    4 smoothings), world size 2 on gloo: every item runs once, the per-rank logs are append-only, and a second start with --resume
    after deleting part of the log re-runs only what is missing."""
    import json
    res, out = _run_sweep(tmp_path, 29617, ["--pairs", "3", "--stage1", "6", "--stage2", "4"])
    assert res["stage1"]["n_items"] == 18 and res["stage1"]["fresh_items"] == 18
    assert res["stage2"]["n_items"] == 12 and res["stage2"]["evaluations"] == 12 * 16
    assert sorted(sum(res["stage1"]["per_rank"].values(), [])) == list(range(18))
    b = res["stage2"]["best_setting"]
    assert b["selected_niter"] in (60, 80, 100, 120) and 0 <= b["extra_smooth"] <= 3 and {"grid_sp_adam", "avg_n", "lambda_weight"} <= set(b)
    logs = sorted(str(p) for p in tmp_path.glob("res.json.rank*.jsonl"))
    assert len(logs) == 2
    recs = [json.loads(l) for f in logs for l in open(f)]
    assert sum("header" in r for r in recs) == 2 and all("header" in json.loads(open(f).readline()) for f in logs)   # every log names its run
    recs = [r for r in recs if "header" not in r]
    assert len(recs) == 30
    # kill simulation: drop the last 5 lines of rank 0's log (one of them cut in the middle), then resume
    lines = open(logs[0]).read().splitlines()
    kept = lines[:-5]
    open(logs[0], "w").write("\n".join(kept) + "\n" + lines[-5][: len(lines[-5]) // 2])
    res2, _ = _run_sweep(tmp_path, 29619, ["--pairs", "3", "--stage1", "6", "--stage2", "4", "--resume"])
    assert res2["stage1"]["n_items"] == 18 and res2["stage2"]["n_items"] == 12
    assert res2["stage1"]["fresh_items"] + res2["stage2"]["fresh_items"] == 5
    assert res2["stage1"]["best_setting"] == res["stage1"]["best_setting"] and res2["stage2"]["best_setting"] == res["stage2"]["best_setting"]


def test_result_log_header_and_stale_files(tmp_path):
    """Per-rank logs carry a header naming the run: --resume ignores files of another sweep (other shape / pair count / setting counts)
    and files without a header; a fresh start removes every rank file of the output name, also those of an earlier run with more ranks."""
    from convexadam_amd.sweep import ResultLog
    out = str(tmp_path / "r.json")
    run_a, run_b = dict(shape=[8, 8, 8], pairs=2), dict(shape=[8, 8, 8], pairs=3)
    for rank in range(3):                                                   # an earlier run with three ranks
        lg = ResultLog(out, rank, False, run_a)
        lg.start()
        lg.add(dict(stage="convex", setting=rank, pair=0, v=rank))
    assert len(list(tmp_path.glob("r.json.rank*.jsonl"))) == 3
    r0 = ResultLog(out, 0, True, run_a)                                    # resume of the same run: all three files count
    assert sorted(k[1] for k in r0.done) == [0, 1, 2] and not r0.ignored_files
    rb = ResultLog(out, 0, True, run_b)                                    # another sweep: nothing is reused, this rank's file starts over
    assert not rb.done and len(rb.ignored_files) == 3
    assert "header" in open(out + ".rank0.jsonl").readline()
    rb.add(dict(stage="convex", setting=5, pair=1, v=9))
    rc = ResultLog(out, 1, True, run_b)
    assert list(rc.done) == [("convex", 5, 1)] and len(rc.ignored_files) == 2
    fresh = ResultLog(out, 0, False, run_b)                                # fresh start with fewer ranks: rank 0 removes every old file
    fresh.start()
    assert [p.name for p in tmp_path.glob("r.json.rank*.jsonl")] == ["r.json.rank0.jsonl"]


def test_sweep_worker_failure_is_reported(tmp_path):
    """An exception inside a worker does not leave the other ranks waiting in the gather: the failing item is recorded, every rank
    finishes its phase, and the run ends with one error message (world size 2, gloo, dry run with an injected failure)."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), CVX_SWEEP_FAIL_ITEM="3")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29623",
           os.path.join(ROOT, "convexadam_amd", "sweep.py"), "--dry-run", "--pairs", "4", "--settings", "2", "--out", str(tmp_path / "f.json")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode != 0 and "sweep: 1 item(s) failed" in r.stdout and "injected failure" in r.stdout, r.stdout[-2000:]


def test_torch_pow_restatement():
    """The label weights of the multi-channel path use `(...).float().pow(.3)` (convex_adam_nnUNet.py:31): torch evaluates the leading
    blocks of 32 elements with Sleef's powf and the trailing n mod 32 with the scalar double pow.  The oracle's restatement
    (orc_torch_pow_at) and the library's host helper (cvx_label_weights_host, which also restates weight.mean()) against torch itself."""
    import ctypes as C
    from oracle import oracle
    from convexadam_amd import _lib
    oracle.build()
    lo = oracle.lib()
    lo.orc_torch_pow_at.restype = C.c_float
    lo.orc_torch_pow_at.argtypes = [C.c_float, C.c_double, C.c_int64, C.c_int64]
    # the vectorised loop covers blocks of 2 x the vector width of THIS host's ATen build (the goldens come from an AVX-512 host: 32)
    block = 32 if "512" in torch.backends.cpu.get_cpu_capability() else 16
    lo.orc_set_pow_block(block)
    _lib.lib().cvx_set_option(b"label_pow_block", block)
    rng = np.random.default_rng(3)
    for n in (1, 5, 31, 32, 33, 64, 95, 200, 4096 + 7):
        x = torch.from_numpy(rng.integers(1, 1 << 24, n).astype(np.float32))
        for y in (0.3, 1.7, 0.123):                                  # (0.5, 2, 3, -1 ... are special-cased by torch: sqrt, x*x, ...)
            ref = torch.pow(x, y).numpy()
            got = np.array([lo.orc_torch_pow_at(float(v), y, i, n) for i, v in enumerate(x.numpy())], np.float32)
            assert np.array_equal(got, ref), (n, y)
    L = _lib.lib()
    for n in (1, 3, 8, 13, 18, 32, 33, 100, 255):
        hf = rng.integers(0, 3000000, n).astype(np.int64)
        hm = rng.integers(0, 3000000, n).astype(np.int64)
        hf[hf + hm == 0] = 1
        w = 1 / ((torch.from_numpy(hf) + torch.from_numpy(hm)) + 1e-32).float().pow(.3)          # the reference's expression (:31-32)
        w /= w.mean()
        pres, wt = np.zeros(n, np.int32), np.zeros(n, np.float32)
        assert L.cvx_label_weights_host(hf.ctypes.data_as(C.c_void_p), hm.ctypes.data_as(C.c_void_p), n - 1, pres.ctypes.data_as(C.c_void_p),
                                        wt.ctypes.data_as(C.c_void_p)) == n
        assert np.array_equal(wt, w.numpy()), n
    lo.orc_set_pow_block(32)
    L.cvx_set_option(b"label_pow_block", 32)


def test_sweep_settings_tables():
    from convexadam_amd.sweep import item_cost, stage1_settings, stage2_settings
    s1, s2 = stage1_settings(100), stage2_settings(75)
    assert len(s1) == 100 and len(s2) == 75 and s1 == stage1_settings(100)
    assert all(1 <= c["mind_r"] <= 3 and 1 <= c["mind_d"] <= 3 and 2 <= c["grid_sp"] <= 5 and 2 <= c["disp_hw"] <= 7 for c in s1)
    assert all(c["disp_hw"] <= 5 for c in s1 if c["grid_sp"] == 2)
    assert all(1 <= c["grid_sp_adam"] <= 4 and 1 <= c["avg_n"] <= 7 and 0.4 <= c["lambda_weight"] <= 1.6 for c in s2)
    shape = (160, 192, 224)
    assert item_cost(dict(grid_sp=4, disp_hw=6), shape) > 10 * item_cost(dict(grid_sp=8, disp_hw=3), shape)   # what the queue orders by


def test_geometry_helpers_importable_under_reference_names():
    """tests/test_convex_adam_mind_aniso.py:10-12, convex_adam_translation.py:9 and apply_convex.py:13,27 of the reference: these
    names are imported from convexAdam.*; the geometry helpers take SimpleITK images (through SimpleITK) or the built-in
    convexadam_amd.imageio.Image, and say that SimpleITK is missing when they are handed anything else.  (Own process: the
    import shim must not shadow the live reference import of tests/test_oracle_vs_reference_live.py.)"""
    code = """
import numpy as np
from convexAdam.convex_adam_utils import resample_img, resample_moving_to_fixed, rescale_displacement_field
from convexAdam.apply_convex import apply_convex, apply_convex_original_moving
from convexAdam.convex_adam_translation import convex_adam_translation, apply_translation
class Foreign:                       # an image object of another library: only SimpleITK's resampler could take it
    def GetSize(self): return (4, 4, 4)
    def GetSpacing(self): return (1.0, 1.0, 1.0)
    def GetOrigin(self): return (0.0, 0.0, 0.0)
    def GetDirection(self): return (1, 0, 0, 0, 1, 0, 0, 0, 1)
try:
    import SimpleITK
except ImportError:
    f = Foreign()
    for fn, args in ((resample_img, (f, (1.0, 1.0, 1.0))), (resample_moving_to_fixed, (f, f)),
                     (rescale_displacement_field, (np.zeros((4, 4, 4, 3)), f, f, f))):
        try:
            fn(*args)
        except ImportError as e:
            assert "SimpleITK" in str(e)
        else:
            raise SystemExit("no ImportError from " + fn.__name__)
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_translation_wrapper_arithmetic():
    """convex_adam_translation.py:12-29, 88-103 of the reference: direction-cosine product and the field -> whole-voxel translation
    reduction (checked against the reference's expressions written out with numpy)."""
    from convexadam_amd.convex_adam_translation import field_to_translation, index_translation_to_world_translation
    rng = np.random.default_rng(4)
    direction = (0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0)
    assert np.array_equal(index_translation_to_world_translation((1.0, 2.0, 3.0), direction), np.array([-2.0, 1.0, 3.0]))
    field = rng.standard_normal((5, 6, 7, 3)) * 4
    spacing_xyz = (0.7, 1.5, 3.0)
    mask = rng.random((5, 6, 7)) > 0.5
    for m in (None, mask):
        t_zyx = np.mean(field[m], axis=0) if m is not None else np.mean(field, axis=(0, 1, 2))
        sp_zyx = np.array(list(spacing_xyz)[::-1])
        ref = tuple(list((np.round(t_zyx / sp_zyx, decimals=0) * sp_zyx)[::-1]))
        assert field_to_translation(field, spacing_xyz, m) == ref


def test_nifti_io_round_trip(tmp_path):
    """Built-in NIfTI-1 reader / writer (used by the file wrappers when nibabel is absent): data types, gzip, 4-D fields, affines
    with rotation / flips, the quaternion (qform) path, scl_slope / scl_inter."""
    import struct
    from convexadam_amd import nifti_io as N
    rng = np.random.default_rng(0)
    th = 0.3
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    aff = np.eye(4); aff[:3, :3] = rot * np.array([0.8, 1.2, 2.5]); aff[:3, 3] = (10, -20, 5.5)
    flip = np.diag([-1.0, 1.0, 1.0, 1.0]); flip[:3, 3] = (90, -126, -72)
    for dt in (np.float32, np.float64, np.int16, np.uint8):
        a = (rng.standard_normal((5, 6, 7)) * 50).astype(dt)
        for ext in (".nii", ".nii.gz"):
            for A in (aff, flip):
                p = str(tmp_path / ("t" + ext))
                N.save(a, A, p)
                b, A2 = N.load(p)
                assert b.dtype == np.float64 and np.array_equal(b, a.astype(np.float64)) and np.allclose(A2, A, atol=1e-6)
    field = rng.standard_normal((4, 5, 6, 3))
    p = str(tmp_path / "disp.nii.gz")
    N.save_image(field, np.eye(4), p)
    assert np.array_equal(N.load_fdata(p), field) and np.array_equal(N.load_affine(p), np.eye(4))
    # qform only (sform_code = 0) and intensity scaling
    p = str(tmp_path / "q.nii")
    N.save((rng.standard_normal((3, 4, 5)) * 10).astype(np.int16), aff, p)
    raw = bytearray(open(p, "rb").read())
    struct.pack_into("<h", raw, 254, 0)                       # sform_code
    struct.pack_into("<2f", raw, 112, 0.5, 3.0)               # scl_slope, scl_inter
    open(p, "wb").write(bytes(raw))
    b, A2 = N.load(p)
    ref = np.frombuffer(bytes(raw), "<i2", 60, 352).reshape((3, 4, 5), order="F").astype(np.float64) * 0.5 + 3.0
    assert np.array_equal(b, ref) and np.allclose(A2, aff, atol=1e-5)
    with pytest.raises(ValueError):
        open(p, "wb").write(b"x" * 400); N.load(p)



# ---- built-in images with geometry, MetaImage files, geometry helpers without SimpleITK ------------------------------
def _blob_image(shape_zyx, spacing, origin, direction, seed=0):
    from convexadam_amd.imageio import Image
    rng = np.random.default_rng(seed)
    return Image(rng.normal(100, 20, shape_zyx).astype(np.float32), spacing, origin, direction)


def test_metaimage_round_trip(tmp_path):
    from convexadam_amd.imageio import read_mha, write_mha
    rot = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    for dtype, compress in ((np.float32, False), (np.int16, True), (np.uint8, False), (np.float64, True)):
        img = _blob_image((5, 6, 7), (0.5, 0.75, 2.0), (-10.0, 3.5, 8.0), rot.reshape(-1))
        img.array = img.array.astype(dtype)
        path = str(tmp_path / "x.mha")
        write_mha(img, path, compress=compress)
        back = read_mha(path)
        assert back.array.dtype == dtype and np.array_equal(back.array, img.array)
        assert back.GetSize() == (7, 6, 5) and np.allclose(back.GetSpacing(), img.GetSpacing()) and np.allclose(back.GetOrigin(), img.GetOrigin())
        assert np.allclose(np.array(back.GetDirection()), rot.reshape(-1))
    # a header written by other tools: big-endian data, keys in another order, mhd-style external file
    raw = np.arange(24, dtype=">i2").reshape(2, 3, 4)
    (tmp_path / "d.raw").write_bytes(raw.tobytes())
    (tmp_path / "d.mhd").write_text("ObjectType = Image\nNDims = 3\nDimSize = 4 3 2\nElementSpacing = 1 2 3\nElementType = MET_SHORT\n"
                                    "BinaryDataByteOrderMSB = True\nOffset = 1 2 3\nElementDataFile = d.raw\n")
    img = read_mha(str(tmp_path / "d.mhd"))
    assert np.array_equal(img.array, raw.astype(np.int16)) and img.GetSpacing() == (1.0, 2.0, 3.0) and img.GetOrigin() == (1.0, 2.0, 3.0)


def test_resampling_follows_itk_geometry():
    """A linear ramp in physical space is reproduced exactly by linear resampling onto any grid that stays inside the source buffer;
    outside the buffer the value is 0; resample_img's size rule is floor(n * old / new + 0.5)."""
    from convexadam_amd.convex_adam_utils import resample_img, resample_moving_to_fixed
    from convexadam_amd.imageio import Image
    th = np.deg2rad(20.0)
    D = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    spacing, origin = np.array([0.8, 1.25, 2.0]), np.array([-5.0, 7.0, 1.0])
    nz, ny, nx = 12, 14, 16
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    phys = origin[:, None] + D @ (spacing[:, None] * np.stack([i, j, k], 0).reshape(3, -1))
    coef = np.array([0.3, -0.7, 1.1])
    src = Image((coef @ phys + 2.0).reshape(nz, ny, nx), spacing, origin, D.reshape(-1))
    iso = resample_img(src, (1.0, 1.0, 1.0))
    assert iso.GetSize() == tuple(int(n * s / 1.0 + 0.5) for n, s in zip((nx, ny, nz), spacing))
    A, o = iso.index_to_physical_matrix()
    kk, jj, ii = np.meshgrid(*[np.arange(n) for n in iso.array.shape], indexing="ij")
    p = o[:, None] + A @ np.stack([ii, jj, kk], 0).reshape(3, -1)
    As, os_ = src.index_to_physical_matrix()
    ci = np.linalg.solve(As, p - os_[:, None])
    inside = np.all((ci >= 0) & (ci <= np.array([nx - 1, ny - 1, nz - 1])[:, None]), 0)
    assert inside.sum() > 1000
    assert np.allclose(iso.array.reshape(-1)[inside], (coef @ p + 2.0)[inside], atol=1e-9)
    outside = np.any((ci < -0.5) | (ci > np.array([nx, ny, nz])[:, None] - 0.5), 0)
    # a grid that sticks out of the source on every side
    big = Image(np.zeros((20, 20, 20)), (1.5, 1.5, 1.5), (-15.0, -2.0, -6.0), np.eye(3).reshape(-1))
    onto = resample_moving_to_fixed(big, src)
    assert onto.GetSize() == big.GetSize() and onto.GetSpacing() == big.GetSpacing()
    assert (onto.array == 0).sum() > 100 and (onto.array != 0).sum() > 100
    del outside


def test_rescale_displacement_field_and_translation_without_simpleitk():
    """convex_adam_utils.py:309-351 / convex_adam_translation.py:12-54 on built-in images: a field estimated on an isotropic resampled
    grid is carried to an anisotropic, rotated moving image: components are resampled, rotated by inv(D_fixed) D_moving and scaled by
    the spacing ratio; apply_translation shifts the origin by the world-space translation."""
    from convexadam_amd.convex_adam_translation import apply_translation, field_to_translation, index_translation_to_world_translation
    from convexadam_amd.convex_adam_utils import rescale_displacement_field, resample_img
    from convexadam_amd.imageio import Image
    Rz = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    fixed = Image(np.zeros((10, 12, 14), np.float32), (1.0, 1.0, 2.0), (0.0, 0.0, 0.0), np.eye(3).reshape(-1))
    fixed_1mm = resample_img(fixed, (1.0, 1.0, 1.0))
    moving = Image(np.zeros((8, 20, 18), np.float32), (0.5, 0.5, 2.5), (6.0, 0.0, 0.0), Rz.reshape(-1))
    field = np.zeros(fixed_1mm.array.shape + (3,))
    field[..., 0], field[..., 1], field[..., 2] = 2.0, -1.0, 0.5                      # z, y, x voxels of the 1 mm grid, constant
    out = rescale_displacement_field(field, moving, fixed, fixed_1mm)
    assert out.shape == moving.array.shape + (3,)
    A, o = moving.index_to_physical_matrix()
    centre = tuple(s // 2 for s in moving.array.shape)
    # constant field: rotation x, y, z -> (x, y, z) @ inv(I) @ Rz, then scaled by spacing ratio (1, 1, 1) / (0.5, 0.5, 2.5) in x, y, z
    v_xyz = np.array([0.5, -1.0, 2.0]) @ Rz
    expect_zyx = (v_xyz * (np.array([1.0, 1.0, 1.0]) / np.array([0.5, 0.5, 2.5])))[::-1]
    assert np.allclose(out[centre], expect_zyx)
    # translation helpers
    assert np.allclose(index_translation_to_world_translation((1.0, 2.0, 3.0), Rz.reshape(-1)), Rz @ np.array([1.0, 2.0, 3.0]))
    moved = apply_translation(moving, (1.0, 2.0, 3.0))
    assert np.allclose(np.array(moving.GetOrigin()) - np.array(moved.GetOrigin()), Rz @ np.array([1.0, 2.0, 3.0])) and moving.GetOrigin() == (6.0, 0.0, 0.0)
    assert field_to_translation(field, (0.5, 0.5, 2.5)) == (0.5, -1.0, 2.5)            # whole voxels of the moving image, in mm (x, y, z)


def test_torch_full_sum_restatement():
    """oracle.torch_sum = ATen's float32 sum of a whole contiguous tensor for a given thread count (two-pass reduction, 8-float vectors,
    4 interleaved cascade accumulators): the one quantity of the hot path whose bits depend on the reference's thread count
    (`mind_var.mean()`, convex_adam_utils.py:61).  Pinned against torch.sum itself."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    orc.build()
    rng = np.random.default_rng(1)
    old = torch.get_num_threads()
    try:
        for T in (1, 2, 3, 8, 16):
            torch.set_num_threads(T)
            for n in (1, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65, 1000, 4097, 32767, 32768, 32769, 65537, 100000, 262147, 1000003):
                x = (rng.random(n, dtype=np.float32) ** 4 * 1e3).astype(np.float32)
                assert float(torch.from_numpy(x).sum()) == orc.torch_sum(x, T), (T, n)
    finally:
        torch.set_num_threads(old)


def test_pinned_pool_marks_entries_busy_under_a_lock():
    """ADVICE round 4 (convex_adam_MIND._PinnedPool): take() marks an entry BUSY inside the lock, give() installs the weak reference to the
    array handed out (or frees the entry); a second take() while the first caller still holds the entry gets another one.  The pool logic
    is exercised here with ordinary host tensors (pinning needs a device)."""
    import gc
    import numpy as np
    import torch
    from convexadam_amd import convex_adam_MIND as M
    pool = M._PinnedPool()
    real_empty = torch.empty
    try:
        torch.empty = lambda *a, pin_memory=False, **k: real_empty(*a, **k)        # no device here: the pool's bookkeeping is what is tested
        e1, b1 = pool.take((4, 5), torch.float64)
        e2, b2 = pool.take((4, 5), torch.float64)                                 # e1 is BUSY: a different entry
        assert e1 is not e2 and b1.data_ptr() != b2.data_ptr() and e1[1] is M._BUSY and e2[1] is M._BUSY
        a1 = b1.numpy()
        pool.give(e1, a1)
        e3, b3 = pool.take((4, 5), torch.float64)                                 # a1 is alive, e2 busy: a third entry
        assert e3 is not e1 and e3 is not e2
        pool.give(e2, None)                                                       # freed without a result (an exception path)
        e4, _ = pool.take((2, 5), torch.float64)
        assert e4 is e2                                                           # the free entry is reused (a smaller request fits)
        del a1
        gc.collect()
        e5, _ = pool.take((4, 5), torch.float64)
        assert e5 is e1                                                           # the result was garbage-collected: its buffer returns
        pool.clear()
        assert all(e[1] is M._BUSY for e in pool._entries)                        # busy entries survive a clear()
    finally:
        torch.empty = real_empty
