"""Self-configuring sweep, sharded over the GPUs of one node (BASELINE configs[4]).

The reference tunes convexAdam in two stages (self_configuring/convex_run_withconfig.py, adam_run_withconfig_shiftSpline.py and their
`*_paired_mind*` twins), each started by hand once per GPU id (:42-43,178-180):

  stage 1  convex only: N1 random settings (MIND radius / dilation, grid_sp, disp_hw) x all validation pairs; every field is scored
           (Dice, Dice of the 30 % hardest labels, HD95, std of the log-Jacobian) and the settings are ranked by the geometric mean of
           their per-metric ranks (convex_run_withconfig.py:63-172)
  stage 2  Adam: with the best stage-1 setting, N2 random settings (grid_sp_adam, smoother of the control grid, lambda); ONE
           120-iteration Adam run per (setting, pair) is scored 16 times -- disp_sample after iterations 60 / 80 / 100 / 120, each
           after 0..3 extra 3^3 mean filters at full resolution -- and (setting, snapshot, smoothing) triples are ranked the same way
           (adam_run_withconfig_shiftSpline.py:143-266)

Here: one process per GPU under `torch.distributed.run`; items = (setting, pair), handed out by a SHARED WORK QUEUE (an atomic counter in
a TCPStore on rank 0; items are queued most-expensive-first, so the slowest rank never ends up with the big cost volumes), no collective
on the data path; every finished item is appended to `<out>.rank<r>.jsonl` at once, so a killed sweep restarts with
`--resume` and skips what is on disk (the reference saves after every setting, convex_run_withconfig.py:156,172); all scoring runs on the
device (csrc/metrics.hip, edt.hip).  torch.distributed is used for the phase barriers and the hand-over of the stage-1 winner only.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        convexadam_amd/sweep.py --pairs 4 --shape 160 192 224 --stage1 100 --stage2 75 --out sweep.json

`--settings N --evaluate` keeps the round-1 behaviour (the build's own 256-point grid, whole pipeline per item).
"""
import argparse
import contextlib
import threading
import glob
import itertools
import json
import os
import sys
import time

import torch
import torch.distributed as dist

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

SNAP_ITERS = (60, 80, 100, 120)          # disp_sample after these iterations (adam_run_withconfig_shiftSpline.py:234: iter 59, 79, ...)
N_EXTRA_SMOOTH = 4                       # 0..3 extra 3^3 mean filters (:244-246)
ADAM_ITERS = 120


# ---- settings ------------------------------------------------------------------------------------------------------------------
def sweep_settings():
    """The build's own 256-point grid (round-1 mode): 4 MIND shapes x 4 grid spacings x 4 search half-widths x 4 lambdas."""
    out = []
    for (r, d), gs, hw, lam in itertools.product(((1, 1), (1, 2), (2, 1), (2, 2)), (4, 5, 6, 8), (3, 4, 5, 6),
                                                 (0.75, 1.0, 1.25, 1.5)):
        out.append(dict(mind_r=r, mind_d=d, grid_sp=gs, disp_hw=hw, lambda_weight=lam, grid_sp_adam=2,
                        selected_niter=80, ic=True))
    return out


def stage1_settings(n, seed=1004):
    """n random convex-stage settings, drawn like the reference's (torch.manual_seed(1004), convex_run_paired_mind.py): MIND radius
    1..3, dilation 1..3, grid_sp 2..5, disp_hw 2..7, the search capped at 5 for grid_sp 2 (its cost volume would not fit)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 4, generator=g)
    out = []
    for a, b, c, e in u.tolist():
        gs = 2 + int(c * 4)
        hw = 2 + int(e * 6)
        if gs == 2:
            hw = min(hw, 5)
        out.append(dict(mind_r=1 + int(a * 3), mind_d=1 + int(b * 3), grid_sp=gs, disp_hw=hw))
    return out


def stage2_settings(n, seed=2004):
    """n random Adam-stage settings (torch.manual_seed(2004), adam_run_withconfig_shiftSpline.py:143-171): grid_sp_adam 1..4, smoother
    index 1..5 shifted by +2 / +1 for grid_sp_adam 1 / 2 (the finer the control grid, the wider the spline), lambda 0.4 .. 1.6."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 3, generator=g)
    out = []
    for a, b, c in u.tolist():
        gsa = 1 + int(a * 4)
        avg = 1 + int(b * 5) + (2 if gsa == 1 else (1 if gsa == 2 else 0))
        out.append(dict(grid_sp_adam=gsa, avg_n=min(avg, 7), lambda_weight=round(0.2 * (2 + int(c * 7)), 1)))
    return out


def smoother_table():
    """avgs of adam_run_withconfig_shiftSpline.py:140-141."""
    from convexadam_amd import convexAdam_hyper_util as HU
    return [HU.GaussianSmoothing(0.7), HU.GaussianSmoothing(1.0)] + [HU.kovesi_spline(s, 4) for s in (1.3, 1.6, 1.9, 2.2, 2.5, 2.8)]


def item_cost(cfg, shape):
    """Rough relative cost of one item (cost-volume bytes + Adam work): orders the queue, nothing else."""
    gs, hw = cfg.get("grid_sp", 6), cfg.get("disp_hw", 4)
    v = (shape[0] // gs) * (shape[1] // gs) * (shape[2] // gs)
    gsa = cfg.get("grid_sp_adam", 2)
    return (2 * hw + 1) ** 3 * v * 8 + ADAM_ITERS * 12 * (shape[0] // gsa) * (shape[1] // gsa) * (shape[2] // gsa) * 40 * (1 if "avg_n" in cfg or cfg.get("lambda_weight", 0) > 0 else 0)


# ---- synthetic validation data -------------------------------------------------------------------------------------------------
SHIFT = (2, -1, 3)          # moving = fixed content rolled by SHIFT voxels: the field to recover is +SHIFT


def _make_pair(shape, idx, device):
    from convexadam_amd.phantom import phantom
    fix = phantom(shape, 100 + idx, 200 + idx)
    mov = torch.roll(phantom(shape, 100 + idx, 300 + idx), SHIFT, (0, 1, 2))
    return fix.to(device), mov.to(device)


def _make_labels(shape, idx, device, num_labels=13):
    """Synthetic anatomy: argmax over smooth random fields, moving = rolled copy; 32 key points."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(500 + idx)
    lab = F.interpolate(torch.randn(1, num_labels + 1, *[max(2, s // 8) for s in shape], generator=g), size=shape, mode="trilinear",
                        align_corners=False).argmax(1)[0].float()
    key_f = torch.rand(32, 3, generator=g) * (torch.tensor([float(s - 9) for s in shape])) + 4.0
    key_m = key_f + torch.tensor([float(v) for v in SHIFT])
    return lab.to(device), torch.roll(lab, SHIFT, (0, 1, 2)).to(device), key_f, key_m, num_labels


# ---- scoring ----------------------------------------------------------------------------------------------------------------------
def evaluate_item(disp, seg_fixed, seg_moving, key_fixed, key_moving, num_labels, robust=None, cache=None):
    """disp (3,H,W,D) device field in voxels -> dict of the reference's evaluation scalars (computed on the device)."""
    from convexadam_amd import convexAdam_hyper_util as HU
    d = disp[None]
    jac = HU.jacobian_determinant_3d(d, False)                                             # convex_run_withconfig.py:137
    jstd, fold = HU.jacobian_log_std_and_folding(jac)                                      # :148-150
    warped = HU.warp_labels_nearest(seg_moving, d)                                         # :141
    counts = HU.label_overlap_counts(seg_fixed, warped, num_labels + 1)                    # shared by Dice and HD95
    dice = HU.dice_coeff(seg_fixed, warped, num_labels + 1, counts=counts)                 # :142
    # what does not depend on the field is computed once per pair (kept in the caller's per-pair cache)
    before = cache.get("before") if cache is not None else None
    if before is None:
        dice0 = HU.dice_coeff(seg_fixed, seg_moving, num_labels + 1)
        tre0 = (key_fixed - key_moving).square().sum(-1).sqrt()
        before = (dice0, tre0)
        if cache is not None:
            cache["before"] = before
    dice0, tre0 = before
    tre, _ = HU.tre_at_keypoints(d, key_fixed, key_moving)                                 # convex_run_paired_mind.py:165-173
    hd95 = HU.cupy_hd95(seg_fixed, warped, num_labels, fixed_cache=cache, counts=counts)   # convex_run_withconfig.py:143 (cache: the fixed map's bit planes)
    if robust is None:                                                                      # the 30 % labels with the lowest initial overlap (:60-61)
        robust = dice0.topk(max(1, int(num_labels * 0.3)), largest=False).indices
    return dict(dice=float(dice.mean()), dice30=float(dice[robust].mean()), dice_before=float(dice0.mean()), jstd=jstd, folding=fold,
                tre=float(tre.mean()), tre_before=float(tre0.mean()), hd95=float(hd95.mean()))


def rank_records(records, keys):
    """Geometric mean of the per-metric ranks over the groups identified by `keys` (means over pairs first)
    (convex_run_withconfig.py:160-168, adam_run_withconfig_shiftSpline.py:259-264; sort_rank of hyper_util:28-31)."""
    from convexadam_amd.convexAdam_hyper_util import sort_rank
    groups = {}
    for r in records:
        groups.setdefault(tuple(r[k] for k in keys), []).append(r)
    ids = sorted(groups)
    mean = lambda k: torch.tensor([sum(x[k] for x in groups[i]) / len(groups[i]) for i in ids])   # noqa: E731
    dice, dice30, jstd, hd95, tre = mean("dice"), mean("dice30"), mean("jstd"), mean("hd95"), mean("tre")
    rank = (sort_rank(-dice) * sort_rank(-dice30) * sort_rank(jstd) * sort_rank(hd95)).pow(1 / 4)
    best = int(rank.argmax())
    return dict(ids=[list(i) for i in ids], dice=dice.tolist(), dice30=dice30.tolist(), jstd=jstd.tolist(), hd95=hd95.tolist(),
                tre=tre.tolist(), rank=rank.tolist(), best=list(ids[best]))


def aggregate_ranks(results, n_settings):
    """Round-1 mode: per-setting means and rank (Dice, TRE, log-Jacobian std, HD95)."""
    from convexadam_amd.convexAdam_hyper_util import sort_rank
    acc = {k: torch.zeros(n_settings) for k in ("dice", "jstd", "tre", "hd95")}
    cnt = torch.zeros(n_settings)
    for r in results:
        if "dice" not in r:
            continue
        for k in acc:
            acc[k][r["setting"]] += r[k]
        cnt[r["setting"]] += 1
    cnt = cnt.clamp(min=1)
    dice, jstd, tre, hd95 = acc["dice"] / cnt, acc["jstd"] / cnt, acc["tre"] / cnt, acc["hd95"] / cnt
    rank = (sort_rank(-dice) * sort_rank(tre) * sort_rank(jstd) * sort_rank(hd95)).pow(1 / 4)
    return dict(dice=dice.tolist(), jstd=jstd.tolist(), tre=tre.tolist(), hd95=hd95.tolist(), rank=rank.tolist(),
                best_setting=int(rank.argmax()))


# ---- sharding ----------------------------------------------------------------------------------------------------------------------
def shard_items(items, rank, world_size):
    """Static round-robin assignment (item i -> rank i mod world_size); kept for the dry-run and as the queue's fallback."""
    return list(items)[rank::world_size]


class WorkQueue:
    """Shared queue over the processes of a sweep: an atomic counter per phase in a TCPStore hosted by rank 0.  `order` lists the
    item ids most-expensive-first; every `next()` hands out one id until the phase is exhausted."""

    def __init__(self, rank, world):
        self.rank, self.world, self.store = rank, world, None
        self.lock = threading.Lock()
        if world > 1:
            port = int(os.environ.get("MASTER_PORT", "29500")) + 17
            self.store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, world, is_master=(rank == 0), wait_for_workers=True)
        self.local = {}

    def next(self, phase, order):
        with self.lock:                                                     # (several worker threads of one rank may draw)
            if self.store is not None:
                i = self.store.add("q_" + phase, 1) - 1
            else:
                i = self.local.get(phase, 0)
                self.local[phase] = i + 1
        return order[i] if i < len(order) else None


class ResultLog:
    """Append-only results of one rank (`<out>.rank<r>.jsonl`), keyed by (stage, setting, pair); `--resume` reads every rank's file.
    Every file starts with a header record describing the run (`run`: shape, pairs, setting counts, mode); on resume, files whose header
    differs from this run's are ignored (their records belong to another sweep), and a fresh start removes EVERY rank file of the
    output name -- also those of an earlier run with more ranks (rank 0 does it; the caller puts a barrier after construction)."""

    def __init__(self, out, rank, resume, run=None):
        self.path = "%s.rank%d.jsonl" % (out, rank) if out else None
        self.done = {}
        self.run = run
        self.ignored_files = []
        if out and resume:
            for f in sorted(glob.glob(out + ".rank*.jsonl")):
                recs, header = [], None
                for line in open(f):
                    line = line.strip()
                    if line:
                        try:
                            r = json.loads(line)
                        except ValueError:
                            continue                                        # a line cut off by the kill
                        if "header" in r:
                            header = r["header"]
                        else:
                            recs.append(r)
                if run is not None and header != run:
                    self.ignored_files.append(f)                            # another sweep's records (or a file without header)
                    continue
                for r in recs:
                    self.done[(r["stage"], r["setting"], r["pair"])] = r
            if self.path and (not os.path.exists(self.path) or self.path in self.ignored_files):
                self._write_header("w")
        elif out:
            if rank == 0:
                for f in glob.glob(out + ".rank*.jsonl"):
                    os.remove(f)

    def start(self):
        """Fresh run: called after the barrier that follows rank 0's clean-up."""
        if self.path and not os.path.exists(self.path):
            self._write_header("w")

    def _write_header(self, mode):
        with open(self.path, mode) as f:
            f.write(json.dumps(dict(header=self.run)) + "\n")
            f.flush()
            os.fsync(f.fileno())

    def add(self, rec):
        self.done[(rec["stage"], rec["setting"], rec["pair"])] = rec
        if self.path:
            with open(self.path, "a") as f:
                f.write(json.dumps(rec) + "\n")
                f.flush()
                os.fsync(f.fileno())


def gather_records(local, world):
    if world == 1:
        return list(local)
    got = [None] * world
    dist.all_gather_object(got, list(local))
    return [r for part in got for r in part]


# ---- the two stages ------------------------------------------------------------------------------------------------------------------
class PairData:
    """Validation pairs and their labels, made on demand and kept on the device (a few pairs per GPU fit easily in 288 GB)."""

    def __init__(self, shape, device):
        self.shape, self.device, self.pairs, self.labels, self.coarse, self.hd = tuple(shape), device, {}, {}, {}, {}
        self.lock = threading.Lock()

    def pair(self, p):
        with self.lock:
            if p not in self.pairs:
                self.pairs[p] = _make_pair(self.shape, p, self.device)
                if self.device.type == "cuda":
                    torch.cuda.current_stream(self.device).synchronize()       # made on this worker's stream, used on every worker's
            return self.pairs[p]

    def label(self, p):
        with self.lock:
            if p not in self.labels:
                self.labels[p] = _make_labels(self.shape, p, self.device)
                if self.device.type == "cuda":
                    torch.cuda.current_stream(self.device).synchronize()
            return self.labels[p]


def _hd_cache(data, p):
    """Per-pair cache of the fixed label map's distance transforms for HD95 (they do not depend on the field being scored); filled once,
    under the lock, and synchronised because the other workers read it on their own streams."""
    with data.lock:
        if p not in data.hd:
            from convexadam_amd import convexAdam_hyper_util as HU
            c = {}
            seg_f, seg_m, _, _, nl = data.labels[p]
            HU.cupy_hd95(seg_f, seg_m, nl, fixed_cache=c)
            torch.cuda.current_stream(data.device).synchronize()
            data.hd[p] = c
        return data.hd[p]


def run_stage1_item(cfg, p, data):
    from convexadam_amd.convex_adam_MIND import register_pair_device
    fix, mov = data.pair(p)
    torch.cuda.current_stream(data.device).synchronize()
    t = time.time()
    disp = register_pair_device(fix, mov, lambda_weight=0, ic=True, **cfg)                  # convex stage + inverse consistency (:100-128)
    torch.cuda.current_stream(data.device).synchronize()
    rec = dict(ms=(time.time() - t) * 1e3)
    lab = data.label(p)
    t = time.time()
    rec.update(evaluate_item(disp, *lab, cache=_hd_cache(data, p)))
    rec["eval_ms"] = (time.time() - t) * 1e3
    return rec


def run_stage2_item(best1, cfg2, p, data, smoothers, adam_mode="exact"):
    """One 120-iteration Adam run from the stage-1 field, scored at 4 iterations x 4 smoothings (adam_run_withconfig_shiftSpline.py:159-246)."""
    from convexadam_amd import convex_adam_utils as U
    from convexadam_amd.convex_adam_MIND import register_pair_device
    fix, mov = data.pair(p)
    H, W, D = data.shape
    key = (p, tuple(sorted(best1.items())))
    with data.lock:                                                                         # the convex stage runs once per pair (:100-126)
        if key not in data.coarse:
            data.coarse[key] = register_pair_device(fix, mov, lambda_weight=0, ic=True, **best1)
            torch.cuda.current_stream(data.device).synchronize()                           # (read on other workers' streams)
        disp_hr = data.coarse[key]
    gsa, lam = cfg2["grid_sp_adam"], cfg2["lambda_weight"]
    torch.cuda.current_stream(data.device).synchronize()
    t = time.time()
    ff = U.MINDSSC(fix[None, None], best1["mind_r"], best1["mind_d"], device=data.device)
    fm = U.MINDSSC(mov[None, None], best1["mind_r"], best1["mind_d"], device=data.device)
    F2, M2 = U.avg_pool(ff, gsa), U.avg_pool(fm, gsa)
    del ff, fm
    h2, w2, d2 = H // gsa, W // gsa, D // gsa
    P0 = U.resize_trilinear(disp_hr[None], (h2, w2, d2)) / float(gsa)
    n_ch = int(F2.shape[1])
    _, st = U.adam_run(F2, M2, P0, lam, ADAM_ITERS, smoother=smoothers[cfg2["avg_n"]], cost_scale=float(n_ch), snapshot_iters=SNAP_ITERS,
                       return_state=True, mode=adam_mode)
    torch.cuda.current_stream(data.device).synchronize()
    ms = (time.time() - t) * 1e3
    recs = []
    lab = data.label(p)
    hdc = _hd_cache(data, p)
    t_eval = time.time()
    for ii in range(len(SNAP_ITERS)):
        field = U.resize_trilinear(st["snapshots"][ii][None] * float(gsa), (H, W, D))
        for kk in range(N_EXTRA_SMOOTH):
            if kk > 0:
                field = U.box_smooth(field, 3, 1)
            r = dict(snap=ii, smooth=kk, ms=ms)
            r.update(evaluate_item(field[0], *lab, cache=hdc))
            recs.append(r)
    eval_ms = (time.time() - t_eval) * 1e3
    for r in recs:
        r["eval_ms"] = eval_ms / len(recs)
    return recs


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--shape", type=int, nargs=3, default=[64, 64, 64])
    ap.add_argument("--stage1", type=int, default=0, help="number of random convex-stage settings (two-stage mode)")
    ap.add_argument("--stage2", type=int, default=0, help="number of random Adam-stage settings (two-stage mode)")
    ap.add_argument("--settings", type=int, default=4, help="round-1 mode: first N settings of the 256-point grid, whole pipeline per item")
    ap.add_argument("--niter", type=int, default=None, help="round-1 mode: override selected_niter")
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--resume", action="store_true", help="skip the items already present in <out>.rank*.jsonl")
    ap.add_argument("--static", action="store_true", help="round-robin assignment instead of the shared queue")
    ap.add_argument("--dry-run", action="store_true", help="no kernels: exercises queue, logs, resume and gather only (CPU/gloo)")
    ap.add_argument("--evaluate", action="store_true", help="round-1 mode: score every item on the device and rank the settings")
    ap.add_argument("--adam-mode", default="fast", choices=("exact", "fast", "fast_all"),
                    help="arithmetic of the stage-2 Adam runs: exact = the reference's evaluation order; fast (default) = throughput arithmetic for the warp "
                         "gradient and the adjoint smoother, forward smoother / regulariser / update in the reference's order; fast_all = separable forward "
                         "smoother too (fastest; further from the reference's fields -- offered because the sweep grades by overlap scores, which agreed to "
                         "three digits between the modes on the synthetic label maps, not the default since round 5)")
    ap.add_argument("--workers", type=int, default=0, help="items in flight per rank, each on its own thread and HIP stream (0 = automatic = 3: the host side of "
                    "an item -- two synchronisations per evaluation, Python between the launches -- is a fifth of its time since the evaluation kernels "
                    "got short; measured on the 192-item example, alternating runs: 38-42 items/s with one worker, 43-44 with two, 45-55 with three or four)")
    a = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    use_gpu = torch.cuda.is_available() and not a.dry_run
    if world > 1:
        dist.init_process_group(backend="nccl" if use_gpu else "gloo", rank=rank, world_size=world)
    if use_gpu:
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    two_stage = a.stage1 > 0
    n_workers = a.workers if a.workers > 0 else (3 if use_gpu else 1)
    queue = WorkQueue(rank, world)
    run_id = dict(shape=list(a.shape), pairs=a.pairs, stage1=a.stage1, stage2=a.stage2, settings=a.settings, niter=a.niter, evaluate=bool(a.evaluate),
                  dry_run=bool(a.dry_run), adam_mode=a.adam_mode)
    log = ResultLog(a.out, rank, a.resume, run_id)
    if world > 1:
        dist.barrier()                                                      # rank 0 has removed the files of an earlier run
    log.start()
    data = PairData(a.shape, device)
    shape = tuple(a.shape)

    def run_phase(phase, settings, worker):
        """Hands out (setting, pair) items of one phase; returns this rank's records (resumed ones included)."""
        items = [(s, p) for s in range(len(settings)) for p in range(a.pairs)]
        order = sorted(range(len(items)), key=lambda i: -item_cost(settings[items[i][0]], shape))
        mine, fresh = [], [0]
        static = shard_items(order, rank, world) if a.static else None
        static_pos = [0]
        lock = threading.Lock()

        def draw():
            if static is None:
                return queue.next(phase, order)
            with lock:
                k = static_pos[0]
                static_pos[0] = k + 1
            return static[k] if k < len(static) else None

        def loop(widx):
            stream = torch.cuda.Stream(device) if use_gpu and n_workers > 1 else None
            ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
            with ctx:
                while True:
                    i = draw()
                    if i is None:
                        break
                    s, p = items[i]
                    if (phase, s, p) in log.done:                           # --resume: already on disk; reported by whoever draws it
                        with lock:
                            mine.append(dict(log.done[(phase, s, p)], item=i))
                        continue
                    rec = dict(stage=phase, setting=s, pair=p, rank=rank, item=i, worker=widx)
                    try:
                        if os.environ.get("CVX_SWEEP_FAIL_ITEM") == str(i):
                            raise RuntimeError("injected failure (test hook CVX_SWEEP_FAIL_ITEM)")
                        rec.update(worker(settings[s], p))
                    except Exception as e:                                  # keep drawing: the other ranks wait in the gather that follows
                        with lock:
                            failures.append(dict(stage=phase, setting=s, pair=p, rank=rank, item=i, error="%s: %s" % (type(e).__name__, e)))
                        continue
                    with lock:
                        log.add(rec)
                        mine.append(rec)
                        fresh[0] += 1
                if stream is not None:
                    stream.synchronize()

        if n_workers > 1:
            ths = [threading.Thread(target=loop, args=(k,)) for k in range(n_workers)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        else:
            loop(0)
        return mine, fresh[0]

    failures = []

    def check_failures():
        """After a phase's gather: every rank learns about every failed item and the run stops with one message."""
        allf = gather_records(failures, world)
        if allf:
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            raise RuntimeError("sweep: %d item(s) failed, e.g. %s" % (len(allf), json.dumps(allf[0])))

    if use_gpu:                                                             # the validation pairs are resident before the clock starts
        for p in range(a.pairs):
            data.pair(p)
            data.label(p)
        torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    t0 = time.time()
    summary = dict(world_size=world, shape=list(shape), pairs=a.pairs, workers_per_rank=n_workers, adam_mode=a.adam_mode)
    if two_stage:
        s1 = stage1_settings(a.stage1)
        if a.dry_run:
            w1 = lambda cfg, p: dict(ms=0.0, dice=0.5 + 0.001 * cfg["grid_sp"], dice30=0.4, jstd=0.1, hd95=3.0, tre=1.0, folding=0.0)   # noqa: E731
        else:
            w1 = lambda cfg, p: run_stage1_item(cfg, p, data)                                                                          # noqa: E731
        mine1, fresh1 = run_phase("convex", s1, w1)
        all1 = gather_records(mine1, world)
        check_failures()
        seen = {}
        for r in all1:
            seen[(r["setting"], r["pair"])] = r
        rank1 = rank_records(list(seen.values()), ("setting",))
        best1 = s1[rank1["best"][0]]
        summary.update(stage1=dict(n_settings=len(s1), n_items=len(seen), fresh_items=sum(gather_records([fresh1], world)), ranking=rank1,
                                   best_setting=best1, per_rank={str(r): sorted(x["item"] for x in all1 if x["rank"] == r) for r in range(world)}))
        if a.stage2 > 0:
            s2 = stage2_settings(a.stage2)
            smoothers = None if a.dry_run else smoother_table()
            if a.dry_run:
                w2 = lambda cfg, p: dict(evals=[dict(snap=i, smooth=k, dice=0.6 + 0.01 * i - 0.001 * k, dice30=0.5, jstd=0.1 + 0.01 * k, hd95=2.5, tre=0.8, folding=0.0, ms=0.0)   # noqa: E731
                                                for i in range(len(SNAP_ITERS)) for k in range(N_EXTRA_SMOOTH)])
            else:
                w2 = lambda cfg, p: dict(evals=run_stage2_item(best1, cfg, p, data, smoothers, a.adam_mode))                                     # noqa: E731
            mine2, fresh2 = run_phase("adam", s2, w2)
            all2 = gather_records(mine2, world)
            check_failures()
            flat, seen2 = [], set()
            for r in all2:
                if (r["setting"], r["pair"]) in seen2:
                    continue
                seen2.add((r["setting"], r["pair"]))
                for e in r["evals"]:
                    flat.append(dict(e, setting=r["setting"], pair=r["pair"]))
            rank2 = rank_records(flat, ("setting", "snap", "smooth"))
            b = rank2["best"]
            summary.update(stage2=dict(n_settings=len(s2), n_items=len(seen2), fresh_items=sum(gather_records([fresh2], world)), evaluations=len(flat), ranking=dict(best=b, best_rank=max(rank2["rank"])),
                                       best_setting=dict(s2[b[0]], selected_niter=SNAP_ITERS[b[1]], extra_smooth=b[2])))
    else:
        settings = sweep_settings()[: a.settings]

        def w0(cfg, p):
            cfg = dict(cfg)
            if a.niter is not None:
                cfg["selected_niter"] = a.niter
            if a.dry_run:
                return dict(ms=0.0, mean_abs_disp=0.0)
            from convexadam_amd.convex_adam_MIND import register_pair_device
            fix, mov = data.pair(p)
            torch.cuda.current_stream(device).synchronize()
            t1 = time.time()
            disp = register_pair_device(fix, mov, **cfg)
            torch.cuda.current_stream(device).synchronize()
            res = dict(ms=(time.time() - t1) * 1e3, mean_abs_disp=float(disp.abs().mean()))
            if a.evaluate:
                res.update(evaluate_item(disp, *data.label(p)))
            return res

        mine, _ = run_phase("pipeline", settings, w0)
        allres = sorted(gather_records(mine, world), key=lambda r: r["item"])
        check_failures()
        summary.update(n_items=len(settings) * a.pairs, items_done=sorted(r["item"] for r in allres),
                       per_rank={str(r): sorted(x["item"] for x in allres if x["rank"] == r) for r in range(world)}, results=allres)
        if a.evaluate and not a.dry_run:
            summary["ranking"] = aggregate_ranks(allres, len(settings))
    if use_gpu:
        torch.cuda.synchronize(device)
    elapsed = torch.tensor([time.time() - t0], dtype=torch.float64)
    if world > 1:
        el = [None] * world
        dist.all_gather_object(el, float(elapsed))
        elapsed = torch.tensor([max(el)])
    if rank == 0:
        summary["wall_s"] = float(elapsed)
        n_items = summary.get("n_items") or (summary.get("stage1", {}).get("n_items", 0) + summary.get("stage2", {}).get("n_items", 0))
        summary["items_per_s"] = n_items / summary["wall_s"] if summary["wall_s"] > 0 else None
        if two_stage and not a.dry_run:
            # where the wall time went (summed over workers and ranks; two workers per rank overlap, so the parts may exceed the wall)
            reg1 = sum(r.get("ms", 0.0) for r in seen.values())
            ev1 = sum(r.get("eval_ms", 0.0) for r in seen.values())
            ph = dict(stage1_registration_s=reg1 / 1e3, stage1_evaluation_s=ev1 / 1e3)
            if "stage2" in summary:
                items2 = {}
                for e in flat:
                    items2.setdefault((e["setting"], e["pair"]), []).append(e)
                ph.update(stage2_mind_adam_s=sum(v[0].get("ms", 0.0) for v in items2.values()) / 1e3,
                          stage2_evaluation_s=sum(e.get("eval_ms", 0.0) for e in flat) / 1e3, stage2_evaluations=len(flat))
            ph["other_s"] = summary["wall_s"] * n_workers - sum(v for k, v in ph.items() if k.endswith("_s"))
            ph["note"] = "seconds summed over the workers of all ranks (workers_per_rank overlap on one GPU); other = queue, logging, Python, idle"
            summary["phases"] = ph
        if a.out:
            with open(a.out, "w") as f:
                f.write(json.dumps(summary))
        print(json.dumps({k: v for k, v in summary.items() if k not in ("results",)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
