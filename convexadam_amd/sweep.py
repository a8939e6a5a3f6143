"""Sharded sweep driver: (hyper-parameter setting x image pair) items over the GPUs of one node.

Counterpart of the reference's manual scheme -- one OS process per GPU started by hand with a GPU id
(self_configuring/convex_run_withconfig.py:42-43,178-180) -- done properly: one process per GPU under
`torch.distributed.run`, items assigned round-robin by rank, NO collective on the data path (a pair is
~20 GB of HBM traffic and never worth splitting over xGMI); torch.distributed is used only
for the start barrier and the final gather of per-item results on rank 0 (RCCL on GPUs, gloo on CPU).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        convexadam_amd/sweep.py --pairs 4 --settings 256 --shape 160 192 224 --evaluate --out sweep.json

With --evaluate every item is scored on the device the way the reference's sweep scripts do it
(convex_run_withconfig.py:136-150, convex_run_paired_mind.py:165-177): Jacobian log-std and folding fraction,
nearest-neighbour warp of the moving label map and Dice against the fixed one, key-point TRE; only those scalars
return to the host, and rank 0 aggregates the per-setting means into the reference's geometric-mean rank
(convex_run_withconfig.py:160-168, sort_rank of hyper_util:28-31).
"""
import argparse
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


def shard_items(items, rank, world_size):
    """Static round-robin assignment (item i -> rank i mod world_size)."""
    return list(items)[rank::world_size]


def sweep_settings():
    """The build's own 256-point grid for BASELINE config 5 (the reference draws 100 + 75 random settings
    from torch.manual_seed(1004)/(2004), convex_run_withconfig.py:65-69; a fixed grid is reproducible
    without torch's RNG stream): 4 MIND shapes x 4 grid spacings x 4 search half-widths x 4 lambdas."""
    out = []
    for (r, d), gs, hw, lam in itertools.product(((1, 1), (1, 2), (2, 1), (2, 2)), (4, 5, 6, 8), (3, 4, 5, 6),
                                                 (0.75, 1.0, 1.25, 1.5)):
        out.append(dict(mind_r=r, mind_d=d, grid_sp=gs, disp_hw=hw, lambda_weight=lam, grid_sp_adam=2,
                        selected_niter=80, ic=True))
    return out


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


def evaluate_item(disp, seg_fixed, seg_moving, key_fixed, key_moving, num_labels):
    """disp (3,H,W,D) device field in voxels -> dict of the reference's evaluation scalars (computed on the device)."""
    from convexadam_amd import convexAdam_hyper_util as HU
    d = disp[None]
    jac = HU.jacobian_determinant_3d(d, False)                                             # convex_run_withconfig.py:137
    jstd, fold = HU.jacobian_log_std_and_folding(jac)                                      # :148-150
    warped = HU.warp_labels_nearest(seg_moving, d)                                         # :141
    dice = HU.dice_coeff(seg_fixed, warped, num_labels + 1)                                # :142
    dice0 = HU.dice_coeff(seg_fixed, seg_moving, num_labels + 1)
    tre, _ = HU.tre_at_keypoints(d, key_fixed, key_moving)                                 # convex_run_paired_mind.py:165-173
    tre0 = (key_fixed - key_moving).square().sum(-1).sqrt()
    hd95 = HU.cupy_hd95(seg_fixed, warped, num_labels)                                     # convex_run_withconfig.py:143
    return dict(dice=float(dice.mean()), dice_before=float(dice0.mean()), jstd=jstd, folding=fold, tre=float(tre.mean()),
                tre_before=float(tre0.mean()), hd95=float(hd95.mean()))


def aggregate_ranks(results, n_settings):
    """Per-setting means over pairs and the reference's rank: prod(sort_rank(metric)) ** (1/n) (convex_run_withconfig.py:160-168)."""
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


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--settings", type=int, default=4, help="use the first N settings of the 256-point grid")
    ap.add_argument("--shape", type=int, nargs=3, default=[64, 64, 64])
    ap.add_argument("--niter", type=int, default=None, help="override selected_niter")
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--dry-run", action="store_true", help="no kernels: exercises sharding + gather only (CPU/gloo)")
    ap.add_argument("--evaluate", action="store_true", help="score every item on the device (Dice, Jacobian, TRE) and rank the settings")
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

    settings = sweep_settings()[: a.settings]
    items = [(s, p) for s in range(len(settings)) for p in range(a.pairs)]
    mine = shard_items(list(enumerate(items)), rank, world)

    results = []
    pair_cache = {}
    label_cache = {}
    if world > 1:
        dist.barrier()
    t0 = time.time()
    for item_id, (s, p) in mine:
        cfg = dict(settings[s])
        if a.niter is not None:
            cfg["selected_niter"] = a.niter
        if a.dry_run:
            results.append(dict(item=item_id, setting=s, pair=p, rank=rank, ms=0.0, mean_abs_disp=0.0))
            continue
        from convexadam_amd.convex_adam_MIND import register_pair_device
        if p not in pair_cache:
            pair_cache[p] = _make_pair(tuple(a.shape), p, device)
        fix, mov = pair_cache[p]
        torch.cuda.synchronize(device)
        t1 = time.time()
        disp = register_pair_device(fix, mov, **cfg)
        torch.cuda.synchronize(device)
        res = dict(item=item_id, setting=s, pair=p, rank=rank, ms=(time.time() - t1) * 1e3, mean_abs_disp=float(disp.abs().mean()))
        if a.evaluate:
            if p not in label_cache:
                label_cache[p] = _make_labels(tuple(a.shape), p, device)
            res.update(evaluate_item(disp, *label_cache[p]))
        results.append(res)
    if use_gpu:
        torch.cuda.synchronize(device)
    elapsed = time.time() - t0

    gathered = [None] * world
    if world > 1:
        dist.all_gather_object(gathered, dict(rank=rank, results=results, elapsed=elapsed))
    else:
        gathered = [dict(rank=rank, results=results, elapsed=elapsed)]
    if rank == 0:
        allres = sorted((r for g in gathered for r in g["results"]), key=lambda r: r["item"])
        wall = max(g["elapsed"] for g in gathered)
        summary = dict(world_size=world, n_items=len(items), items_done=[r["item"] for r in allres],
                       per_rank={str(g["rank"]): [r["item"] for r in g["results"]] for g in gathered},
                       wall_s=wall, items_per_s=(len(items) / wall if wall > 0 else None), results=allres)
        if a.evaluate and not a.dry_run:
            summary["ranking"] = aggregate_ranks(allres, len(settings))
        txt = json.dumps(summary)
        if a.out:
            with open(a.out, "w") as f:
                f.write(txt)
        print(json.dumps({k: v for k, v in summary.items() if k not in ("results",)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
