#!/usr/bin/env python
"""Turns the output of tools/profile_round.sh (gpurun_out/round) into the committed files under profiles/:
   python tools/make_profiles.py gpurun_out/round r01_v4
 - <tag>_bench_kernel_stats.txt   per-kernel durations of `bench.py --steps 5 --warmup 1` (rocprofv3 --kernel-trace --stats)
 - <tag>_pmc_hbm_traffic.txt      FETCH_SIZE / WRITE_SIZE per kernel and the corrected HBM bytes per launch
 - <tag>_pmc_sq_counters.txt      SQ counters per kernel
 - pmc_hbm_traffic.json           bytes per launch of the correlation stage (read by bench.py for roofline.traffic)
 - <tag>_bench_line.json          the bench line of the same box
FETCH_SIZE counts half of the bytes of a 16-byte-per-lane streamed read on gfx950 (calibrated on k_argmin4: 270.5 MB streamed -> factor
2.00) and ALL bytes of kernels that read shorter contiguous pieces (k_mind_finish_pool: 330.3 MB read, 330.5 MB counted -> factor 1.00);
WRITE_SIZE is exact; both are in KiB.  bytes = (factor * FETCH_SIZE + WRITE_SIZE) * 1024 with the PER-KERNEL factor of FETCH_FACTOR
(calibrated where the read volume is known, 2.0 -- an upper bound -- elsewhere; the table prints the factor it used and the factor-1 figure)."""

# FETCH_SIZE correction per kernel (substring match): measured = known read bytes / counted bytes on the benchmark configuration
# (defaults; main() replaces them by the factors calibrated ON THE RUN BEING PROCESSED wherever a kernel of KNOWN_READS was launched)
FETCH_FACTOR = {"k_mind_finish_pool": 1.0, "k_argmin4": 2.0, "k_to_chunked": 2.0}
DEFAULT_FACTOR = 2.0
V_FULL, V_COARSE, K_DISP = 160 * 192 * 224, 26 * 32 * 37, 2197
# read volume of kernels whose input is streamed exactly once (benchmark configuration), bytes per launch
KNOWN_READS = {"k_argmin4": K_DISP * V_COARSE * 4, "k_cert_plain_stream": K_DISP * V_COARSE * 4, "k_mind_finish_pool": 12 * V_FULL * 4,
               "k_to_chunked": 12 * (V_FULL // 8) * 4}


def fetch_factor(kernel):
    for name, f in FETCH_FACTOR.items():
        if name in kernel:
            return f
    return DEFAULT_FACTOR
import collections
import csv
import glob
import io
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def pmc(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    return acc, cnt


def corr_sha():
    """sha256 over the correlation-stage sources (bench.py compares it with the tree it runs on -> roofline.traffic_stale)."""
    import hashlib                              # (same recipe as bench.py::corr_sources_sha)
    h = hashlib.sha256()
    for name in ("corrfused.hip", "correlate.hip", "corrbox.hip", "corrcert.hip", "certify.hip"):
        with open(os.path.join(ROOT, "convexadam_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main(src, tag):
    prof = os.path.join(ROOT, "profiles")
    db = glob.glob(src + "/stats/**/*_results.db", recursive=True)
    if db:
        out = subprocess.run([sys.executable, os.path.join(HERE, "rocpd_stats.py"), db[0]], stdout=subprocess.PIPE, text=True).stdout
        with open(os.path.join(prof, tag + "_bench_kernel_stats.txt"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batched   (6 pairs, package default mode = exact)\n" + out)
    db = glob.glob(src + "/stats_fast/**/*_results.db", recursive=True)
    if db:
        out = subprocess.run([sys.executable, os.path.join(HERE, "rocpd_stats.py"), db[0]], stdout=subprocess.PIPE, text=True).stdout
        with open(os.path.join(prof, tag + "_fast_mode_bench_kernel_stats.txt"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batched --adam-mode fast   (6 pairs, opt-in fast Adam mode)\n" + out)
    fa, fc = pmc(src + "/fetch")
    wa, wc = pmc(src + "/write")
    # calibrate the FETCH_SIZE factor on this very run: known read bytes / counted bytes, clamped to [1, 2] (the counter tallies a 128-byte
    # line fetched by a 16-byte-per-lane access as 64 B and shorter pieces 1:1; a ratio outside that range means the kernel did not stream
    # what KNOWN_READS assumes -- e.g. another configuration -- and the default stays)
    calibrated = {}
    for k in fa:
        for name, nbytes in KNOWN_READS.items():
            if name in k and fc[k].get("FETCH_SIZE"):
                counted = fa[k]["FETCH_SIZE"] / fc[k]["FETCH_SIZE"] * 1024
                if counted > 0 and 0.9 <= nbytes / counted <= 2.2:
                    calibrated[name] = min(2.0, max(1.0, nbytes / counted))
    FETCH_FACTOR.update(calibrated)
    rows = []
    for k in sorted(set(fa) | set(wa)):
        n = max(fc[k].get("FETCH_SIZE", 0), wc[k].get("WRITE_SIZE", 0))
        fe = fa[k].get("FETCH_SIZE", 0.0) / max(fc[k].get("FETCH_SIZE", 1), 1)
        wr = wa[k].get("WRITE_SIZE", 0.0) / max(wc[k].get("WRITE_SIZE", 1), 1)
        rows.append((k, n, fe, wr, (fetch_factor(k) * fe + wr) * 1024 / 1e6))
    rows.sort(key=lambda r: -r[4] * r[1])
    buf = io.StringIO()
    buf.write("# HBM traffic per launch from rocprofv3 PMC counters (two separate passes, MI355X_MICROARCH.md section HBM):\n"
              "#   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched\n"
              "#   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched\n"
              "# Counters are in KiB; WRITE_SIZE is exact, FETCH_SIZE reads 1/2 of a 16-byte-per-lane streamed read on gfx950 and 1/1 of shorter pieces ->\n"
              "# bytes = (factor*FETCH_SIZE + WRITE_SIZE) * 1024 with the per-kernel factor shown (calibrated on THIS run where the read volume is known: "
              + ", ".join("%s %.2f" % kv for kv in sorted(calibrated.items())) + ";\n"
              "# 2.00 = upper bound elsewhere; the last column is the factor-1 lower bound).\n")
    buf.write("%-44s %6s %14s %14s %6s %18s %14s\n" % ("kernel", "calls", "FETCH_KiB/call", "WRITE_KiB/call", "factor", "corrected_MB/call", "factor1_MB/call"))
    for k, n, fe, wr, mb in rows:
        buf.write("%-44s %6d %14.1f %14.1f %6.2f %18.1f %14.1f\n" % (k[-44:], n, fe, wr, fetch_factor(k), mb, (fe + wr) * 1024 / 1e6))
    # calibration of the FETCH_SIZE factor on kernels whose read volume is known exactly (benchmark configuration): the guide's x2 holds for
    # accesses that fetch whole 128-byte lines (tallied at 64 B); kernels that read shorter contiguous pieces are counted 1:1
    known = KNOWN_READS
    buf.write("# FETCH_SIZE calibration on this run (known read bytes / counted bytes):\n")
    for k, n, fe, wr, mb in rows:
        for name, nbytes in known.items():
            if nbytes and name in k:
                buf.write("#   %-40s reads %7.1f MB, FETCH_SIZE %7.1f MB -> factor %.2f\n" % (k[-40:], nbytes / 1e6, fe * 1024 / 1e6, nbytes / (fe * 1024)))
    open(os.path.join(prof, tag + "_pmc_hbm_traffic.txt"), "w").write(buf.getvalue())
    stage = [(r[0], r[1], r[2], r[3], r[4]) for r in rows if any(s in r[0] for s in ("k_corr_prep", "k_corr_raw", "k_corr_tail", "k_corr_box", "k_corr_fused"))]
    # per launch of the fused kernel: since round 5 ONE launch carries both directions of the pair (with two k_corr_prep launches)
    nfused = max([r[1] for r in stage if "k_corr_fused" in r[0] or "k_corr_box" in r[0]] + [1])
    total = sum(r[4] * r[1] for r in stage) * 1e6 / nfused
    cc = [r for r in rows if any(k in r[0] for k in ("k_argmin_voxel", "k_argmin_wave", "k_gather_box3", "k_keys_to_idx_min", "k_cert_voxel", "k_cert_wave", "k_cert_gather",
                                                       "k_cert_arm", "k_cert_plain_finalize", "k_cert_plain_resolve"))]
    cc_pair = sum(r[4] * r[1] for r in cc) * 1e6              # the PMC passes register ONE pair: calls x bytes per call
    # the descriptor stage, bytes per image: two passes (this run) and the single-pass kernels (the passes made with CVX_MIND_SINGLE=1)
    def mind_bytes(rs, names):
        return sum(r[4] for r in rs if any(nm in r[0] for nm in names)) * 1e6
    two_pass_names = ("k_minmax_partial", "k_mind_stats_init", "k_mind_march<", "k_mind_finish_pool")
    single_names = ("k_minmax_partial", "k_mind_stats_init", "k_mind_march_pool", "k_mind_repair")
    mind_two = mind_bytes(rows, two_pass_names)
    sfa, sfc = pmc(src + "/single_fetch")
    swa, swc = pmc(src + "/single_write")
    srows = []
    for k in sorted(set(sfa) | set(swa)):
        n = max(sfc[k].get("FETCH_SIZE", 0), swc[k].get("WRITE_SIZE", 0))
        fe = sfa[k].get("FETCH_SIZE", 0.0) / max(sfc[k].get("FETCH_SIZE", 1), 1)
        wr = swa[k].get("WRITE_SIZE", 0.0) / max(swc[k].get("WRITE_SIZE", 1), 1)
        srows.append((k, n, fe, wr, (fetch_factor(k) * fe + wr) * 1024 / 1e6))
    mind_one = mind_bytes(srows, single_names) if srows else None
    if srows:
        ssa, ssc = pmc(src + "/single_sq")
        sqa, sqc = pmc(src + "/sq")
        sb = io.StringIO()
        sb.write("# MIND-SSC descriptor stage of the benchmark pair, per image: two passes (default) against the single-pass kernels (CVX_MIND_SINGLE=1)\n"
                 "# traffic = (factor*FETCH_SIZE + WRITE_SIZE)*1024 per launch (factor 2.00 = upper bound for 16-byte streamed reads), algorithmic bytes 97.8 MB per image\n")
        sb.write("%-40s %8s %12s %12s %12s %14s %14s %14s\n" % ("kernel", "calls", "FETCH_KiB", "WRITE_KiB", "MB/call", "insts_valu", "insts_lds", "lds_conflict"))
        for title, rs, names, qa, qc in (("two passes", rows, two_pass_names, sqa, sqc), ("single pass", srows, single_names, ssa, ssc)):
            sb.write("# %s\n" % title)
            for r in rs:
                if any(nm in r[0] for nm in names):
                    q = lambda nm: qa[r[0]].get(nm, 0) / max(qc[r[0]].get(nm, 1), 1)
                    sb.write("%-40s %8d %12.1f %12.1f %12.1f %14.0f %14.0f %14.0f\n" % (r[0][-40:], r[1], r[2], r[3], r[4], q("SQ_INSTS_VALU"), q("SQ_INSTS_LDS"), q("SQ_LDS_BANK_CONFLICT")))
        sb.write("# per image: two passes %.1f MB, single pass %.1f MB (algorithmic 97.8 MB)\n" % (mind_two / 1e6, mind_one / 1e6))
        db = glob.glob(src + "/single_stats/**/*_results.db", recursive=True)
        if db:
            out = subprocess.run([sys.executable, os.path.join(HERE, "rocpd_stats.py"), db[0]], stdout=subprocess.PIPE, text=True).stdout
            sb.write("# kernel durations with CVX_MIND_SINGLE=1 (rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1): MIND kernels only\n")
            sb.write("".join(l + "\n" for l in out.splitlines() if "k_mind" in l or "k_minmax" in l or l.startswith("kernel")))
        open(os.path.join(prof, tag + "_mind_single_pass.txt"), "w").write(sb.getvalue())
    json.dump({"correlate_stage_bytes_per_launch": total, "coupled_convex_bytes_per_pair": cc_pair, "mind_two_pass_bytes_per_image": mind_two, "mind_single_pass_bytes_per_image": mind_one,
               "plain_argmin_bytes_per_direction": sum(r[4] * r[1] for r in rows if "k_cert_plain_stream" in r[0] or "k_argmin4" in r[0]) * 1e6 / 2,
               "coupled_convex_kernels": {r[0]: {"calls": r[1], "bytes_per_call": r[4] * 1e6} for r in cc}, "source": "profiles/%s_pmc_hbm_traffic.txt" % tag,
               "kernels": {r[0]: r[4] * 1e6 for r in stage},
               "formula": "(factor*FETCH_SIZE + WRITE_SIZE)*1024 (factor 2: 16-byte streaming reads) summed over k_corr_prep, k_corr_fused (k_corr_tail_compact when the volume has an interleaved-order tail), per launch = per direction", "measured_at_commit": os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip(),
               "corr_sources_sha16": corr_sha()},
              open(os.path.join(prof, "pmc_hbm_traffic.json"), "w"), indent=1)
    sa, sc = pmc(src + "/sq")
    names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"]
    with open(os.path.join(prof, tag + "_pmc_sq_counters.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc %s -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched\n" % " ".join(names))
        f.write("# per-dispatch averages (SQ cycle counters are in quad-cycles)\n")
        f.write("%-44s %6s " % ("kernel", "calls") + " ".join("%13s" % n[3:].lower()[:13] for n in names) + "\n")
        for k in sorted(sa, key=lambda k: -sa[k].get("SQ_WAVE_CYCLES", 0)):
            f.write("%-44s %6d " % (k[-44:], sc[k].get("SQ_WAVES", 0)) + " ".join("%13.0f" % (sa[k].get(n, 0) / max(sc[k].get(n, 1), 1)) for n in names) + "\n")
    for name, dst in (("bench_line.json", tag + "_bench_line.json"), ("bench_under_rocprof.json", tag + "_bench_line_under_rocprofv3.json")):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            open(os.path.join(prof, dst), "w").write(open(p).read())
    print("correlation stage HBM bytes per launch: %.1f MB" % (total / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
