#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace statistics of the default bench command, then the two PMC passes that give
# HBM traffic (separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Output under gpurun_out/round/.
# Post-process here with: python tools/make_profiles.py gpurun_out/round <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batched > $O/bench_under_rocprof.json 2> $O/stats.log
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats_fast -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batched --adam-mode fast > $O/bench_fast_under_rocprof.json 2> $O/stats_fast.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched > /dev/null 2> $O/fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched > /dev/null 2> $O/write.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched > /dev/null 2> $O/sq.log
# the single-pass descriptor (option mind_single = 1): kernel durations and the three counter passes of its kernels
CVX_MIND_SINGLE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/single_stats -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batched > /dev/null 2> $O/single_stats.log
CVX_MIND_SINGLE=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/single_fetch -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched > /dev/null 2> $O/single_fetch.log
CVX_MIND_SINGLE=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/single_write -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched > /dev/null 2> $O/single_write.log
CVX_MIND_SINGLE=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/single_sq -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched > /dev/null 2> $O/single_sq.log
cd $R && python bench.py --steps 10 --warmup 2 > $O/bench_line.json 2> $O/bench.log
ls $O
