#!/bin/bash
# per-kernel durations of a command on the GPU box:  tools/gpu_kstats.sh <outdir under gpurun_out> <command...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="cd $R && $*"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/raw -o r -- bash -c "$CMD" > $O/cmd.out 2> $O/cmd.err
DB=$(find $O/raw -name '*_results.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB > $O/kernel_stats.txt
rm -rf $O/raw
head -30 $O/kernel_stats.txt
