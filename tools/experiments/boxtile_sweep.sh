#!/bin/bash
# per-kernel durations of the Adam loop for box_fwd_tile variants (rocprofv3 kernel trace):  tools/experiments/boxtile_sweep.sh v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python tools/experiments/check_boxtile.py 2>&1 | tail -3
for v in "$@"; do
  ADAM_REPS=2 tools/gpu_kstats.sh bt_$v python tools/time_adam.py "box_fwd_tile=$v" > /dev/null 2>&1
  echo "== box_fwd_tile=$v: $(grep 'us / iteration' gpurun_out/bt_$v/cmd.out | tail -1)"
  grep -E "box3_tile|box3_march|warp_grad_fast|box3_fast" gpurun_out/bt_$v/kernel_stats.txt | cut -c1-130
done
