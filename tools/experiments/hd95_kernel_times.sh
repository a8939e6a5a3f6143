#!/bin/bash
# k_surface_dist_hist durations by misregistration (rocprofv3 kernel trace of tools/experiments/time_hd95.py, one amplitude per run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for amp in "$@"; do
  rm -rf /tmp/hdp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/hdp -o r -- python $R/tools/experiments/time_hd95.py $amp 2>/dev/null | grep "^amp"
  python $R/tools/rocpd_stats.py $(find /tmp/hdp -name "*_results.db" | head -1) | grep -E "surf|k_label_bits|hist_order|percentile|label_overlap|Fill" | cut -c1-120
done
