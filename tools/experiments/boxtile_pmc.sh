#!/bin/bash
# SQ counters of the forward-box kernels for box_fwd_tile variants:  tools/experiments/boxtile_pmc.sh v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  O=$R/gpurun_out/btpmc_$v; rm -rf $O; mkdir -p $O
  ADAM_REPS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -o r -- python $R/tools/time_adam.py "box_fwd_tile=$v" > $O/log 2>&1
  ADAM_REPS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -o r -- python $R/tools/time_adam.py "box_fwd_tile=$v" >> $O/log 2>&1
  echo "== box_fwd_tile=$v"
  python $R/tools/pmc_summary.py $O box3_tile box3_march > $O/summary.txt; cat $O/summary.txt
  rm -rf $O/sq $O/sq2
done
