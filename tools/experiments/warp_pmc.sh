#!/bin/bash
# Memory-path counters of the Adam loop's kernels (tools/time_adam.py, fast mode; ADAM_STORAGE=fp16 for half-precision records):
#   tools/experiments/warp_pmc.sh            -> gpurun_out/warp_pmc_<storage>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R ADAM_REPS=1
for st in fp32 fp16; do
  O=$R/gpurun_out/warp_pmc_$st.txt; : > $O
  i=0
  for set in "TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "MemUnitBusy MemUnitStalled" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
    i=$((i+1)); rm -rf /tmp/wp
    ADAM_STORAGE=$st timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/wp -o r -- python $R/tools/time_adam.py "" > /dev/null 2>&1
    python - "$st" >> $O <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/wp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in acc:
    if any(t in k for t in ('warp_grad', 'box3')):
        print(k[-46:], ' '.join('%s=%.4g' % (c, acc[k][c] / cnt[k][c]) for c in sorted(acc[k])))
PY
  done
  echo "== $st"; cat $O
done
