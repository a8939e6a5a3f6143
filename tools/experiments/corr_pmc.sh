#!/bin/bash
# Memory-path counters of the fused correlation kernel at the benchmark shape (tools/time_corr.py 12:26:32:37:6) -> gpurun_out/corr_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/corr_pmc.txt; : > $O
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "MemUnitBusy MemUnitStalled" \
           "TA_BUSY_avr TA_BUSY_max" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  rm -rf /tmp/cp
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/cp -o r -- python $R/tools/time_corr.py 12:26:32:37:6 > /dev/null 2>&1
  python - >> $O <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/cp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in acc:
    if 'k_corr_fused' in k:
        print(k[-46:], ' '.join('%s=%.4g' % (c, acc[k][c] / cnt[k][c]) for c in sorted(acc[k])))
PY
done
cat $O
