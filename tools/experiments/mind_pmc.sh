#!/bin/bash
# Counters of the two MIND kernels of the pipeline (bench.py --steps 1 --warmup 0) -> gpurun_out/mind_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/mind_pmc.txt; : > $O
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "MemUnitBusy MemUnitStalled WriteUnitStalled" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  rm -rf /tmp/mp
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/mp -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched > /dev/null 2>&1
  python - >> $O <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/mp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in acc:
    if 'k_mind' in k or 'k_argmin4' in k or 'k_box3_fast' in k:
        print(k[-40:], ' '.join('%s=%.4g' % (c, acc[k][c] / cnt[k][c]) for c in sorted(acc[k])))
PY
done
cat $O
