#!/bin/bash
# SQ / memory counters of the certified-fast correlation kernel at the benchmark shape -> gpurun_out/cert_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/cert_pmc.txt; : > $O
SETS=${CC_SETS:-"A B C D E"}
for key in $SETS; do
  case $key in
    A) set="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR";;
    B) set="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU";;
    C) set="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR";;
    D) set="GRBM_GUI_ACTIVE MemUnitBusy MemUnitStalled TA_BUSY_avr";;
    E) set="TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum";;
    F) set="SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED";;
  esac
  rm -rf /tmp/cp
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/cp -o r -- python $R/tools/experiments/cert_time.py $CC_MASK > /dev/null 2>&1
  python - >> $O <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/cp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in acc:
    if 'k_corr_cert' in k or 'k_corr_fused' in k:
        print(k[-40:], ' '.join('%s=%.4g' % (c, acc[k][c] / cnt[k][c]) for c in sorted(acc[k])))
PY
done
cat $O
