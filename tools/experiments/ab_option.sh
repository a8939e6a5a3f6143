#!/bin/bash
# A/B of one CVX_* environment switch on the benchmark pair, alternating runs on one box:  tools/experiments/ab_option.sh CVX_MIND_RECORDS 0 1 [repeats]
V=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do for x in $A $B; do
  env $V=$x python bench.py --no-cpu-baseline --no-batched 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); st=d['stages_ms']
print('$V=$x value %.1f ms %.4f' % (d['value'], d['ms_per_step']), ' '.join('%s %.4f' % (k[:10], v) for k, v in st.items()))"
done; done
