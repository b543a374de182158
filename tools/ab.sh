#!/bin/bash
# Developer tool (GPU box): A/B of environment switches on the bench loop.
#   bash tools/ab.sh [-w workload] "<env assignments>" ...   (use "A=1" for the default)
W=journal-1pct
if [ "$1" = "-w" ]; then W=$2; shift 2; fi
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== [$v] $W"
  env $v python bench.py $BENCH_ARGS --workload $W --cpu-iters 0 --no-coloring --no-fp32-operands --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['value_min'], d['value_max'], 'first', d['config']['first_order_steps'], d['config']['first_order_one_half_matrix_steps'], d['device_us_per_step'], 'spmm', d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
