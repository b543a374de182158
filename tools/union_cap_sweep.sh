#!/bin/bash
# Developer tool (GPU box): matrix-core SpMM and whole-step throughput against the union cap of the matrix-core blocking.
for cap in 640 512 480 448 416 384; do
  echo "== MMW_MF_UNION_CAP=$cap"
  MMW_MF_UNION_CAP=$cap MMW_BENCH_MODES=2 python tools/spmm_bench.py journal-1pct 2>&1 | tail -1
  MMW_MF_UNION_CAP=$cap MMW_BENCH_MODES=2 MMW_BENCH_LANCZOS=1 python tools/spmm_bench.py journal-1pct 2>&1 | tail -1
  MMW_MF_UNION_CAP=$cap python bench.py --cpu-iters 0 --no-coloring 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('it/s', d['value'], 'spmm', d['roofline']['avg_launch_us'], d['roofline']['avg_launch_us_back_to_back'], d['device_us_per_step'])"
done
