#!/bin/bash
# Developer tool: LDS counters of the blocked SpMM micro-benchmark (run on the GPU box from the repo root).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_spmm
mkdir -p $OUT
for c in ${PMC_SETS:-"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVES SQ_WAVE_CYCLES"}; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n -- python3 $GRAFT_REPO_ROOT/tools/spmm_bench.py journal-1pct > $OUT/$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "spmm" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"][:40], r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, v in acc.items(): print(k, "launch-rows", v[0], "mean", v[1] / v[0])
PY
