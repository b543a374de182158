#!/bin/bash
# Developer tool: issue / wait counters of the blocked SpMM micro-benchmark (run on the GPU box from the repo root).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_spmm2
mkdir -p $OUT
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/spmm_bench.py journal-1pct > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "spmm_blk" not in r["Kernel_Name"] or "Li0" in r["Kernel_Name"]: pass
        if "k_spmm_blk" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"][:32], r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, v in acc.items(): print(k, "n", v[0], "mean %.4g" % (v[1] / v[0]))
PY
