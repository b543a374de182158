#!/bin/bash
# Developer tool (GPU box, from the repo root): one rocprofv3 --pmc pass over a python command, per-kernel means.
#   bash tools/pmc.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" bench.py --steps 30 --warmup 5 --cpu-iters 0
C="$1"; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$(echo $C | tr ' ' '_' | cut -c1-60)
rm -rf $OUT; mkdir -p $OUT
S=$1; shift
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/$S "$@" > $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, v in sorted(acc.items()):
        if v[1] > 0: print("%-42s %-24s n %5d mean %.4g" % (k[0], k[1], v[0], v[1] / v[0]))
PY
