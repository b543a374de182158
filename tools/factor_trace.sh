#!/bin/bash
# Developer tool (GPU box, from the repo root): the kernels of ONE factor call in launch order, from a rocprofv3 kernel trace of
# tools/probe_timeline.py restricted to one slot count.   bash tools/factor_trace.sh 186
Z=${1:-186}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ftrace
rm -rf $OUT; mkdir -p $OUT
ZS=$Z rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/probe_timeline.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the factor starts after the last k_sddmm / k_loss of the loop
last = max(i for i, r in enumerate(rows) if "k_loss" in r["Kernel_Name"] or "k_sddmm" in r["Kernel_Name"])
fac = rows[last + 1:]
t0 = int(fac[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in fac)
acc = collections.OrderedDict()
busy = 0
for r in fac:
    n = r["Kernel_Name"].split("(")[0][-44:]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d; busy += d
print("factor span %.2f ms, kernels busy %.2f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e3, len(fac)))
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%8.2f ms %6d x %8.1f us  %s" % (d / 1e3, c, d / c, n))
PY
rm -rf $OUT
