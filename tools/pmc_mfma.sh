#!/bin/bash
# Developer tool (GPU box, from the repo root): matrix-core counters of the kernels that run on MFMA -- the rounding's fp64
# projection (k_project_mfma), the factor's Gram / tall GEMM (k_gram, k_gemm_tall) and the bf16 SpMM / SDDMM of the loop.
#   bash tools/pmc_mfma.sh r02_e   ->  gpurun_out/<tag>_pmc_mfma_journal-1pct.json (copy it to profiles/)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
rm -rf $OUT; mkdir -p $OUT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-fp32-operands --no-coloring-warm > $OUT/$n.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(t in k for t in ("mfma", "k_gram", "k_gemm_tall")): continue
        acc[k][r["Counter_Name"]][0] += 1; acc[k][r["Counter_Name"]][1] += float(r["Counter_Value"])
for f in glob.glob("$OUT/*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in acc:
            dur[k][0] += 1; dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rec = {"workload": "journal-1pct", "command": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-iters 0 (two passes)",
       "note": "per-launch means over the loop and the colouring run; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs that ran the kernel, "
               "SQ_BUSY_CYCLES is per SE; mfma_busy_frac = MFMA busy cycles / (1024 SIMDs x mean launch duration x 2.1 GHz)", "kernels": {}}
for k, d in acc.items():
    e = {c: round(v[1] / v[0], 1) for c, v in d.items()}
    e["launches"] = max(v[0] for v in d.values())
    if dur[k][0]:
        e["mean_us"] = round(dur[k][1] / dur[k][0], 2)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["mean_us"] * 2100.0), 4)
    rec["kernels"][k] = e
json.dump(rec, open("$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_mfma_journal-1pct.json", "w"), indent=1)
for k, e in rec["kernels"].items(): print(k[:60], e)
PY
rm -rf $OUT  # raw traces are large: only the summaries above travel back
