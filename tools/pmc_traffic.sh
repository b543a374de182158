#!/bin/bash
# Developer tool (GPU box, from the repo root): fabric-side traffic of every kernel of the default bench workload, two
# separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one pass), corrected as MI355X_MICROARCH.md prescribes.
#   bash tools/pmc_traffic.sh r01_f      -> gpurun_out/<tag>_pmc_traffic_journal-1pct.json (copy it to profiles/)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
# shipped path first (optimistic chunks: the exponential as one first-order product, k_spmm_mfma<4, ...>: every launch of it does work)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/S_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 150 --warmup 10 --cpu-iters 0 --no-coloring --no-fp32-operands > $OUT/S_$c.log 2>&1
done
export MMW_SYNC_PLAN=1   # exact launches only: no early-exited stages in the per-kernel means (Lanczos epilogue, k_spmm_mfma<1, ...>)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --cpu-iters 0 --no-coloring --no-fp32-operands > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][c][0] += 1; acc[k][c][1] += float(r["Counter_Value"])
allk = {k: {c: {"launches": v[0], "mean_KB": round(v[1] / v[0], 1)} for c, v in d.items()} for k, d in acc.items()}
accS = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/S_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0]
            accS[k][c][0] += 1; accS[k][c][1] += float(r["Counter_Value"])
shipped = {k: {c: {"launches": v[0], "mean_KB": round(v[1] / v[0], 1)} for c, v in d.items()} for k, d in accS.items()
           if any(t in k for t in ("k_spmm_mfma<4", "k_spmm_mfma<(int)4", "k_spmm_mfma<5", "k_spmm_mfma<(int)5", "k_sddmm_mfma", "k_loss", "k_dual_h", "k_dual_scal"))}
key = [k for k in allk if "k_spmm_mfma<1" in k or "k_spmm_mfma<(int)1" in k] or [k for k in allk if "k_spmm_blk2<float, 1>" in k or "k_spmm_blk2<float, (int)1>" in k]
rec = {"workload": "journal-1pct",
       "command": "MMW_SYNC_PLAN=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 50 --warmup 5 --cpu-iters 0 --no-coloring",
       "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reads 1/2 of a wide (16 B/lane) coalesced read stream -> x2; WRITE_SIZE exact; unit KB -> x1024",
       "note": "exact launches (plan read back every iteration); the whole working set (~110 MB) is resident in the 256 MiB Infinity Cache, whose hits these fabric-side counters include",
       "all_kernels_KB": allk}
if key:
    k = key[0]
    f, w = allk[k]["FETCH_SIZE"], allk[k]["WRITE_SIZE"]
    rec.update({"kernel": k.split("(")[0] + " (Lanczos epilogue)", "FETCH_SIZE_KB_mean": f["mean_KB"], "WRITE_SIZE_KB_mean": w["mean_KB"],
                "launches": f["launches"], "traffic_bytes_per_launch": int((2 * f["mean_KB"] + w["mean_KB"]) * 1024)})
rec["shipped_path_kernels_KB"] = shipped
k5 = [k for k in shipped if "k_spmm_mfma<5" in k or "k_spmm_mfma<(int)5" in k]
if k5:
    f, w = shipped[k5[0]]["FETCH_SIZE"], shipped[k5[0]]["WRITE_SIZE"]
    rec["traffic_bytes_per_launch_first_order_one_half"] = int((2 * f["mean_KB"] + w["mean_KB"]) * 1024)
    rec["first_order_one_half_kernel"] = k5[0].split("(")[0] + " (first-order epilogue, matrix in one fp16 half; shipped path)"
kf = [k for k in shipped if "k_spmm_mfma<4" in k or "k_spmm_mfma<(int)4" in k]
if kf:
    f, w = shipped[kf[0]]["FETCH_SIZE"], shipped[kf[0]]["WRITE_SIZE"]
    rec["traffic_bytes_per_launch_first_order"] = int((2 * f["mean_KB"] + w["mean_KB"]) * 1024)
    rec["first_order_kernel"] = kf[0].split("(")[0] + " (first-order epilogue; shipped path, no MMW_SYNC_PLAN)"
json.dump(rec, open("$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_traffic_journal-1pct.json", "w"), indent=1)
print({k: rec.get(k) for k in ("kernel", "FETCH_SIZE_KB_mean", "WRITE_SIZE_KB_mean", "launches", "traffic_bytes_per_launch", "traffic_bytes_per_launch_first_order", "traffic_bytes_per_launch_first_order_one_half")})
PY
rm -rf $OUT  # raw traces are large: only the summaries above travel back
