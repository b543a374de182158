#!/bin/bash
# Developer tool (GPU box, from the repo root): every profile the round's documents cite, summaries only, under gpurun_out/.
#   bash tools/collect_profiles.sh r03 [part]     part: all (default) | bench | traffic | mfma | workloads
R=${1:-rXX}
PART=${2:-all}
cd $GRAFT_REPO_ROOT
if [ "$PART" = all ] || [ "$PART" = bench ]; then
  python bench.py > gpurun_out/${R}_g_bench_journal-1pct_plain.json 2> gpurun_out/${R}_g.err
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_l_bench_journal-1pct_steps20.json 2> gpurun_out/${R}_l.err   # the driver's command
  bash tools/profile_bench.sh ${R}_b_bench_journal-1pct --cpu-iters 0 --no-coloring --no-fp32-operands > gpurun_out/${R}_b.log 2>&1
  bash tools/profile_bench.sh ${R}_a_coloring_journal-1pct --cpu-iters 0 --steps 20 --warmup 5 --no-fp32-operands --no-coloring-warm > gpurun_out/${R}_a.log 2>&1
  python tools/spmm_bench.py journal-1pct > gpurun_out/${R}_f_spmm_micro.log 2>&1
  MMW_BENCH_LANCZOS=1 python tools/spmm_bench.py journal-1pct >> gpurun_out/${R}_f_spmm_micro.log 2>&1
  MMW_STAMPS=1 MMW_BENCH_FIRST=1 MMW_BENCH_MODES=2 python tools/spmm_bench.py journal-1pct >> gpurun_out/${R}_f_spmm_micro.log 2>&1
  MMW_SD_STAMPS=1 python tools/probe_first.py full 48 2>&1 | grep -i stamps | tail -2 >> gpurun_out/${R}_f_spmm_micro.log
fi
if [ "$PART" = all ] || [ "$PART" = traffic ]; then bash tools/pmc_traffic.sh ${R}_c > gpurun_out/${R}_c.log 2>&1; fi
if [ "$PART" = all ] || [ "$PART" = mfma ]; then bash tools/pmc_mfma.sh ${R}_e > gpurun_out/${R}_e.log 2>&1; fi
if [ "$PART" = all ] || [ "$PART" = workloads ]; then
  for w in er-1pct er-50k er-5pct-2k journal-native; do
    bash tools/profile_bench.sh ${R}_d_bench_$w --workload $w --cpu-iters 0 --no-coloring --no-fp32-operands > gpurun_out/${R}_d_$w.log 2>&1
  done
  python bench.py --workload er-5pct-2k --instances-per-gpu 8 --cpu-iters 0 > gpurun_out/${R}_h_bench_er-5pct-2k_x8.json 2> gpurun_out/${R}_h.err
  python bench.py --gpus 2 --backend gloo --single-device --workload er-5pct-2k --instances-per-gpu 4 --cpu-iters 0 > gpurun_out/${R}_i_bench_2ranks_1gpu.json 2> gpurun_out/${R}_i.err
fi
ls -la gpurun_out | tail -30
