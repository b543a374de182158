cd $GRAFT_REPO_ROOT
A="bench.py --workload er-50k --steps 10 --warmup 3 --cpu-iters 0 --no-coloring --no-fp32-operands --repeats 1"
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  echo "#### $c"
  bash tools/pmc.sh "$c" $A | grep -E "k_sddmm|k_spmm|k_loss|k_dual_h"
done
