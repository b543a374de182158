"""Developer tool (GPU box): the record gather of sharding.gather_records on RCCL with one rank -- API check and timing."""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sig_sdp_mmw_amd import sharding

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
recs = [[0, 186, 0.0, 3.25, 20, 2000.0]]
for i in range(4):
    t0 = time.perf_counter()
    table = sharding.gather_records(recs, 1, 0, 1, dist=dist, device="cuda", _force_collective=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("gather %.0f us, barrier %.0f us" % ((t1 - t0) * 1e6, (t3 - t2) * 1e6), table.shape, np.allclose(table[0], recs[0]))
dist.destroy_process_group()
