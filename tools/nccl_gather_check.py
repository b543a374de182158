"""Developer tool (GPU box): the record gather of sharding.gather_records on RCCL with one rank -- API check and timing."""
import os, time, torch, numpy as np
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
buf=np.arange(12,dtype=np.float64).reshape(2,6)
mine=torch.from_numpy(buf).to("cuda")
flat=torch.empty((2,6),dtype=mine.dtype,device=mine.device)
dist.all_gather_into_tensor(flat, mine); torch.cuda.synchronize()
for i in range(3):
    t0=time.perf_counter(); mine=torch.from_numpy(buf).to("cuda"); flat=torch.empty((2,6),dtype=mine.dtype,device=mine.device); dist.all_gather_into_tensor(flat, mine); t=flat.cpu().numpy(); t1=time.perf_counter()
    torch.cuda.synchronize(); t2=time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); t3=time.perf_counter()
    print("gather %.0f us, barrier %.0f us"%((t1-t0)*1e6,(t3-t2)*1e6), np.array_equal(t,buf))
dist.destroy_process_group()
