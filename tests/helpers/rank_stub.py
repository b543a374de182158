"""Stand-in for one bench rank (CPU): joins the gloo group named by the launcher's environment, contributes one objective
record through the same gather the benchmark uses, rank 0 prints one JSON line.  Used by tests/test_bench_launcher.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.distributed as dist  # noqa: E402

from sig_sdp_mmw_amd import sharding  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    raise SystemExit(7)
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
per = 2
mine = sharding.instances_of_rank(world * per, rank, world)
table = sharding.gather_records([[i, 10 + i, 0, 0.5 * i, 3, 1.0] for i in mine], world * per, rank, world, dist=dist)
dist.barrier()
if rank == 0:
    print(json.dumps({"n": int(table.shape[0]), "ids": [int(x) for x in table[:, 0]], "local_rank": os.environ["LOCAL_RANK"],
                      "master": os.environ["MASTER_ADDR"]}), flush=True)
dist.destroy_process_group()
