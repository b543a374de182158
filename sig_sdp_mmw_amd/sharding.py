"""Instance sharding across the GPUs of one node.

The MMW path shards by *instance*: graphs (seeds) and slot counts are independent
(`for seed in range(REPEAT)` in every reference driver, e.g. sim_script/journal_version/sim_all_bler.py:30-36),
so instance i goes to rank i mod world, every rank solves its instances on its own GPU with no
communication, and one small all_gather (RCCL over xGMI when the backend is "nccl", gloo on CPU)
collects a fixed-size objective record per instance.  A single instance is never split ("replicas only").
"""
import time

import numpy as np

RECORD_FIELDS = ("instance_id", "Z", "remainder", "max_violation", "iters", "wall_us")
RECORD_LEN = len(RECORD_FIELDS)


def instances_of_rank(n_instances, rank, world):
    """Round-robin placement: instance i -> rank i % world."""
    return list(range(rank, n_instances, world))


_RCCL_BUFFERS = {}  # (rows per rank, world, device) -> page-locked staging and device tensors of the record gather, made on first use


def _rccl_buffers(per_rank, world, device):
    import torch
    key = (per_rank, world, str(device))
    if key not in _RCCL_BUFFERS:
        _RCCL_BUFFERS[key] = (torch.empty((per_rank, RECORD_LEN), dtype=torch.float64).pin_memory(),
                              torch.empty((per_rank, RECORD_LEN), dtype=torch.float64, device=device),
                              torch.empty((world * per_rank, RECORD_LEN), dtype=torch.float64, device=device),
                              torch.empty((world * per_rank, RECORD_LEN), dtype=torch.float64).pin_memory())
    return _RCCL_BUFFERS[key]


def gather_records(local_records, n_instances, rank, world, dist=None, device=None, _force_collective=False):
    """All ranks end up with the (n_instances, RECORD_LEN) table ordered by instance id.

    `local_records`: list of RECORD_LEN-float rows produced on this rank.  With world == 1 no
    collective is issued.  `dist` is torch.distributed (already initialised) for world > 1.
    """
    per_rank = (n_instances + world - 1) // world
    buf = np.full((per_rank, RECORD_LEN), np.nan, dtype=np.float64)
    for j, rec in enumerate(local_records):
        buf[j] = np.asarray(rec, dtype=np.float64)
    if (world == 1 and not _force_collective) or dist is None:
        table = buf
    else:
        import torch
        table = None
        if device is not None and str(device) != "cpu":
            # RCCL: one collective into one tensor and ONE copy back (a list of `world` outputs costs `world` device-to-host copies,
            # ~20 us each -- a tenth of a 20-step timed region on 8 GPUs)
            if hasattr(dist, "all_gather_into_tensor"):  # (an older torch.distributed lacks it on every rank alike)
                host_in, dev_in, dev_out, host_out = _rccl_buffers(per_rank, world, device)
                host_in.copy_(torch.from_numpy(buf))
                dev_in.copy_(host_in, non_blocking=True)
                dist.all_gather_into_tensor(dev_out, dev_in)
                host_out.copy_(dev_out, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                table = host_out.numpy().copy()
        if table is None:
            mine = torch.from_numpy(buf)
            if device is not None:
                mine = mine.to(device)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            table = np.concatenate([p.cpu().numpy() for p in parts], axis=0)
    table = table[~np.isnan(table[:, 0])]
    order = np.argsort(table[:, 0], kind="stable")
    return table[order]


def solve_sharded(n_instances, make_instance, solve_one, rank=0, world=1, dist=None, device=None):
    """Solve `n_instances` independent instances, `solve_one(instance_id, state, Z) -> (remainder,
    max_violation, iters)` on this rank's share, and gather the records.

    `make_instance(instance_id) -> (state, Z)`.
    """
    mine = []
    for i in instances_of_rank(n_instances, rank, world):
        state, Z = make_instance(i)
        t0 = time.perf_counter()
        rem, vio, iters = solve_one(i, state, Z)
        mine.append([i, Z, rem, vio, iters, (time.perf_counter() - t0) * 1e6])
    return gather_records(mine, n_instances, rank, world, dist=dist, device=device)
