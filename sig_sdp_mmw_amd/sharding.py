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


def gather_records(local_records, n_instances, rank, world, dist=None, device=None):
    """All ranks end up with the (n_instances, RECORD_LEN) table ordered by instance id.

    `local_records`: list of RECORD_LEN-float rows produced on this rank.  With world == 1 no
    collective is issued.  `dist` is torch.distributed (already initialised) for world > 1.
    """
    per_rank = (n_instances + world - 1) // world
    buf = np.full((per_rank, RECORD_LEN), np.nan, dtype=np.float64)
    for j, rec in enumerate(local_records):
        buf[j] = np.asarray(rec, dtype=np.float64)
    if world == 1 or dist is None:
        table = buf
    else:
        import torch
        mine = torch.from_numpy(buf)
        if device is not None:
            mine = mine.to(device)
        table = None
        if device is not None and str(device) != "cpu":
            # RCCL: one collective into one tensor and ONE copy back (a list of `world` outputs costs `world` device-to-host copies,
            # ~20 us each -- a tenth of a 20-step timed region on 8 GPUs)
            try:
                flat = torch.empty((world * per_rank, RECORD_LEN), dtype=mine.dtype, device=mine.device)
                dist.all_gather_into_tensor(flat, mine)
                table = flat.cpu().numpy()
            except AttributeError:  # an older torch.distributed without the call (the same on every rank)
                table = None
        if table is None:
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
