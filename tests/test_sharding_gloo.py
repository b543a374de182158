"""World-size-2 gloo test of the instance sharding + objective gather (the N > 1 path of bench.py and
sig_sdp_mmw_amd.sharding), on CPU.  The per-instance solve is the CPU oracle here: the test covers
placement, record layout and the collective, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from sig_sdp_mmw_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_instance(i):
    from sig_sdp_mmw_amd.graphs import er_contention_graph
    return er_contention_graph(40 + 4 * i, 0.2, seed=100 + i), 5 + (i % 3)


def _solve_one(i, state, Z):
    from oracle import mmw_oracle as orc
    rng = np.random.RandomState(i)
    o = orc.MMWOracle(nit=3, eta=0.05)
    o.run(Z, state, lambda it, K, D: orc.sketch_rows(rng.randn(K, D)), keep_trace=True, factor=False)
    return 0, float(np.max(o.trace["e_this"][-1])), 3


def _worker(rank, world, port, n_inst, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    table = sharding.solve_sharded(n_inst, _make_instance, _solve_one, rank=rank, world=world, dist=dist)
    dist.barrier()
    q.put((rank, table))
    dist.destroy_process_group()


def test_placement_is_round_robin_and_complete():
    for n, w in ((5, 2), (64, 8), (3, 4), (8, 8)):
        seen = sorted(i for r in range(w) for i in sharding.instances_of_rank(n, r, w))
        assert seen == list(range(n))
        assert all(i % w == r for r in range(w) for i in sharding.instances_of_rank(n, r, w))


def test_single_rank_needs_no_collective():
    t = sharding.solve_sharded(3, _make_instance, _solve_one)
    assert t.shape == (3, sharding.RECORD_LEN) and list(t[:, 0]) == [0, 1, 2]


@pytest.mark.timeout(300)
def test_world_size_2_gloo_gathers_every_instance():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    n_inst, world = 5, 2
    port = _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_inst, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = sharding.solve_sharded(n_inst, _make_instance, _solve_one)
    for r in range(world):
        t = got[r]
        assert t.shape == (n_inst, sharding.RECORD_LEN)
        assert np.array_equal(t[:, :5], ref[:, :5])  # identical objectives on every rank, ordered by instance id
