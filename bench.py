#!/usr/bin/env python3
"""Benchmark of the MMW hot path on MI355X: MMW iterations/s on the configuration BASELINE.json's
metric is quoted on (N = 10 k, 1 %-sparse interference graph from the journal generator, fp32).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--expm lanczos|taylor]

One "step" = one MMW iteration (DUAL, LOSS, EXPM + X on the pattern, averaging) on a synthetic instance
that is resident in HBM before the timed region starts; the sketches are generated on the device.
N > 1: one process per GPU (torchrun), each rank solves its own independent instance (weak scaling,
instance sharding -- the path has no data-path collective); the per-instance objectives are gathered
with one RCCL all_gather at the end of the timed region.

The JSON line also carries
  roofline     : the CSR SpMM inside exp(L/2)R -- algorithmic bytes per launch
                 divided by its mean launch duration, measured with HIP events on the solver's stream
                 in a second pass over the same steps;
  cpu_baseline : the CPU oracle (oracle/mmw_oracle.py, a NumPy/SciPy port of the reference loop)
                 timed on this host on a bounded number of iterations of the same instance (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)

WORKLOADS = {
    # name: (description, factory, Z or None (= first bisection midpoint of the reference's bounds), dtype)
    "journal-1pct": ("N=10003 journal generator env(cell_size=28, rho=0.0319, seed=s), 0.98% dense (SURVEY 3b)",
                     lambda seed: ("journal", dict(cell_size=28, sta_density_per_1m2=0.0319, seed=seed)), None, "f32"),
    "journal-native": ("N=10092 journal generator env(cell_size=58, rho=75e-4, seed=s), 0.24% dense (SURVEY 3a)",
                       lambda seed: ("journal", dict(cell_size=58, sta_density_per_1m2=75e-4, seed=seed)), None, "f32"),
    "er-1pct": ("N=10000 Erdos-Renyi 1% (SURVEY 3c), Z=32", lambda seed: ("er", dict(K=10000, p=0.01, seed=seed)), 32, "f32"),
    "er-5pct-2k": ("N=2000 Erdos-Renyi 5% (config 2), Z=32, fp64", lambda seed: ("er", dict(K=2000, p=0.05, seed=seed)), 32, "f64"),
    "er-50k": ("N=50000 Erdos-Renyi 0.2% (config 5), Z=32", lambda seed: ("er", dict(K=50000, p=0.002, seed=seed)), 32, "f32"),
    "dense-200": ("N=200 dense (config 1), Z=32, fp64", lambda seed: ("er", dict(K=200, p=1.0, seed=seed)), 32, "f64"),
}


def make_state(kind, kw):
    from sig_sdp_mmw_amd import graphs
    return graphs.journal_graph(**kw) if kind == "journal" else graphs.er_contention_graph(**kw)


def first_midpoint(state):
    """The slot count the reference's binary search probes first (binary_search_relaxation.py:13-29,46)."""
    S, Q, _ = state
    T = (S + S.T).tocsr()
    ub = int(np.max(np.diff(T.indptr))) + 1  # the diagonal stays stored after setdiag(0), as executed
    lb = int(np.max(np.diff(Q.indptr))) + 1
    return (lb + ub) // 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)   # nit = 150 is the reference's production setting
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="journal-1pct", choices=sorted(WORKLOADS))
    ap.add_argument("--expm", default="lanczos", choices=["lanczos", "taylor"])
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"])
    ap.add_argument("--eta", type=float, default=0.04)
    ap.add_argument("--cpu-iters", type=int, default=-1, help="oracle iterations for cpu_baseline (-1: auto, 0: skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --single-device rehearses N>1 on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." %
                             (args.gpus, args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    import torch
    dist = None
    if args.single_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"

    from sig_sdp_mmw_amd import _lib

    desc, factory, Zfix, dt_default = WORKLOADS[args.workload]
    dtype_name = args.dtype or dt_default
    dtype = _lib.F32 if dtype_name == "f32" else _lib.F64
    w = 4 if dtype_name == "f32" else 8
    kind, kw = factory(rank)  # every rank gets its own instance (seed = rank)
    state = make_state(kind, kw)
    Z = Zfix if Zfix is not None else first_midpoint(state)
    nit = args.warmup + args.steps
    solver = _lib.Solver(Z, state, nit, args.eta, dtype=dtype, device=local_rank)
    method = _lib.EXPM_LANCZOS if args.expm == "lanczos" else _lib.EXPM_TAYLOR
    solver.set_expm(method, 12, 1e-6 if dtype_name == "f32" else 1e-9)
    seed = 1234 + rank

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup, then the timed region: exactly `steps` iterations
    solver.iterate(args.warmup, None, seed)
    solver.sync()
    barrier()
    t0 = time.perf_counter()
    solver.iterate(args.steps, None, seed)
    solver.sync()
    from sig_sdp_mmw_amd import sharding
    # per-instance objective record, gathered to every rank (RCCL all_gather over xGMI, 48 B per instance)
    rec = [rank, Z, 0.0, float(np.max(solver.read(_lib.F_E_THIS))), args.steps, (time.perf_counter() - t0) * 1e6]
    table = sharding.gather_records([rec], world, rank, world, dist=dist, device=coll_dev if dist is not None else None)
    assert table.shape[0] == world
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    info = solver.read(_lib.F_EXPM_INFO)
    m_used = int(info[1])

    # ---- roofline pass: the same number of steps again with HIP events around every kernel class
    solver.reset(nit)
    solver.iterate(args.warmup, None, seed)
    solver.set_profile(True)
    solver.iterate(args.steps, None, seed)
    kt = solver.kernel_times()
    solver.set_profile(False)
    spmm_us, spmm_n = kt["spmm"]
    K, D, nnzL, C = solver.K, solver.D, solver.nnzL, solver.C
    b_spmm = nnzL * (w + 4) + (K + 1) * 4 + 2 * K * D * w
    spmm_avg_us = spmm_us / max(spmm_n, 1)
    achieved = b_spmm / (spmm_avg_us * 1e-6) / 1e9 if spmm_n else 0.0
    traffic = None  # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/)
    for fn in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if fn.endswith(".json") and "pmc_traffic" in fn:
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if rec.get("workload") == args.workload and dtype_name == "f32" and args.expm == "lanczos":
                    traffic = rec["traffic_bytes_per_launch"]
            except Exception:
                pass
    blocked = bool(solver.read(_lib.F_BLOCKING)[0])
    kname = ("k_spmm_blk2 (LDS-staged locality-blocked CSR SpMM, 128-byte half tiles" if blocked else "k_spmm (generic CSR gather SpMM") + " of the %s step)" % args.expm
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_launch": int(b_spmm), "avg_launch_us": round(spmm_avg_us, 2), "launches": int(spmm_n),
                "launches_per_step": round(spmm_n / max(args.steps, 1), 2)}
    phases = {k: round(v[0] / max(args.steps, 1), 2) for k, v in kt.items() if v[1]}

    out = {
        "metric": "mmw_iterations_per_sec", "value": round(world * args.steps / elapsed, 2), "unit": "it/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": args.workload, "description": desc, "K": K, "Z": Z, "D": D, "nnzL": nnzL, "C": C,
                   "eta": args.eta, "expm": args.expm, "krylov_order": m_used, "rng": "device-philox4x32",
                   "instances": world, "parallelism": "instance-sharded x%d" % world},
        "roofline": roofline,
        "device_us_per_step": phases,
        "objectives": {"max_violation_per_instance": [round(float(x), 6) for x in table[:, 3]]},
    }

    # ---- CPU baseline: the oracle on this host, bounded sample of the same instance (rank 0, N = 1 only)
    if rank == 0 and world == 1 and args.cpu_iters != 0:
        from oracle import mmw_oracle as orc
        rng = np.random.RandomState(0)
        pat = orc.Pattern(Z, state)  # one-off state processing, outside the timed sample like on the GPU side
        iters = args.cpu_iters
        if iters < 0:  # size the sample to ~15 s of CPU work from one probe iteration
            p0 = time.perf_counter()
            orc.MMWOracle(nit=1, eta=args.eta).run(Z, state, lambda i, K_, D_: orc.sketch_rows(rng.randn(K_, D_)), factor=False, pattern=pat)
            iters = int(min(60, max(2, 15.0 / max(time.perf_counter() - p0, 1e-3))))
        o = orc.MMWOracle(nit=iters, eta=args.eta)
        c0 = time.perf_counter()
        o.run(Z, state, lambda i, K_, D_: orc.sketch_rows(rng.randn(K_, D_)), factor=False, pattern=pat)
        c1 = time.perf_counter()
        out["cpu_baseline"] = {"value": round(iters / (c1 - c0), 4), "unit": "it/s", "cores": 1, "kind": "port",
                               "sample": "%d MMW iterations of the same instance with the NumPy/SciPy oracle "
                                         "(O(nnz) dual, scipy expm_multiply, host randn, float64), %.1f s" % (iters, c1 - c0),
                               "host_cpus": os.cpu_count()}
    if rank == 0:
        print(json.dumps(out), flush=True)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
