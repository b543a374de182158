#!/usr/bin/env python3
"""Benchmark of the MMW hot path on MI355X: MMW iterations/s on the configuration BASELINE.json's
metric is quoted on (N = 10 k, 1 %-sparse interference graph from the journal generator, fp32), and the
wall-clock of the whole binary search to a converged colouring on the same instance.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--instances-per-gpu M] [--expm lanczos|taylor]

One "step" = one MMW iteration (DUAL, LOSS, EXPM + X on the pattern, averaging) on a synthetic instance
that is resident in HBM before the timed region starts; the sketches are generated on the device.

N > 1: one process per GPU.  Either an external launcher started the ranks (torchrun: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or -- with WORLD_SIZE unset -- this process starts them itself:
it spawns N fresh children *before touching the GPU* (no torch.cuda, no HIP call in the parent), relays
rank 0's JSON line and exits non-zero if any rank fails.  Every rank solves its own independent instances
(weak scaling, instance sharding -- the path has no data-path collective); the per-instance objectives are
gathered with one RCCL all_gather that closes the timed region.

--instances-per-gpu M (BASELINE configs[3]: 64 N=2000 graphs, 8 per GPU: `--workload er-5pct-2k
--instances-per-gpu 8`): M resident handles per rank, their iterations enqueued round-robin on M HIP streams.

`value` is the MEDIAN of --repeats (7) timed regions -- each: back to the initial point, W untimed warm-up steps, exactly K
steps between two barriers (+ device synchronisation), closed by the gather of the objective records -- with `value_min` /
`value_max`, the tail of the median region (`region_tail_ms`: read-back, gather, barrier) and the rate without it.
`config.operand_precision` says what the fp32 headline multiplies in (16-bit operands on the matrix cores, fp32 accumulation,
certified against the tolerance); `value_fp32_operands` is the same loop on strictly-fp32 operands (N = 1).

The JSON line also carries
  roofline     : the CSR SpMM inside exp(L/2)R -- algorithmic bytes per launch
                 divided by its mean launch duration, measured with HIP events on the solver's stream
                 in a second pass over the same steps (matrix-core product: the start / stop events its launch
                 carries itself, hipExtLaunchKernelGGL -- the dispatch's own begin and end, what a kernel trace
                 reports; two marker packets around the launch add 1-3 us of marker and dispatch latency);
  coloring     : wall-clock of the binary search on the slot count down to a feasible colouring of the same instance
                 (rank 0, N = 1, one instance per GPU), with the per-probe phase times -- the REFERENCE's search: every probe
                 restarts from Y = 1/C, X = I and runs nit iterations (mmw.py:62-68), state through the host CSR;
  coloring_device_state : the same search with the state handed over on the device (mmw_env_create -> mmw_create_from_env);
  coloring_warm_start   : the opt-in warm-started variant (NOT the reference's search), labelled as such;
  cpu_baseline : the CPU oracle (oracle/mmw_oracle.py, a NumPy/SciPy port of the reference loop)
                 timed on this host on a bounded number of iterations of the same instance (rank 0, N=1).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
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


def make_state(kind, kw, geometry=False):
    from sig_sdp_mmw_amd import graphs
    if kind == "journal":
        return graphs.journal_graph(return_geometry=True, **kw) if geometry else graphs.journal_graph(**kw)
    st = graphs.er_contention_graph(**kw)
    return (st, None) if geometry else st


def first_midpoint(state):
    """The slot count the reference's binary search probes first (binary_search_relaxation.py:13-29,46)."""
    from sig_sdp_mmw_amd.binary_search import binary_search_relaxation
    lb, ub = binary_search_relaxation().set_bounds(state)
    return (lb + ub) // 2


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, cmd, env=None, timeout=None):
    """Start `n` fresh processes running `cmd` (a list), one per rank, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set; nothing in this (parent) process touches the GPU.  Rank 0's stdout is
    captured and returned, the other ranks' stdout goes to stderr.  Returns (exit code, rank-0 stdout):
    the exit code is the first non-zero child status (the remaining children are then terminated by PID)."""
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = str(n)
    base["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(base)
        e["RANK"] = str(r)
        e["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    t0 = time.time()
    rc = 0
    out0 = ""
    live = set(range(n))
    try:
        while live:
            for r in sorted(live):
                p = procs[r]
                if r == 0:
                    try:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or ""
                    except subprocess.TimeoutExpired:
                        continue
                elif p.poll() is None:
                    continue
                live.discard(r)
                if p.returncode != 0 and rc == 0:
                    rc = p.returncode
            if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
                if rc == 0:
                    rc = 124
                break
            if live and 0 not in live:
                time.sleep(0.1)
    finally:
        for r in live:  # a rank failed or timed out: stop exactly the children started here
            p = procs[r]
            if p.poll() is None:
                p.terminate()
        for r in live:
            try:
                o, _ = procs[r].communicate(timeout=20)
                if r == 0:
                    out0 += o or ""
            except subprocess.TimeoutExpired:
                procs[r].kill()
    return rc, out0


def coloring_block(state, dtype_name, nit, eta, warm, geometry=None, speculate=False, device_state=False):
    """Wall-clock of the whole binary search (binary_search_relaxation.run) to a feasible colouring, with the device-RNG /
    batched-rounding fast path of the drop-in class (the flow of sim_script/journal_version/sim_mmw_time.py:30-36)."""
    from sig_sdp_mmw_amd.binary_search import binary_search_relaxation
    from sig_sdp_mmw_amd.mmw import mmw
    bs = binary_search_relaxation()
    bs.verbose = False
    bs.speculate = bool(speculate)
    alg = mmw(nit=nit, eta=eta, dtype=dtype_name, rng="device", seed=1, warm_start=warm)
    bs.feasibility_check_alg = alg
    np.random.seed(0)
    env0 = None
    if device_state:  # the state stays where the device generator made it (mmw_create_from_env): no host CSR, no host pattern build
        from sig_sdp_mmw_amd import _lib
        from sig_sdp_mmw_amd.graphs import min_sinr_dec, _NOISE_FLOOR_DBM
        env0 = _lib.DeviceEnv(geometry["sta_locs"], geometry["ap_locs"], min_sinr=min_sinr_dec(), noise_floor_dbm=_NOISE_FLOOR_DBM)
        state = env0.device_state()
    t0 = time.perf_counter()
    z_vec, Z, rem = bs.run(state)
    wall = time.perf_counter() - t0
    if env0 is not None:
        env0.close()
    per = bs.LOGGED_NP_DATA["bs_search_per_it"]
    lg = alg.LOGGED_NP_DATA
    alg.close()
    ms = lambda us: round(float(us) / 1e3, 2)
    score = None
    if geometry is not None:  # what every reference driver prints after the colouring (pd_mmw_template.py:31-33): BLER on the device scorer
        from sig_sdp_mmw_amd import _lib
        from sig_sdp_mmw_amd.graphs import min_sinr_dec, _NOISE_FLOOR_DBM
        env = _lib.DeviceEnv(geometry["sta_locs"], geometry["ap_locs"], min_sinr=min_sinr_dec(), noise_floor_dbm=_NOISE_FLOOR_DBM)
        s0 = time.perf_counter()
        sinr, bler = env.evaluate(z_vec, int(Z))
        score = {"bler_mean": float(np.mean(bler)), "bler_max": float(np.max(bler)), "sinr_min": float(np.min(sinr)),
                 "scorer_ms": round((time.perf_counter() - s0) * 1e3, 2)}
        env.close()
    its = [int(x) for x in lg["mmw_iters"][:, 5]] if "mmw_iters" in lg and not speculate else None
    return {"wall_s": round(wall, 4), "Z": int(Z), "rem": int(rem), "probes": int(per.shape[0]),
            "nit_per_probe": its if its is not None else int(nit),  # iterations every probe actually ran
            "semantics": ("warm-started probes (opt-in, NOT the reference's search): later probes continue from the previous probe's iterate and run "
                          "ceil(nit/3) iterations" if warm else
                          "reference: every probe restarts from Y = 1/C, X = I and runs nit iterations (mmw.py:62-68)"),
            "score": score, "state": "device-resident (mmw_create_from_env)" if device_state else "host CSR (mmw_create)",
            "warm_start": bool(warm), "mids": [int(x) for x in per[:, 5]], "rems": [int(x) for x in per[:, 7]],
            "iterations": its,
            "speculation": ({"probes_solved_ahead_and_dropped": int(bs.LOGGED_NP_DATA["bs_speculation"][0, 4])}
                            if speculate and "bs_speculation" in bs.LOGGED_NP_DATA else None),
            "per_probe_ms": dict({"solve": [ms(x) for x in per[:, 8]], "rounding": [ms(x) for x in per[:, 9]]},
                                 **({} if speculate else {"state_process": [ms(x) for x in lg["mmw_state_process"][:, 5]],
                                                          "factor": [ms(x) for x in lg["mmw_xavg"][:, 5]]}))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)   # nit = 150 is the reference's production setting
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="journal-1pct", choices=sorted(WORKLOADS))
    ap.add_argument("--instances-per-gpu", type=int, default=None,
                    help="resident handles per rank (default 1; 8 for `--workload er-5pct-2k --gpus 8`, BASELINE configs[3])")
    ap.add_argument("--repeats", type=int, default=7, help="timed regions (each: reset, W untimed warm-up steps, K timed steps); value = median")
    ap.add_argument("--instance-threads", action="store_true", help="(default with --instances-per-gpu M > 1) one host thread per instance")
    ap.add_argument("--no-instance-threads", action="store_true", help="with --instances-per-gpu M: one host thread enqueues all instances round-robin")
    ap.add_argument("--expm", default="lanczos", choices=["lanczos", "taylor"])
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"])
    ap.add_argument("--eta", type=float, default=0.04)
    ap.add_argument("--cpu-iters", type=int, default=-1, help="oracle iterations for cpu_baseline (-1: auto, 0: skip)")
    ap.add_argument("--no-coloring", action="store_true", help="skip the binary search to a converged colouring")
    ap.add_argument("--coloring-nit", type=int, default=150)
    ap.add_argument("--coloring-speculate", action="store_true", help="two probes in flight on two handles (binary_search.speculate; opt-in)")
    ap.add_argument("--coloring-cold", action="store_true", help="(kept for compatibility: the cold, reference-semantics search is always reported as `coloring`)")
    ap.add_argument("--no-coloring-warm", action="store_true", help="skip the additional warm-started search (`coloring_warm_start`)")
    ap.add_argument("--no-fp32-operands", action="store_true", help="skip the second handle with strictly-fp32 operands (`value_fp32_operands`)")
    ap.add_argument("--state", default="host", choices=["device", "host"],
                    help="journal workloads: 'device' = the instance is generated on the GPU and handed to the solver there "
                         "(mmw_env_create -> mmw_create_from_env); 'host' (default, as in rounds 1-2) = the NumPy generator's CSR through mmw_create; the colouring block reports both")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --single-device rehearses N>1 on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: be the launcher.  Nothing above this line has touched the GPU.
        rc, out0 = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        sys.stdout.write(out0)
        sys.stdout.flush()
        raise SystemExit(rc)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.instances_per_gpu is None:
        args.instances_per_gpu = 8 if (args.workload == "er-5pct-2k" and args.gpus == 8) else 1  # configs[3]: 64 graphs, 8 per GPU
    M = max(1, args.instances_per_gpu)

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
    from sig_sdp_mmw_amd import sharding

    desc, factory, Zfix, dt_default = WORKLOADS[args.workload]
    dtype_name = args.dtype or dt_default
    dtype = _lib.F32 if dtype_name == "f32" else _lib.F64
    w = 4 if dtype_name == "f32" else 8
    n_inst = world * M
    mine = sharding.instances_of_rank(n_inst, rank, world)  # instance i -> rank i % world; its seed is its id
    made = [make_state(*factory(i), geometry=True) for i in mine]
    states = [m[0] for m in made]
    geometry0 = made[0][1]
    Zs = [Zfix if Zfix is not None else first_midpoint(st) for st in states]
    nit = args.warmup + args.steps
    method = _lib.EXPM_LANCZOS if args.expm == "lanczos" else _lib.EXPM_TAYLOR
    solvers = []
    state_mode = "host CSR (mmw_create)"
    for (st, geo), Z in zip(made, Zs):
        if args.state == "device" and geo is not None:  # the journal generator on the device, the state handed over there (f2)
            from sig_sdp_mmw_amd.graphs import min_sinr_dec, _NOISE_FLOOR_DBM
            env_i = _lib.DeviceEnv(geo["sta_locs"], geo["ap_locs"], min_sinr=min_sinr_dec(), noise_floor_dbm=_NOISE_FLOOR_DBM, device=local_rank)
            s = _lib.Solver.from_env(env_i, Z, nit, args.eta, dtype=dtype)
            env_i.close()  # the handle keeps nothing of the generator
            state_mode = "device-resident (mmw_env_create -> mmw_create_from_env)"
        else:
            s = _lib.Solver(Z, st, nit, args.eta, dtype=dtype, device=local_rank)
        s.set_expm(method, 12, 1e-6 if dtype_name == "f32" else 1e-9)
        solvers.append(s)
    solver, state, Z = solvers[0], states[0], Zs[0]
    seeds = [1234 + i for i in mine]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n):
        """n iterations of every resident instance; with several instances the calls are interleaved in chunks so that
        their streams overlap on the device (every mmw_iterate returns once its work is enqueued)."""
        if len(solvers) == 1:
            solvers[0].iterate(n, None, seeds[0])
            solvers[0].sync()
        elif not args.no_instance_threads:
            # one host thread per resident instance: the launches of a step are ~20 small kernels, and one thread enqueues ~10 k
            # iterations per second whatever the GPU could overlap (the library calls release the GIL).  Measured on one MI355X:
            # er-5pct-2k x 8: 7 230 -> 7 680 it/s; journal-native x 4: 22 800 -> 26 500 it/s
            def one(s, sd):
                s.iterate(n, None, sd)
                s.sync()
            ths = [threading.Thread(target=one, args=(s, sd)) for s, sd in zip(solvers, seeds)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        else:
            chunk = 16
            for s0 in range(0, n, chunk):
                for s, sd in zip(solvers, seeds):
                    s.iterate(min(chunk, n - s0), None, sd)
            for s in solvers:
                s.sync()

    # ---- the timed regions.  Each one: back to the initial point, `warmup` untimed steps, then exactly `steps` iterations of every
    # instance between two barriers (+ device synchronisation), closed by the gather of the per-instance objective records.  The region is
    # short (2 ms at the driver's 20 steps), so it is repeated and the line reports the median with its spread.
    def gather(recs):
        return sharding.gather_records(recs, n_inst, rank, world, dist=dist, device=coll_dev if dist is not None else None)

    def timed_region(group, seeds_g):
        """-> (seconds of the whole region, seconds of its tail = records + gather + barrier, table)"""
        for s in group:
            s.reset(nit)
        run_group(group, seeds_g, args.warmup)
        for s in group:
            s.read(_lib.F_E_MAX)  # the read-back of the objective record, once untimed (also waits for the warm-up)
        barrier()
        t0 = time.perf_counter()
        run_group(group, seeds_g, args.steps)
        tA = time.perf_counter()
        wall_us = (tA - t0) * 1e6
        # per-instance objective record, gathered to every rank (RCCL all_gather over xGMI, 48 B per instance)
        recs = [[i, Zi, 0.0, float(s.read(_lib.F_E_MAX)[0]), args.steps, wall_us] for i, Zi, s in zip(mine, Zs, group)]  # max_c e_c(X), reduced on the device
        table = gather(recs)
        assert table.shape[0] == n_inst, table.shape
        barrier()
        t1 = time.perf_counter()
        el, tail = t1 - t0, t1 - tA
        if dist is not None:  # the slowest rank's clock, after the region
            tt = torch.tensor([el, tail], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el, tail = float(tt[0].item()), float(tt[1].item())
        return el, tail, table

    def run_group(group, seeds_g, n):
        if group is solvers:
            return run_steps(n)
        for s, sd in zip(group, seeds_g):
            s.iterate(n, None, sd)
        for s in group:
            s.sync()

    # the first collective of a shape sets its transport up: three untimed gathers (on one rank the first RCCL gather took 2.9 ms, the
    # second 0.28 ms, later ones 0.06-0.09 ms; the barrier 17 / 0.1 / 0.04 ms)
    for _ in range(3):
        gather([[i, Zi, 0.0, 0.0, 0, 0.0] for i, Zi in zip(mine, Zs)])
        barrier()
    reps = max(1, args.repeats)
    regions = [timed_region(solvers, seeds) for _ in range(reps)]
    order = sorted(range(reps), key=lambda r: regions[r][0])
    elapsed, tail_s, table = regions[order[reps // 2]]  # the median region
    el_all = [r[0] for r in regions]
    if os.environ.get('BENCH_DEBUG'):
        print('[bench] regions (us): ' + ' '.join('%.0f(tail %.0f)' % (r[0] * 1e6, r[1] * 1e6) for r in regions), file=sys.stderr)
    info = solver.read(_lib.F_EXPM_INFO)
    m_used = int(info[1])

    # ---- roofline pass: the same number of steps again with HIP events around every kernel class (first instance)
    solver.reset(nit)
    solver.iterate(args.warmup, None, seeds[0])
    solver.set_profile(True)
    solver.iterate(args.steps, None, seeds[0])
    kt = solver.kernel_times()
    solver.set_profile(False)
    spmm_sync_us, spmm_sync_n = kt["spmm"]
    # the same brackets around the shipped path (chunks without plan readback: softmax inside the violation pass, lagged plans,
    # sketch riding in the LOSS launch, the exponential as one first-order product): what a step of the timed region consists of
    solver.reset(nit)
    solver.iterate(args.warmup, None, seeds[0])
    solver.set_profile(2)
    first0 = float(solver.read(_lib.F_DUAL_INFO)[2])
    first16_0 = float(solver.read(_lib.F_DUAL_INFO)[3])
    solver.iterate(args.steps, None, seeds[0])
    kt2 = solver.kernel_times()
    solver.set_profile(False)
    first_iters = int(float(solver.read(_lib.F_DUAL_INFO)[2]) - first0)  # steps of this pass whose exponential was ONE first-order product
    first16_iters = int(float(solver.read(_lib.F_DUAL_INFO)[3]) - first16_0)  # ... with the matrix in one fp16 half
    # the roofline figure is the SpMM of the path the timed region runs (launches of the shipped path, HIP events on the solver's stream)
    spmm_us, spmm_n = kt2["spmm"] if kt2["spmm"][1] else kt["spmm"]
    K, D, nnzL, C = solver.K, solver.D, solver.nnzL, solver.C
    b_spmm = nnzL * (w + 4) + (K + 1) * 4 + 2 * K * D * w
    # launches that do work: a chunk enqueued without plan readback may launch a spare Lanczos stage that the device-side estimate
    # skips (it returns at once, ~2 us); the synchronous pass counts exact launches for the same steps.  The skipped launches' time
    # stays in the numerator: the figure is the SpMM time of the shipped path per working launch.
    spmm_work = min(spmm_n, spmm_sync_n) if spmm_sync_n else spmm_n
    spmm_avg_us = spmm_us / max(spmm_work, 1)
    achieved = b_spmm / (spmm_avg_us * 1e-6) / 1e9 if spmm_work else 0.0
    traffic = None  # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/)
    for fn in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if fn.endswith(".json") and "pmc_traffic" in fn:
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if rec.get("workload") == args.workload and dtype_name == "f32" and args.expm == "lanczos":
                    traffic = rec.get("traffic_bytes_per_launch_first_order") if first_iters else None
                    if first16_iters * 2 >= args.steps and rec.get("traffic_bytes_per_launch_first_order_one_half"):
                        traffic = rec["traffic_bytes_per_launch_first_order_one_half"]
                    if traffic is None:
                        traffic = rec["traffic_bytes_per_launch"]
                    traffic_src = "profiles/" + fn
            except Exception:
                pass
    kinfo = solver.spmm_kernel_info()
    # the same kernel on the same matrix, 30 launches inside ONE pair of events: the bracket of the per-launch measurement above
    # (event markers + the dispatch after a host synchronisation, ~5 us) amortised; this is the figure rocprofv3's mean agrees with
    b2b_us = None
    try:
        os.environ["MMW_BENCH_LANCZOS"] = "1"
        b2b_us = solver.bench_spmm({0: 0, 1: 1, 2: 1, 3: 2}[int(solver.read(_lib.F_SPMM_KIND)[0])], 30)  # generic / LDS-staged / matrix cores
    except Exception:
        b2b_us = None
    finally:
        os.environ.pop("MMW_BENCH_LANCZOS", None)
    b2b_first_us = None  # ... and the first-order product (fp16 operands, its whole epilogue) the same way, when the timed region ran on it
    if first_iters and int(solver.read(_lib.F_SPMM_KIND)[0]) == 3:
        try:
            os.environ["MMW_BENCH_FIRST"] = "1"
            b2b_first_us = solver.bench_spmm(2, 30)
        except Exception:
            b2b_first_us = None
        finally:
            os.environ.pop("MMW_BENCH_FIRST", None)
    spmm_kind = int(solver.read(_lib.F_SPMM_KIND)[0])
    if dtype_name == "f64":
        operand_precision = "fp64 values, operands and accumulation"
    elif spmm_kind == 3:
        operand_precision = ("state (L, X, Y, sketch) in fp32; matrix-core products on 16-bit operands with fp32 accumulation: " +
                             ("exp(L/2)R as one first-order product y = u + (L/2 - mu I)u in %d of %d steps, u read as ONE fp16 plane (identity term from the fp32 u), "
                              "L as fp16 hi+lo of 2^20 L (one fp16 half in %d steps), certified per column against tol 1e-6 incl. the measured fp16 rounding; "
                              % (first_iters, args.steps, first16_iters) if first_iters else "") +
                             "Lanczos-step products on bf16 hi+lo of L and u (3 partial products); X on the pattern from bf16 hi+lo planes of y; "
                             "`value_fp32_operands` is the same run with strictly-fp32 operands (MMW_NO_MFMA=1)")
    else:
        operand_precision = "fp32 values, operands and accumulation (no matrix-core product on this pattern)"
    roofline = {"bound": "hbm", "kernel": kinfo["name"] + " of the %s step" % args.expm +
                ("; first-order epilogue (y = u + (L/2 - mu I)u, the whole exponential) in %d of %d steps" % (first_iters, args.steps) if first_iters else ""),
                "limiter": kinfo["limiter"],
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src if traffic is not None else None,
                "bytes_per_launch": int(b_spmm), "avg_launch_us": round(spmm_avg_us, 2),
                "timing": ("HIP start/stop events carried by each launch (hipExtLaunchKernelGGL)" if kinfo["name"].startswith("k_spmm_mfma") and not os.environ.get("MMW_KT_MARKERS")
                           else "HIP event markers around each launch"),
                "avg_launch_us_back_to_back": None if b2b_us is None else round(b2b_us, 2),  # the Lanczos-epilogue launch (two bf16 planes)
                "avg_launch_us_back_to_back_first_order": None if b2b_first_us is None else round(b2b_first_us, 2),
                "frac_back_to_back": None if not (b2b_first_us or b2b_us) else round(b_spmm / ((b2b_first_us or b2b_us) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "avg_launch_us_synchronous": round(spmm_sync_us / max(spmm_sync_n, 1), 2), "launches": int(spmm_work),
                "launches_enqueued": int(spmm_n), "launches_per_step": round(spmm_work / max(args.steps, 1), 2)}
    phases_sync = {k: round(v[0] / max(args.steps, 1), 2) for k, v in kt.items() if v[1]}
    phases = {k: round(v[0] / max(args.steps, 1), 2) for k, v in kt2.items() if v[1]}

    out = {
        "metric": "mmw_iterations_per_sec", "value": round(n_inst * args.steps / elapsed, 2), "unit": "it/s",
        "value_min": round(n_inst * args.steps / max(el_all), 2), "value_max": round(n_inst * args.steps / min(el_all), 2),
        "timed_regions": reps, "value_is": "median over the timed regions (each: reset, warm-up, K steps between barriers)",
        # the tail of the median region (objective read-back, gather of the records, closing barrier), and the rate without it:
        # at N > 1 and short regions the difference is what the collective costs
        "region_tail_ms": round(tail_s * 1e3, 4), "value_without_tail": round(n_inst * args.steps / max(elapsed - tail_s, 1e-9), 2),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": args.workload, "description": desc, "K": K, "Z": Z, "D": D, "nnzL": nnzL, "C": C,
                   "eta": args.eta, "expm": args.expm, "krylov_order": m_used, "first_order_steps": first_iters, "first_order_one_half_matrix_steps": first16_iters, "rng": "device-philox4x32", "state": state_mode,
                   "operand_precision": operand_precision,
                   "instances": n_inst, "instances_per_gpu": M,
                   "parallelism": "instance-sharded x%d" % world + (", %d resident per GPU" % M if M > 1 else "")},
        "instances_per_s": round(n_inst / elapsed, 3),  # solves of `steps` iterations per second, whole job
        "roofline": roofline,
        "device_us_per_step": phases,
        "device_us_per_step_synchronous": phases_sync,  # the profiling mode the roofline figure comes from: every class in launches of its own
        "objectives": {"max_violation_per_instance": [round(float(x), 6) for x in table[:, 3]]},
    }

    # ---- wall-clock to a converged colouring: the whole binary search on the same instance (rank 0, N = 1 only)
    # ---- the same timed regions with strictly-fp32 operands (no 16-bit split anywhere: MMW_NO_MFMA=1 is read at handle creation)
    if rank == 0 and world == 1 and M == 1 and dtype_name == "f32" and spmm_kind == 3 and not args.no_fp32_operands:
        os.environ["MMW_NO_MFMA"] = "1"
        try:
            s32 = _lib.Solver(Z, state, nit, args.eta, dtype=dtype, device=local_rank)
        finally:
            os.environ.pop("MMW_NO_MFMA", None)
        s32.set_expm(method, 12, 1e-6)
        r32 = sorted(timed_region([s32], seeds[:1])[0] for _ in range(min(reps, 3)))
        e32 = r32[len(r32) // 2]
        k32 = s32.spmm_kernel_info()["name"]
        s32.reset(nit)
        s32.iterate(args.warmup, None, seeds[0])
        s32.set_profile(2)
        s32.iterate(args.steps, None, seeds[0])
        kt32 = s32.kernel_times()
        s32.set_profile(False)
        us32 = kt32["spmm"][0] / max(kt32["spmm"][1], 1)
        out["value_fp32_operands"] = round(args.steps / e32, 2)
        out["fp32_operands"] = {"value": round(args.steps / e32, 2), "unit": "it/s", "ms_per_step": round(e32 / args.steps * 1e3, 4), "spmm_kernel": k32,
                                "spmm_avg_launch_us": round(us32, 2), "spmm_launches_per_step": round(kt32["spmm"][1] / max(args.steps, 1), 2),
                                "spmm_roofline_frac": round(b_spmm / (us32 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us32 > 0 else None}
        s32.close()

    if rank == 0 and world == 1 and M == 1 and not args.no_coloring:
        for s in solvers[1:]:
            s.close()
        # reference semantics first (cold probes): THE colouring figure; the warm-started search is an opt-in variant, labelled as such
        out["coloring"] = coloring_block(state, dtype_name, args.coloring_nit, args.eta, warm=False, geometry=geometry0, speculate=args.coloring_speculate)
        if geometry0 is not None:  # the same search with the state handed over on the device (f2): no host round trip of S / Q, no host pattern build
            out["coloring_device_state"] = coloring_block(state, dtype_name, args.coloring_nit, args.eta, warm=False, geometry=geometry0, device_state=True)
        if not args.no_coloring_warm:
            out["coloring_warm_start"] = coloring_block(state, dtype_name, args.coloring_nit, args.eta, warm=True, geometry=geometry0,
                                                        device_state=geometry0 is not None)

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
    for s in solvers:
        s.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
