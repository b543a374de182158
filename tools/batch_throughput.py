"""Developer tool: BASELINE configs[3] on one GPU -- a batch of independent N=2000 instances, 8 resident handles,
their iterations enqueued round-robin on 8 HIP streams vs one after the other (run on the GPU box)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sig_sdp_mmw_amd import _lib
from sig_sdp_mmw_amd.graphs import er_contention_graph

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 150
Z = 32
states = [er_contention_graph(2000, 0.05, seed=100 + i) for i in range(n_inst)]
hs = [_lib.Solver(Z, st, nit, 0.04, dtype=_lib.F64) for st in states]
for h in hs:  # warm
    h.iterate(5, None, 1); h.sync(); h.reset(nit)
t0 = time.perf_counter()
for h in hs:
    h.iterate(nit, None, 1)
    h.sync()
t_seq = time.perf_counter() - t0
for h in hs:
    h.reset(nit)
t0 = time.perf_counter()
chunk = 16
for s in range(0, nit, chunk):
    for h in hs:
        h.iterate(min(chunk, nit - s), None, 1)   # asynchronous: the 8 streams overlap on the device
for h in hs:
    h.sync()
t_con = time.perf_counter() - t0
print(json.dumps({"instances": n_inst, "nit": nit, "sequential_s": round(t_seq, 4), "concurrent_s": round(t_con, 4),
                  "instances_per_s_sequential": round(n_inst / t_seq, 2), "instances_per_s_concurrent": round(n_inst / t_con, 2),
                  "iterations_per_s_concurrent": round(n_inst * nit / t_con, 1)}))
