"""Developer tool: time the generic and the blocked SpMM on a bench workload (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, make_state, first_midpoint
from sig_sdp_mmw_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "journal-1pct"
desc, factory, Zfix, dt = WORKLOADS[name]
kind, kw = factory(0)
state = make_state(kind, kw)
Z = Zfix if Zfix is not None else first_midpoint(state)
t0 = time.time()
s = _lib.Solver(Z, state, 20, 0.04, dtype=_lib.F32 if dt == "f32" else _lib.F64)
print("create %.3fs" % (time.time() - t0), "K", s.K, "D", s.D, "nnzL", s.nnzL, "blocking", s.read(_lib.F_BLOCKING))
s.iterate(3, None, 1)
w = 4 if dt == "f32" else 8
b = s.nnzL * (w + 4) + (s.K + 1) * 4 + 2 * s.K * s.D * w
modes = [int(x) for x in os.environ.get('MMW_BENCH_MODES', '0,1,2').split(',')]
for blocked in modes:  # 0 generic gather, 1 LDS-staged fp32, 2 matrix cores
    try:
        us = s.bench_spmm(blocked, 30)
        print("blocked=%d  %.1f us  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (blocked, us, b / us / 1e3, b / us / 1e3 / 80))
    except _lib.MMWError as e:
        print("blocked=%d" % blocked, e)
