#!/usr/bin/env python3
"""Developer tool (GPU box): where the wall-clock of one bisection probe goes on the benchmark instance -- set_slots, the loop, the
reads, the factor, the rounding -- for a few slot counts in a row, as the colouring runs them (device-resident state).
   python tools/probe_timeline.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sig_sdp_mmw_amd import _lib
from sig_sdp_mmw_amd.graphs import journal_geometry, min_sinr_dec, _NOISE_FLOOR_DBM

sta_locs, ap_locs = journal_geometry(28, 0.0319, 0)
env = _lib.DeviceEnv(sta_locs, ap_locs, min_sinr=min_sinr_dec(), noise_floor_dbm=_NOISE_FLOOR_DBM, device=0)
t0 = time.perf_counter()
s = _lib.Solver.from_env(env, 186, 150, 0.04, dtype=_lib.F32)
s.sync()
print("create %.1f ms" % ((time.perf_counter() - t0) * 1e3))
s.set_expm(_lib.EXPM_LANCZOS, 12, 1e-6)
def T():
    return time.perf_counter()
for rep, Z in enumerate([int(z) for z in os.environ.get("ZS", "186,105,65,45,35,40").split(",")]):
    a = T(); s.set_slots(Z, 150); s.sync(); b = T()
    s.set_timing(int(os.environ.get("STRIDE", "8")))
    s.iterate(150, None, seed=77 + rep); c = T()
    s.sync(); d = T()
    us = s.read(_lib.F_PHASE_US); info = s.read(_lib.F_EXPM_INFO); e = T()
    rank = min(s.K - 1, (Z - 1) * 2)
    X = s.factor(rank, seed=5); f = T()
    print("Z %3d: set_slots %.2f | enqueue %.2f | sync %.2f | reads %.2f | factor(rank %d) %.2f ms | loop per it %.1f us, replays %d" %
          (Z, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, (e - d) * 1e3, rank, (f - e) * 1e3, (d - b) * 1e6 / 150, int(s.read(_lib.F_BLOCKING)[3])))
