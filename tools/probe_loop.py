"""Developer tool (GPU box): the loop time of the probes of a warm-started binary search, outside the class (no factor, no rounding)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, make_state, first_midpoint
from sig_sdp_mmw_amd import _lib

timing = len(sys.argv) > 1 and sys.argv[1] == "timing"
desc, factory, Zfix, dt = WORKLOADS["journal-1pct"]
state = make_state(*factory(0))
Z0 = first_midpoint(state)
s = _lib.Solver(Z0, state, 150, 0.04, dtype=_lib.F32)
s.set_expm(_lib.EXPM_LANCZOS, 12, 1e-6)
s.set_timing(timing)
for rep in range(2):
    if rep:
        s.set_slots(Z0, 150)
        s.set_timing(timing)
    i0 = s.read(_lib.F_DUAL_INFO).copy()
    t0 = time.perf_counter(); s.iterate(150, None, seed=5); s.sync(); t1 = time.perf_counter()
    print("cold Z=%d 150 it: %.2f ms  info %s replays %d" % (Z0, (t1 - t0) * 1e3, s.read(_lib.F_DUAL_INFO) - i0, s.read(_lib.F_BLOCKING)[3]))
    for Z in (105, 65, 45, 35, 40):
        t0 = time.perf_counter(); s.set_slots(Z, 50, warm=True); s.set_timing(timing); t1 = time.perf_counter()
        i0 = s.read(_lib.F_DUAL_INFO).copy()
        t2 = time.perf_counter(); s.iterate(50, None, seed=5); s.sync(); t3 = time.perf_counter()
        print("warm Z=%d 50 it: set_slots %.2f ms loop %.2f ms (%.0f us/it) info %s replays %d" % (Z, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t3 - t2) * 1e6 / 50,
              s.read(_lib.F_DUAL_INFO) - i0, s.read(_lib.F_BLOCKING)[3]))
