"""Developer tool: wall-clock of the whole binary search to a converged colouring (run on the GPU box)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, make_state
from sig_sdp_mmw_amd.binary_search import binary_search_relaxation
from sig_sdp_mmw_amd.mmw import mmw

name = sys.argv[1] if len(sys.argv) > 1 else "journal-1pct"
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 150
desc, factory, Zfix, dt = WORKLOADS[name]
kind, kw = factory(0)
state = make_state(kind, kw)
bs = binary_search_relaxation()
bs.verbose = False
alg = mmw(nit=nit, eta=0.04, dtype=dt, rng="device", seed=1)
bs.feasibility_check_alg = alg
np.random.seed(0)
t0 = time.perf_counter()
z_vec, Z, rem = bs.run(state)
t1 = time.perf_counter()
per = bs.LOGGED_NP_DATA["bs_search_per_it"]
print(json.dumps({"workload": name, "K": int(state[0].shape[0]), "nit": nit, "wall_s": round(t1 - t0, 3), "Z": int(Z), "rem": int(rem),
                  "solves": int(per.shape[0]), "mids": [int(x) for x in per[:, 5]], "rems": [int(x) for x in per[:, 7]],
                  "solve_s": [round(x / 1e6, 3) for x in per[:, 8]], "round_s": [round(x / 1e6, 3) for x in per[:, 9]],
                  "xavg_s": [round(x / 1e6, 3) for x in alg.LOGGED_NP_DATA["mmw_xavg"][:, 5]],
                  "state_s": [round(x / 1e6, 3) for x in alg.LOGGED_NP_DATA["mmw_state_process"][:, 5]]}))
