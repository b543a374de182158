import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MMW_VERBOSE"] = "1"
from bench import WORKLOADS, make_state, first_midpoint
from sig_sdp_mmw_amd import _lib
desc, factory, Zfix, dt = WORKLOADS["journal-1pct"]
kind, kw = factory(0)
state = make_state(kind, kw)
Z = first_midpoint(state)
for i in range(3):
    t0 = time.time()
    s = _lib.Solver(Z, state, 20, 0.04, dtype=_lib.F32)
    s.sync()
    print("create %d: %.3f s" % (i, time.time() - t0), flush=True)
    s.close()
