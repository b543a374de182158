import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import WORKLOADS, make_state, first_midpoint
from sig_sdp_mmw_amd import _lib
for name in ("journal-1pct", "er-1pct", "er-5pct-2k", "dense-200", "journal-native"):
    desc, factory, Zfix, dt = WORKLOADS[name]
    kind, kw = factory(0)
    state = make_state(kind, kw)
    Z = Zfix if Zfix is not None else first_midpoint(state)
    s = _lib.Solver(Z, state, 160, 0.04, dtype=_lib.F32 if dt == "f32" else _lib.F64)
    s.iterate(10, None, 1); s.sync()
    t0 = time.perf_counter(); s.iterate(150, None, 1); s.sync(); dt_ = time.perf_counter() - t0
    print(name, "it/s %.0f" % (150 / dt_), "replays", int(s.read(_lib.F_BLOCKING)[3]), "order", s.read(_lib.F_EXPM_INFO)[1])
    s.close()
