"""Developer tool: how conservative is the a-priori Krylov order?  Runs the bench instance and, for L at a few iteration
counts, compares exp(L/2) R from the device at loosened tolerances (so the plan picks order 1, 2, 3) against scipy's
expm_multiply.  Run on the GPU box."""
import os, sys
import numpy as np
from scipy.sparse.linalg import expm_multiply
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, make_state, first_midpoint
from sig_sdp_mmw_amd import _lib
from oracle import mmw_oracle as orc

desc, factory, Zfix, dt = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "journal-1pct"]
kind, kw = factory(0)
state = make_state(kind, kw)
Z = Zfix if Zfix is not None else first_midpoint(state)
nit = 150
s = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
P = orc.Pattern(Z, state)
rng = np.random.default_rng(0)
Dw = 64  # columns are independent: a narrow block is enough for the error
R = orc.sketch_rows(rng.standard_normal((P.K, Dw)))
done = 0
for upto in (5, 20, 60, 100, 149):
    s.iterate(upto - done, None, 1); done = upto
    A = P.csr(s.read(_lib.F_LVAL)) * 0.5
    ref = expm_multiply(A, R)
    row = []
    for tol in (1e-6, 1e-4, 1e-2, 1e0, 1e2):
        out, info = _lib.expm_apply(A, R, dtype=_lib.F32, method=_lib.EXPM_LANCZOS, max_order=12, tol=tol)
        err = np.linalg.norm(out - ref) / np.linalg.norm(ref)
        row.append("tol %.0e: m=%d s=%d err %.1e" % (tol, info["order"], info["substeps"], err))
    print("it %3d  one-norm %.4f | " % (upto, info["one_norm"]) + " | ".join(row), flush=True)
