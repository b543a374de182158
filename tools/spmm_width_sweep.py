"""Developer tool: SpMM time vs block width D on the bench graph (run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, make_state
from sig_sdp_mmw_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "journal-1pct"
desc, factory, Zfix, dt = WORKLOADS[name]
kind, kw = factory(0)
state = make_state(kind, kw)
for Z in (2, 4, 8, 16, 32, 64, 128, 186):
    for dtype, w in ((_lib.F32, 4), (_lib.F64, 8)):
        s = _lib.Solver(Z, state, 4, 0.04, dtype=dtype)
        s.iterate(2, None, 1)
        b = s.nnzL * (w + 4) + (s.K + 1) * 4 + 2 * s.K * s.D * w
        row = "Z=%3d D=%3d w=%d alg=%.1fMB" % (Z, s.D, w, b / 1e6)
        for blocked in (0, 1):
            try:
                us = s.bench_spmm(blocked, 30)
                row += "  %s %.1f us %.0f GB/s (%.1f%%)" % ("blk" if blocked else "gen", us, b / us / 1e3, b / us / 1e3 / 80)
            except _lib.MMWError:
                pass
        print(row, flush=True)
        s.close()
