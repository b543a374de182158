#!/usr/bin/env python3
"""Developer aid: which form the chunks of a device-RNG run take (MMW_VERBOSE=1 prints every plan read back)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sig_sdp_mmw_amd import _lib
from sig_sdp_mmw_amd.graphs import journal_graph
small = len(sys.argv) > 1 and sys.argv[1] == "small"
state, Z = (journal_graph(16, 0.02, seed=4), 24) if small else (journal_graph(28, 0.0319, 0), 186)
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 60
s = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
s.set_expm(_lib.EXPM_LANCZOS, 12, 1e-6)
done = 0
while done < nit:
    n = min(16, nit - done)
    s.iterate(n, None, 9)
    s.sync()
    done += n
    print(done, "dual_info", s.read(_lib.F_DUAL_INFO), "expm", s.read(_lib.F_EXPM_INFO), "replays", s.read(_lib.F_BLOCKING)[3], flush=True)
s.close()
