"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/mmw_hip.h declares,
and its native host-side state processing (device = -1 handle, no HIP call) matches the golden
vectors captured from the reference (mmw.py:26-60)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, csr_from, state_from
from sig_sdp_mmw_amd import _lib


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mmw_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mmw_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.EXPORTS) == names
    assert L.mmw_version() >= 100


def test_host_pattern_matches_reference(run_case):
    name, g = run_case
    state = state_from(g)
    Z = int(g["Z"])
    s = _lib.Solver(Z, state, 3, 0.1, device=-1)
    assert s.C == int(g["C"]) and s.E_asso == int(g["E_asso"])
    ST = csr_from(g, "ST")
    assert np.array_equal(s.read_i32(_lib.I_ST_INDPTR), ST.indptr)
    assert np.array_equal(s.read_i32(_lib.I_ST_INDICES), ST.indices)
    assert np.array_equal(s.read(_lib.F_ST_DATA), ST.data)
    np.testing.assert_allclose(s.read(_lib.F_S_SUM), g["S_sum"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(s.read(_lib.F_NORM_H), g["norm_H"], rtol=4e-16, atol=0)
    for f, key in [(_lib.I_GAIN_X, "nz_idx_gain_x_ut"), (_lib.I_GAIN_Y, "nz_idx_gain_y_ut"),
                   (_lib.I_ASSO_X, "nz_idx_asso_x_ut"), (_lib.I_ASSO_Y, "nz_idx_asso_y_ut")]:
        assert np.array_equal(s.read_i32(f), g[key])
    # the fixed pattern holds every L_accu / X pattern the reference produced
    indptr, indices = s.read_i32(_lib.I_L_INDPTR), s.read_i32(_lib.I_L_INDICES)
    K = s.K
    mine = set(zip(np.repeat(np.arange(K), np.diff(indptr)).tolist(), indices.tolist()))
    L0 = csr_from(g, "Laccu0").tocoo()
    assert set(zip(L0.row.tolist(), L0.col.tolist())) <= mine
    assert s.nnzL == K + 2 * (s.E_gain + s.E_asso)
    dp = s.read_i32(_lib.I_DIAG_POS)
    assert np.array_equal(indices[dp], np.arange(K))
    s.close()


def test_host_only_handle_refuses_device_work(run_case):
    name, g = run_case
    s = _lib.Solver(int(g["Z"]), state_from(g), 3, 0.1, device=-1)
    with pytest.raises(_lib.MMWError):
        s.iterate(1)
    with pytest.raises(_lib.MMWError):
        s.read(_lib.F_Y)
    s.close()


def test_malformed_state_is_rejected():
    import scipy.sparse
    from sig_sdp_mmw_amd.graphs import er_contention_graph
    S, Q, h = er_contention_graph(40, 0.2, 1)
    Qbad = Q.tolil()
    i, j = Q.nonzero()[0][0], Q.nonzero()[1][0]
    Qbad[j, i] = 0
    Qbad = scipy.sparse.csr_matrix(Qbad)
    Qbad.eliminate_zeros()
    with pytest.raises(_lib.MMWError, match="symmetric"):
        _lib.Solver(5, (S, Qbad, h), 3, 0.1, device=-1)
    with pytest.raises(_lib.MMWError, match="Z must be"):
        _lib.Solver(1, (S, Q, h), 3, 0.1, device=-1)
    with pytest.raises(_lib.MMWError):
        _lib.Solver(5, (S, Q, h[:-1]), 3, 0.1, device=-1)


def test_no_device_is_a_loud_error():
    try:
        n = _lib.device_count()
    except _lib.MMWError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    from sig_sdp_mmw_amd.graphs import er_contention_graph
    with pytest.raises(_lib.MMWError):
        _lib.Solver(5, er_contention_graph(40, 0.2, 1), 3, 0.1, device=0)
