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


@pytest.mark.parametrize("kind", ["journal", "journal-dense", "er", "two-components", "tiny"])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_locality_blocking_invariants_on_the_host(kind, dtype, monkeypatch):
    """csrc/blocking.h: grown row blocks, parity-ordered chunks and parity-paired SDDMM slots, checked by the library's own
    verifier on a host-only handle (every row in one block, unions cover the columns, entries a permutation, budgets,
    complementary parities) -- no GPU needed."""
    import scipy.sparse as sp
    from sig_sdp_mmw_amd.graphs import er_contention_graph, journal_graph
    if kind == "journal":
        state = journal_graph(10, 0.02, seed=1)
    elif kind == "journal-dense":
        state = journal_graph(6, 0.3, seed=2)
    elif kind == "er":
        state = er_contention_graph(400, 0.05, 5)
    elif kind == "tiny":
        state = er_contention_graph(12, 0.5, 1)
    else:  # two disconnected geometric components
        a, b = journal_graph(6, 0.05, seed=3), journal_graph(5, 0.05, seed=4)
        S = sp.block_diag([a[0], b[0]], format="csr")
        Q = sp.block_diag([a[1], b[1]], format="csr")
        state = (S, Q, np.concatenate([a[2], b[2]]))
    monkeypatch.setenv("MMW_CHECK_BLOCKING", "1")
    s = _lib.Solver(6, state, 3, 0.05, dtype=_lib.F32 if dtype == "f32" else _lib.F64, device=-1)  # raises MMWError on a violated invariant
    assert s.K == state[0].shape[0]
    s.close()
