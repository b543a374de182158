"""CPU checks of host-side logic that needs no device: the generator restatement against the reference's states, the
bisection bounds, the environment defaults of the drop-in class."""
import numpy as np
import pytest
import scipy.sparse

from conftest import load_golden, state_from
from sig_sdp_mmw_amd.binary_search import binary_search_relaxation
from sig_sdp_mmw_amd.graphs import journal_graph


@pytest.mark.parametrize("name,cell,seed", [("env75", 5, 0), ("env300", 10, 0), ("env192", 8, 3)])
def test_journal_graph_is_bit_identical_to_the_reference_generator(name, cell, seed):
    """tests/golden/run_*.npz hold `env(cell_size, 75e-4, seed).generate_S_Q_hmax()` of the reference (env.py:168-196)."""
    g = load_golden("run_" + name)
    S0, Q0, h0 = state_from(g)
    S, Q, h = journal_graph(cell, 75e-4, seed)
    for a, b in ((S, S0), (Q, Q0)):
        a = scipy.sparse.csr_matrix(a); b = scipy.sparse.csr_matrix(b)
        a.sort_indices(); b.sort_indices()
        assert a.shape == b.shape
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)
    assert np.array_equal(h, h0)


def test_upper_bound_counts_the_stored_diagonal_like_setdiag():
    """binary_search_relaxation.py:21-26: S.setdiag(0) leaves one stored entry per row, present before or not."""
    g = load_golden("bs_run")
    for name in ("env75", "env108"):
        state = state_from(g, name + "_")
        want = [int(x) for x in g[name + "_bounds"][0]]
        assert list(binary_search_relaxation().set_bounds(state)) == want
        S = state[0].tolil()
        S.setdiag(0)
        S = scipy.sparse.csr_matrix(S)
        S.eliminate_zeros()  # the same gains without any stored diagonal
        assert list(binary_search_relaxation().set_bounds((S, state[1], state[2]))) == want


def test_class_defaults_follow_the_environment(monkeypatch):
    from sig_sdp_mmw_amd.mmw import mmw
    for k in ("MMW_DTYPE", "MMW_RNG", "MMW_EXPM_TOL", "MMW_WARM_START"):
        monkeypatch.delenv(k, raising=False)
    a = mmw(nit=150, eta=0.04)
    assert (a.dtype, a.rng, a.expm_tol, a.warm_start) == ("f64", "host", 1e-9, False)  # the reference's behaviour
    monkeypatch.setenv("MMW_DTYPE", "f32"); monkeypatch.setenv("MMW_RNG", "device"); monkeypatch.setenv("MMW_EXPM_TOL", "3e-6")
    monkeypatch.setenv("MMW_WARM_START", "1")
    b = mmw(nit=150, eta=0.04)
    assert (b.dtype, b.rng, b.expm_tol, b.warm_start, b.round_batch) == ("f32", "device", 3e-6, True, True)
    c = mmw(nit=150, eta=0.04, dtype="f64", rng="host", expm_tol=1e-12, warm_start=False)
    assert (c.dtype, c.rng, c.expm_tol, c.warm_start) == ("f64", "host", 1e-12, False)
    monkeypatch.setenv("MMW_DTYPE", "bf16")
    with pytest.raises(ValueError):
        mmw()
