"""The problem generator and the scorer on the device (mmw_env_*, csrc/env_device.h) against the reference's own outputs:
states of `env.generate_S_Q_hmax` (tests/golden/run_*.npz) and `env.evaluate_sinr / evaluate_bler` (tests/golden/eval.npz).
Patterns and integer data must be equal; float64 values within 1e-12 relative (device log10 / pow / erfc vs NumPy's)."""
import numpy as np
import pytest
import scipy.sparse

from conftest import load_golden, state_from
from sig_sdp_mmw_amd import scorer
from sig_sdp_mmw_amd.graphs import journal_graph, journal_graph_device

pytestmark = pytest.mark.gpu


def same_csr(a, b, rtol):
    a = scipy.sparse.csr_matrix(a); b = scipy.sparse.csr_matrix(b)
    a.sort_indices(); b.sort_indices()
    assert a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    np.testing.assert_allclose(a.data, b.data, rtol=rtol, atol=0)


@pytest.mark.parametrize("name,cell,seed", [("env75", 5, 0), ("env300", 10, 0), ("env192", 8, 3)])
def test_device_generator_matches_the_reference_states(name, cell, seed):
    g = load_golden("run_" + name)
    S0, Q0, h0 = state_from(g)
    (S, Q, h), env = journal_graph_device(cell, 75e-4, seed)
    same_csr(S, S0, 1e-12)
    same_csr(Q, Q0, 0)
    np.testing.assert_allclose(h, h0, rtol=1e-12)
    assert S.has_sorted_indices and Q.has_sorted_indices
    env.close()


def test_device_generator_at_the_benchmark_size():
    (S, Q, h), env = journal_graph_device(28, 0.0319, 0)
    S0, Q0, h0 = journal_graph(28, 0.0319, 0)  # the NumPy restatement, itself bit-identical to the reference on the fixtures
    same_csr(S, S0, 1e-12)
    same_csr(Q, Q0, 0)
    np.testing.assert_allclose(h, h0, rtol=1e-12)
    env.close()


def assert_scores_match(asso, z, Z, sinr, bler, sinr_ref, bler_ref):
    """Equality up to the one freedom the reference itself has.  Users of one AP are power-controlled to the SAME receive
    power, so when several of them share a slot their SINRs agree to the last bit or two and the reference's "strongest
    survives" (env.py:214-224) is decided by the rounding of libm's pow -- not reproducible on another math library.  Every
    (AP, slot) group must therefore hold the same VALUES (one survivor, the rest at the floor); users alone in their group
    must match one by one."""
    key = asso.astype(np.int64) * (Z + 1) + np.where((z >= 0) & (z < Z), z, Z).astype(np.int64)
    order = np.lexsort((np.arange(key.size), key))
    bounds = np.flatnonzero(np.r_[True, key[order][1:] != key[order][:-1], True])
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        idx = order[lo:hi]
        np.testing.assert_allclose(np.sort(sinr[idx]), np.sort(sinr_ref[idx]), rtol=1e-10)
        np.testing.assert_allclose(np.sort(bler[idx]), np.sort(bler_ref[idx]), rtol=1e-8, atol=1e-300)
        if idx.size == 1 or np.all(z[idx] >= Z):
            np.testing.assert_allclose(sinr[idx], sinr_ref[idx], rtol=1e-10)
    assert int(np.sum(sinr == 1e-3)) == int(np.sum(sinr_ref == 1e-3))
    np.testing.assert_allclose(np.mean(bler), np.mean(bler_ref), rtol=1e-9)
    np.testing.assert_allclose(np.max(bler), np.max(bler_ref), rtol=1e-9)


@pytest.mark.parametrize("name", ["env75", "env108"])
def test_device_scorer_matches_the_reference(name):
    g = load_golden("eval")
    cell, seed = int(g[name + "_cell_size"]), int(g[name + "_seed"])
    (S, Q, h), env = journal_graph_device(cell, 75e-4, seed)
    _, geo = journal_graph(cell, 75e-4, seed, return_geometry=True)
    K = S.shape[0]
    for suffix, z, Z in (("", g[name + "_z_vec"], int(g[name + "_Z"])), ("_bad", (np.arange(K) % 3).astype(float), 3)):
        sinr, bler = env.evaluate(z, Z)
        assert_scores_match(geo["asso"], np.asarray(z), Z, sinr, bler, g[name + "_sinr" + suffix], g[name + "_bler" + suffix])
    env.close()


def test_device_scorer_at_the_benchmark_size_against_the_host_restatement():
    (S, Q, h), env = journal_graph_device(28, 0.0319, 0)
    state, geo = journal_graph(28, 0.0319, 0, return_geometry=True)
    rx = scorer.receive_power(geo["sta_locs"], geo["ap_locs"])
    K = S.shape[0]
    rng = np.random.default_rng(1)
    for Z in (40, 7):
        z = rng.integers(0, Z, size=K).astype(float)
        z[::97] = Z + 3  # users left outside every slot keep the floor value
        sinr, bler = env.evaluate(z, Z)
        assert_scores_match(geo["asso"], z, Z, sinr, bler, scorer.evaluate_sinr(rx, z, Z), scorer.evaluate_bler(rx, z, Z))
    env.close()
