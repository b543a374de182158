"""GPU parity of the device rounding (mmw_round: fp64 MFMA projection + greedy assignment) with the
reference's sdp_solver.rounding_one_attempt: EXACT agreement of the integer slot assignment on
identical (gX, randv), for feasible and infeasible slot counts, plus larger random instances against
the CPU oracle."""
import numpy as np
import pytest

from conftest import load_golden, state_from
from oracle import mmw_oracle as orc
from sig_sdp_mmw_amd import _lib

pytestmark = pytest.mark.gpu


def normalise(r):
    return r / np.linalg.norm(r, axis=-1, keepdims=True)


def test_attempts_match_reference_exactly(run_case):
    name, g = run_case
    state = state_from(g)
    s = _lib.Solver(int(g["Z"]), state, 1, 0.1)
    for pre, Zk, gXk in (("att_", "round_Z", "round_gX"), ("small_att_", "small_Z", "small_gX")):
        Z = int(g[Zk])
        gX = g[gXk]
        rv = normalise(g[pre + "randv"])
        z, rem = s.round(Z, gX, rv)  # all attempts of the fixture in one batch
        for a in range(rv.shape[0]):
            assert int(rem[a]) == int(g[pre + "rem"][a]), (name, pre, a)
            zr = g[pre + "z"][a]
            un = z[a] < 0
            assert int(un.sum()) == int(rem[a])
            assert np.array_equal(z[a][~un], zr[~un].astype(np.int32))
            # the reference fills the unassigned users with its randint draw, in index order
            left = g[pre + "randint"][a]
            left = left[left >= 0]
            filled = z[a].astype(np.float64)
            filled[un] = left[:int(un.sum())]
            assert np.array_equal(filled, zr)
    s.close()


@pytest.mark.parametrize("K,p,Z,Dp,seed", [(400, 0.05, 12, 22, 1), (1000, 0.02, 40, 78, 2), (700, 0.03, 7, 12, 3),
                                            (1500, 0.01, 33, 64, 4)])
def test_random_instances_match_oracle(K, p, Z, Dp, seed):
    from sig_sdp_mmw_amd.graphs import er_contention_graph
    state = er_contention_graph(K, p, seed, hi=1.2)
    rng = np.random.default_rng(seed)
    gX = rng.standard_normal((K, Dp)) * rng.uniform(0.2, 2.0, size=(K, 1))
    rv = normalise(rng.standard_normal((3, Z, Dp)))
    s = _lib.Solver(Z, state, 1, 0.1)
    z, rem = s.round(Z, gX, rv)
    for a in range(3):
        zo, _, remo, un = orc.rounding_one_attempt(Z, gX, state, rv[a], randint=lambda Zs, size: np.full(size, -1))
        assert int(rem[a]) == remo
        assert np.array_equal(z[a], zo.astype(np.int32))
    s.close()


def test_journal_instance_matches_oracle():
    from sig_sdp_mmw_amd.graphs import journal_graph
    state = journal_graph(12, 75e-4, seed=4)  # K = 432
    K = state[0].shape[0]
    rng = np.random.default_rng(0)
    for Z in (6, 14, 40):
        Dp = min(K - 1, 2 * (Z - 1))
        gX = rng.standard_normal((K, Dp))
        rv = normalise(rng.standard_normal((2, Z, Dp)))
        s = _lib.Solver(max(Z, 2), state, 1, 0.1)
        z, rem = s.round(Z, gX, rv)
        for a in range(2):
            zo, _, remo, un = orc.rounding_one_attempt(Z, gX, state, rv[a], randint=lambda Zs, size: np.full(size, -1))
            assert int(rem[a]) == remo and np.array_equal(z[a], zo.astype(np.int32))
        # feasibility of what was assigned: per slot, interference within h_max and one user per AP
        S, Q, h = state
        Sd = S.toarray()
        np.fill_diagonal(Sd, 0)
        for a in range(2):
            for zz in range(Z):
                mem = np.where(z[a] == zz)[0]
                if mem.size:
                    assert np.all(Sd[np.ix_(mem, mem)].sum(axis=0) <= h[mem] + 1e-12)
                    assert Q[np.ix_(mem, mem)].nnz == 0
        s.close()
