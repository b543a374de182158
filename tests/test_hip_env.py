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


# ---- f2, the hand-over: the solver handle built on the device straight from the generator (mmw_create_from_env) ----------------
I_FIELDS = ["I_L_INDPTR", "I_L_INDICES", "I_ST_INDPTR", "I_ST_INDICES", "I_GAIN_X", "I_GAIN_Y", "I_ASSO_X", "I_ASSO_Y", "I_DIAG_POS", "I_ASSO_POS"]
LOOP_FIELDS = ["F_E_THIS", "F_E_ACCU", "F_Y", "F_LVAL", "F_XVAL", "F_XAVG", "F_YAVG", "F_XHALF"]


def _relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


@pytest.mark.parametrize("cell,rho,Z,dtype", [(10, 75e-4, 12, "f64"), (16, 0.02, 24, "f32"), (28, 0.0319, 186, "f32")])
def test_handle_from_the_generator_is_the_handle_from_its_host_state(cell, rho, Z, dtype, monkeypatch):
    """mmw_create_from_env against mmw_create on mmw_env_state's arrays: every list and pattern field equal, S_sum / norm_H / S_T' data
    bit-identical, and -- with the blocking forced to the pattern-only order of the CSR entry point (MMW_ENV_RCM=1) -- three device-RNG
    iterations equal bit for bit.  With its own spatial row blocks the handle agrees to rounding."""
    from sig_sdp_mmw_amd import _lib
    dt = _lib.F64 if dtype == "f64" else _lib.F32
    state, env = journal_graph_device(cell, rho, 0)
    nit = 3
    a = _lib.Solver(Z, state, nit, 0.04, dtype=dt)
    a.iterate(nit, None, seed=5)
    ref = {f: a.read(getattr(_lib, f)) for f in LOOP_FIELDS}
    for variant in ("rcm", "spatial"):
        if variant == "rcm":
            monkeypatch.setenv("MMW_ENV_RCM", "1")
        else:
            monkeypatch.delenv("MMW_ENV_RCM")
        b = _lib.Solver.from_env(env, Z, nit, 0.04, dtype=dt)
        assert (b.K, b.Z, b.D, b.nnzL, b.nnzST, b.E_gain, b.E_asso, b.C) == (a.K, a.Z, a.D, a.nnzL, a.nnzST, a.E_gain, a.E_asso, a.C)
        for f in I_FIELDS:
            assert np.array_equal(a.read_i32(getattr(_lib, f)), b.read_i32(getattr(_lib, f))), f
        for f in ("F_S_SUM", "F_NORM_H", "F_ST_DATA"):
            assert np.array_equal(a.read(getattr(_lib, f)), b.read(getattr(_lib, f))), f
        b.iterate(nit, None, seed=5)
        for f in LOOP_FIELDS:
            got = b.read(getattr(_lib, f))
            if variant == "rcm":
                assert np.array_equal(got, ref[f]), f
            else:
                assert _relerr(got, ref[f]) < (1e-9 if dtype == "f64" else 2e-5), f
        if variant == "spatial" and cell == 28:  # the spatial row blocks are at least as local as the grown ones
            assert b.read(_lib.F_BLOCKING)[0] == 1.0 and b.read(_lib.F_SPMM_KIND)[0] == 3.0
        # another slot count on the same handle: the Z-dependent scalars follow (norm_H from the cached squared row sums)
        a.set_slots(Z - 3, nit)
        b.set_slots(Z - 3, nit)
        assert np.array_equal(a.read(_lib.F_NORM_H), b.read(_lib.F_NORM_H))
        a.set_slots(Z, nit)
        b.close()
    a.close()
    env.close()


def test_rounding_on_a_handle_from_the_generator():
    """sdp_solver.rounding_one_attempt on the device-built handle: exactly the host-built handle's slots (same S_gain lists, same order)."""
    from sig_sdp_mmw_amd import _lib
    state, env = journal_graph_device(16, 0.02, 0)
    Z = 24
    a = _lib.Solver(Z, state, 1, 0.04, dtype=_lib.F32)
    b = _lib.Solver.from_env(env, Z, 1, 0.04, dtype=_lib.F32)
    rng = np.random.default_rng(2)
    K = a.K
    for Zr in (Z, 5):
        gX = rng.standard_normal((K, 2 * (Z - 1)))
        rv = rng.standard_normal((3, Zr, gX.shape[1]))
        rv /= np.linalg.norm(rv, axis=2, keepdims=True)
        za, ra = a.round(Zr, gX, rv)
        zb, rb = b.round(Zr, gX, rv)
        assert np.array_equal(za, zb) and np.array_equal(ra, rb)
    a.close(); b.close(); env.close()


def test_bisection_on_a_device_resident_state():
    """The whole search with the state never materialised on the host: DeviceEnv.device_state() -> bounds from the device's count pass
    (equal to set_bounds on the host matrices), handles by mmw_create_from_env, a feasible colouring scored by the same generator."""
    from sig_sdp_mmw_amd import _lib
    from sig_sdp_mmw_amd.binary_search import binary_search_relaxation
    from sig_sdp_mmw_amd.mmw import mmw
    state, env = journal_graph_device(16, 0.02, 0)
    assert binary_search_relaxation().set_bounds(state) == env.bounds()
    dstate = env.device_state()
    bs = binary_search_relaxation()
    bs.verbose = False
    alg = mmw(nit=60, eta=0.04, dtype="f32", rng="device", seed=1)
    bs.feasibility_check_alg = alg
    np.random.seed(0)
    z_vec, Z, rem = bs.run(dstate)
    assert dstate._host is None, "the search must not have pulled the state to the host"
    assert rem == 0 and Z >= env.bounds()[0]
    S, Q, h = state
    So = S.copy().tolil(); So.setdiag(0); So = So.tocsr()
    for zz in range(Z):
        mem = np.where(z_vec == zz)[0]
        assert np.all(np.asarray(So[mem][:, mem].sum(axis=0)).ravel() <= h[mem] + 1e-12) and Q[mem][:, mem].nnz == 0
    sinr, bler = env.evaluate(z_vec, Z)
    assert np.all(np.isfinite(bler))
    alg.close(); env.close()
