"""GPU parity of the LOG_GAP branch (mmw_gap), the epilogue factor (mmw_factor) and the drop-in class
end to end, against the golden vectors of the reference.

The factor has sign/rotation freedom (svds), so it is compared through X_half X_half^T
(SURVEY.md §8c fixture 4); tolerance 1e-6 relative Frobenius in fp64, 1e-4 in fp32.
"""
import numpy as np
import pytest

from conftest import csr_from, load_golden, relerr, state_from
from oracle import mmw_oracle as orc
from sig_sdp_mmw_amd import _lib
from sig_sdp_mmw_amd.mmw import mmw

pytestmark = pytest.mark.gpu


def test_gap_matches_reference():
    for name in ("env75", "env192", "dense60"):
        g = load_golden("run_" + name)
        state = state_from(g)
        Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
        s = _lib.Solver(Z, state, nit, eta)
        s.set_expm(_lib.EXPM_LANCZOS, 16, 1e-13)
        for i in range(nit):
            gap = s.gap()
            ref = g["gap"][i]
            assert abs(gap[0] - ref[0]) <= 1e-8 * abs(ref[0]) + 1e-12, (name, i, gap, ref)
            assert abs(gap[1] - ref[1]) <= 1e-6 * abs(ref[1]) + 1e-9, (name, i, gap, ref)
            assert abs(gap[2] - ref[2]) <= 1e-6 * (abs(ref[0]) + abs(ref[1])) + 1e-9
            s.iterate(1, g["randv"][i])
        with pytest.raises(_lib.MMWError):
            s.gap()  # nothing left to precede
        s.close()


def test_factor_matches_reference(run_case):
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    K = state[0].shape[0]
    rank = int(min(K - 1, (Z - 1) * 2))
    ref = orc.projector(g["X_half_ret"])
    for dtype, tolx, bar in ((_lib.F64, 1e-13, 1e-6), (_lib.F32, 1e-7, 2e-4)):
        s = _lib.Solver(Z, state, nit, eta, dtype=dtype)
        s.set_expm(_lib.EXPM_LANCZOS, 16, tolx)
        with pytest.raises(_lib.MMWError):
            s.factor(rank)  # before the iterations
        s.iterate(nit, g["randv"][:nit])
        X = s.factor(rank, seed=5)
        assert X.shape == (K, rank)
        assert relerr(orc.projector(X), ref) < bar, (name, dtype)
        # columns come in ascending |lambda| like svds' ascending singular values
        n2 = np.sum(X * X, axis=0)
        assert np.all(np.diff(n2) >= -1e-9 * n2.max())
        assert np.array_equal(s.read(_lib.F_FACTOR, K * rank).reshape(K, rank), X)
        s.close()


def test_factor_on_larger_instance_against_oracle_svds():
    from sig_sdp_mmw_amd.graphs import journal_graph
    state = journal_graph(10, 75e-4, seed=7)  # K = 300
    Z, nit, eta = 14, 12, 0.08
    K = state[0].shape[0]
    s = _lib.Solver(Z, state, nit, eta)
    s.iterate(nit, None, seed=3)
    rank = min(K - 1, 2 * (Z - 1))
    X = s.factor(rank, seed=1)
    import scipy.sparse
    xavg = scipy.sparse.csr_matrix((s.read(_lib.F_XAVG) / nit, s.read_i32(_lib.I_L_INDICES), s.read_i32(_lib.I_L_INDPTR)), shape=(K, K))
    ref = orc.factor_xavg(xavg, rank)
    assert relerr(orc.projector(X), orc.projector(ref)) < 1e-6
    s.close()


def test_class_end_to_end_seeded_like_the_reference(run_case):
    """Same seed, host-compatible RNG: the class consumes np.random exactly like the reference, so the
    whole run (loop + factor) reproduces the reference's X_half X_half^T and its stream position."""
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    K = state[0].shape[0]
    alg = mmw(nit=nit, eta=eta, expm_tol=1e-13, expm_max_order=16)
    alg.LOG_GAP = bool(int(g["log_gap"]))
    np.random.seed(int(g["seed"]))
    ok, X_half = alg.run_with_state(0, Z, state)
    assert ok is True
    assert X_half.shape == g["X_half_ret"].shape
    assert relerr(orc.projector(X_half), orc.projector(g["X_half_ret"])) < 1e-6
    assert int(np.random.get_state()[2]) == int(g["rng_pos_after"])
    # log tables: same keys and shapes as the reference's (sim_mmw_time.py:48-52 reads column 5)
    for key in ("mmw_dual", "mmw_loss", "mmw_expm", "mmw_per_it", "mmw_xavg", "mmw_all_it", "mmw_state_process", "gap"):
        k = "log_" + key + "_shape"
        if k in g:
            assert alg.LOGGED_NP_DATA[key].shape == tuple(g[k]), key
    assert np.all(alg.LOGGED_NP_DATA["mmw_expm"][:, 5] > 0)
    if alg.LOG_GAP:
        np.testing.assert_allclose(alg.LOGGED_NP_DATA["gap"][:, 3], g["gap"][:, 0], rtol=1e-7)
        np.testing.assert_allclose(alg.LOGGED_NP_DATA["gap"][:, 4], g["gap"][:, 1], rtol=1e-5, atol=1e-8)
    # rounding through the class on the reference's own gX, same seed -> identical colours
    np.random.seed(int(g["round_seed"]))
    z_vec, Zr, rem = alg.rounding(int(g["round_Z"]), g["round_gX"], state)
    assert Zr == int(g["round_Z"]) and int(rem) == int(g["round_rem"])
    assert z_vec.dtype == np.float64 and np.array_equal(z_vec, g["round_z_vec"])


def test_class_device_rng_fp32_runs_and_is_feasible():
    from sig_sdp_mmw_amd.graphs import journal_graph
    state = journal_graph(8, 75e-4, seed=1)  # K = 192
    K = state[0].shape[0]
    alg = mmw(nit=40, eta=0.05, dtype="f32", rng="device", seed=3)
    ok, X_half = alg.run_with_state(0, 12, state)
    assert X_half.shape == (K, 22) and np.all(np.isfinite(X_half))
    z_vec, Z, rem = alg.rounding(12, X_half, state)
    assert z_vec.shape == (K,) and np.all((z_vec >= 0) & (z_vec < 12))
    # the seam is still monkey-patchable and callable (mmw.py:180 looks it up on the class)
    import scipy.sparse
    L = scipy.sparse.identity(K, format="csr") * 0.01
    np.random.seed(0)
    out = mmw.expm_half_randsk(L, 6)
    assert out.shape == (K, 6)
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), np.exp(0.01), rtol=1e-10)


def test_binary_search_end_to_end_on_device():
    """bs.run with the device solver: converges to a feasible colouring within the reference's bracket."""
    from sig_sdp_mmw_amd.binary_search import binary_search_relaxation
    g = load_golden("bs_run")
    for name in ("env75", "env108"):
        state = state_from(g, name + "_")
        bs = binary_search_relaxation()
        bs.verbose = False
        alg = mmw(nit=30, eta=0.04)
        bs.feasibility_check_alg = alg
        np.random.seed(int(g[name + "_seed"]))
        z_vec, Z, rem = bs.run(state)
        lb, ub = (int(x) for x in g[name + "_bounds"][0])
        assert rem == 0 and lb <= Z <= ub
        assert abs(Z - int(g[name + "_Z"])) <= 2  # same problem, different factor rotation -> same slot count +-
        # the colouring is feasible: per slot no shared AP and interference within h_max
        S, Q, h = state
        Sd = S.toarray()
        np.fill_diagonal(Sd, 0)
        for zz in range(Z):
            mem = np.where(z_vec == zz)[0]
            if mem.size:
                assert np.all(Sd[np.ix_(mem, mem)].sum(axis=0) <= h[mem] + 1e-12)
                assert Q[np.ix_(mem, mem)].nnz == 0


def test_replayed_chunks_keep_the_phase_timers_consistent(monkeypatch):
    """Without the a-posteriori stop the a-priori order rises during the run, chunks enqueued with too few stages are
    restored and replayed, and the per-iteration phase timers (one row per iteration, like the reference's log) must still
    come out one row per iteration."""
    from sig_sdp_mmw_amd.graphs import journal_graph
    monkeypatch.setenv("MMW_NO_APOST", "1")
    state = journal_graph(8, 75e-4, seed=1)  # K = 192
    s = _lib.Solver(12, state, 40, 0.05, dtype=_lib.F32)
    s.set_timing(True)
    s.iterate(40, None, seed=3)
    s.sync()
    t = s.read(_lib.F_PHASE_US).reshape(40, 4)
    assert np.all(t > 0)
    info = s.read(_lib.F_BLOCKING)
    assert info[3] >= 1, "this configuration is expected to replay at least one chunk"
    s.close()


def _sym(b, kind, rng):
    if kind == "random":
        A = rng.standard_normal((b, b))
        return (A + A.T) / 2
    if kind == "gram":  # what the Rayleigh-Ritz step sees: nearly diagonal, decaying spectrum
        Qr, _ = np.linalg.qr(rng.standard_normal((b, b)))
        lam = np.exp(-np.arange(b) / (b / 8.0)) * np.where(rng.random(b) < 0.3, -1.0, 1.0)
        return (Qr * lam) @ Qr.T * 0.02 + np.diag(lam)
    if kind == "clustered":  # repeated eigenvalues: any basis of the eigenspace is right
        Qr, _ = np.linalg.qr(rng.standard_normal((b, b)))
        lam = np.repeat([3.0, 3.0, -1.0, 0.5], (b + 3) // 4)[:b]
        return (Qr * lam) @ Qr.T
    if kind == "diagonal":
        return np.diag(rng.standard_normal(b))
    return np.zeros((b, b))


@pytest.mark.parametrize("b", [1, 2, 31, 64, 65, 100, 250, 444])
@pytest.mark.parametrize("kind", ["random", "gram", "clustered", "diagonal", "zero"])
def test_sym_eig_matches_lapack(b, kind):
    """The Rayleigh-Ritz eigensolve (block Jacobi, kernels_dense.h) against numpy.linalg.eigh: eigenvalues, orthogonality and
    the decomposition itself, at the sizes the factor uses it (one block pair ... 14 block columns, odd block counts padded)."""
    rng = np.random.default_rng(b * 7 + len(kind))
    G = _sym(b, kind, rng)
    # repeated eigenvalues slow this ordering of rotations down to a linear rate (the one-launch-per-round solver it replaces
    # needs 20-28 sweeps on the same matrices): more room there; the Rayleigh-Ritz matrices ("gram") take 6-8
    limit = 80 if kind == "clustered" else 30
    theta, Q, sweeps = _lib.sym_eig(G, 1e-13, limit)
    assert sweeps < limit
    ref = np.linalg.eigvalsh(G)
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(np.sort(theta) - ref).max() < 1e-11 * scale * max(1, b) ** 0.5
    assert np.abs(Q.T @ Q - np.eye(b)).max() < 1e-12 * max(1, b) ** 0.5
    assert np.abs(G @ Q - Q * theta).max() < 1e-11 * scale * max(1, b) ** 0.5


@pytest.mark.gpu
def test_resident_factor_is_the_host_factor_and_rounds_in_place():
    """mmw_factor(out = NULL) leaves X_half on the device, mmw_round(gX = NULL) reads it there: the same factor bit for bit, the
    same assignment as with the array copied out and in (binary_search_relaxation.py:50-53 only hands it over); a factor somebody
    still holds is copied out before the handle overwrites it."""
    from sig_sdp_mmw_amd import _lib
    from sig_sdp_mmw_amd.graphs import journal_graph
    state = journal_graph(12, 0.02, seed=3)
    Z, nit = 12, 30
    s = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    s.iterate(nit, None, seed=2)
    rank = 2 * (Z - 1)
    host = s.factor(rank, seed=11)
    dev = s.factor(rank, seed=11, resident=True)
    assert isinstance(dev, _lib.DeviceFactor) and dev.shape == host.shape and dev._host is None
    rng = np.random.default_rng(0)
    randv = rng.standard_normal((4, Z, rank))
    randv /= np.linalg.norm(randv, axis=2, keepdims=True)
    z_dev, rem_dev = s.round(Z, dev, randv)
    assert dev._host is None  # rounded where it lies
    z_host, rem_host = s.round(Z, host, randv)
    assert np.array_equal(z_dev, z_host) and np.array_equal(rem_dev, rem_host)
    assert np.array_equal(np.asarray(dev), host)
    assert np.allclose((dev * 2.0)[3], 2.0 * host[3]) and dev.T.shape == (rank, s.K) and float(np.linalg.norm(dev)) == float(np.linalg.norm(host))
    # the next factor overwrites the device copy: a resident one still held is copied out first
    s.reset(nit)
    s.iterate(nit, None, seed=3)
    held = s.factor(rank, seed=11, resident=True)
    newer = s.factor(rank, seed=12, resident=True)
    assert held._host is not None and not held.on_device_of(s) and newer.on_device_of(s)
    z_a, _ = s.round(Z, held, randv)  # goes through the host copy
    z_b, _ = s.round(Z, np.asarray(held), randv)
    assert np.array_equal(z_a, z_b)
    s.close()
    assert np.asarray(newer).shape == (s.K, rank)  # copied out when its handle went away
