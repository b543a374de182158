"""The BASELINE.json configurations as GPU parity cases.

configs[0] (N=200 dense, 50 iterations) and configs[1] (N=2000, 5 % ER, fp64) are small enough for the CPU oracle:
trajectory parity.  configs[2] / [4] (N=10 k / 50 k) run at full size and are checked through size-independent
properties of the path: trace(X) = K, unit-simplex weights, symmetry of X on the pattern, agreement of the
locality-blocked and the generic kernels, exp(0) = I, and a rounding that is feasible by construction.
configs[3] (a batch of independent N=2000 instances) is covered as several resident handles on one device.
"""
import numpy as np
import pytest
import scipy.sparse

from conftest import relerr
from oracle import mmw_oracle as orc
from sig_sdp_mmw_amd import _lib
from sig_sdp_mmw_amd.graphs import er_contention_graph, journal_graph

pytestmark = pytest.mark.gpu


def sketches(K, D, n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([orc.sketch_rows(rng.standard_normal((K, D))) for _ in range(n)])


def test_config0_dense200_50_iterations_fp64():
    state = er_contention_graph(200, 1.0, seed=0)
    Z, nit, eta = 32, 50, 0.04
    sk = sketches(200, 2 * Z, nit, 1)
    o = orc.MMWOracle(nit=nit, eta=eta)
    o.run(Z, state, lambda i, K, D: sk[i], keep_trace=True, factor=False)
    s = _lib.Solver(Z, state, nit, eta, dtype=_lib.F64)
    s.set_expm(_lib.EXPM_LANCZOS, 16, 1e-13)
    s.iterate(nit, sk)
    assert relerr(s.read(_lib.F_XHALF), o.trace["X_half"][-1]) < 1e-8
    assert relerr(s.read(_lib.F_Y), o.trace["Y"][-1]) < 1e-8
    assert relerr(s.read(_lib.F_XAVG) / nit, o.xavg) < 1e-9
    s.close()


@pytest.mark.parametrize("dtype,bar", [(_lib.F64, 1e-8), (_lib.F32, 1e-5)])
def test_config1_er2000_5pct(dtype, bar):
    state = er_contention_graph(2000, 0.05, seed=0)
    Z, nit, eta = 32, 4, 0.04
    sk = sketches(2000, 2 * Z, nit, 2)
    o = orc.MMWOracle(nit=nit, eta=eta)
    o.run(Z, state, lambda i, K, D: sk[i], keep_trace=True, factor=False)
    s = _lib.Solver(Z, state, nit, eta, dtype=dtype)
    s.iterate(nit, sk)
    assert relerr(s.read(_lib.F_XHALF), o.trace["X_half"][-1]) < bar
    assert relerr(s.read(_lib.F_LVAL), o.trace["lval"][-1]) < max(bar, 1e-6) * 10
    s.close()


def check_invariants(s, nit_done, dtype):
    tol = 1e-4 if dtype == _lib.F32 else 1e-10
    K = s.K
    ip, ix = s.read_i32(_lib.I_L_INDPTR), s.read_i32(_lib.I_L_INDICES)
    xv = s.read(_lib.F_XVAL)
    X = scipy.sparse.csr_matrix((xv, ix, ip), shape=(K, K))
    assert abs(X.diagonal().sum() - K) < tol * K           # trace normalisation, mmw.py:184-185
    assert abs(X - X.T).max() < tol                         # symmetric fill of both triangles, mmw.py:192
    d = X.diagonal()
    coo = X.tocoo()
    assert np.all(np.abs(coo.data) <= np.sqrt(d[coo.row] * d[coo.col]) * (1 + 10 * tol) + tol)  # Gram matrix entries
    Y = s.read(_lib.F_Y)
    assert abs(Y.sum() - 1.0) < tol and Y.min() >= 0         # softmax weights
    L = scipy.sparse.csr_matrix((s.read(_lib.F_LVAL), ix, ip), shape=(K, K))
    assert abs(L - L.T).max() < tol * max(1e-3, abs(L).max())
    # running sums after a FULL run of nit iterations hold nit terms each: the last Y / X are not averaged (mmw.py:77-78,203)
    ya = s.read(_lib.F_YAVG)
    assert abs(ya.sum() - nit_done) < 1e-3 * nit_done
    xa = scipy.sparse.csr_matrix((s.read(_lib.F_XAVG), ix, ip), shape=(K, K))
    assert abs(xa.diagonal().sum() - K * nit_done) < 1e-3 * K * nit_done


@pytest.mark.parametrize("workload", ["journal-1pct", "er-1pct"])
def test_config2_n10k_full_size_properties(workload, monkeypatch):
    if workload == "journal-1pct":
        state, Z = journal_graph(28, 0.0319, 0), 186
    else:
        state, Z = er_contention_graph(10000, 0.01, 0), 32
    nit = 6
    s = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    info = s.read(_lib.F_BLOCKING)
    assert info[0] == (1.0 if workload == "journal-1pct" else 0.0)
    s.iterate(nit, None, seed=7)
    check_invariants(s, nit, _lib.F32)
    xh_blk = s.read(_lib.F_XHALF)
    assert np.all(np.isfinite(xh_blk))
    if workload == "journal-1pct":  # the same iterations through the generic kernels
        monkeypatch.setenv("MMW_BLOCKING", "0")
        t = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
        t.iterate(nit, None, seed=7)
        assert relerr(xh_blk, t.read(_lib.F_XHALF)) < 2e-5
        assert relerr(s.read(_lib.F_XVAL), t.read(_lib.F_XVAL)) < 2e-5
        t.close()
    # exp of the zero matrix is the identity: the very first X_half is the sketch itself
    u = _lib.Solver(Z, state, 1, 0.0, dtype=_lib.F32)
    u.iterate(1, None, seed=3)
    assert relerr(u.read(_lib.F_XHALF), u.read(_lib.F_SKETCH)) < 1e-6
    u.close()
    s.close()


@pytest.mark.timeout(1200)
def test_config4_n50k_rounding_batch_256_vectors():
    """N=50 000, 0.2 % ER: a batch of 8 projections of Z=32 vectors (256 rows) in one launch: the first and the last attempt of the
    batch are EXACTLY the oracle's attempts on the same vectors (sdp_solver.py:27-107), and feasible by construction."""
    state = er_contention_graph(50000, 0.002, seed=1, hi=1.5)
    K, Z, Dp = 50000, 32, 62
    s = _lib.Solver(Z, state, 3, 0.04, dtype=_lib.F32)
    s.iterate(3, None, seed=1)
    check_invariants(s, 3, _lib.F32)
    rng = np.random.default_rng(0)
    gX = rng.standard_normal((K, Dp))
    rv = rng.standard_normal((8, Z, Dp))
    rv /= np.linalg.norm(rv, axis=2, keepdims=True)
    z, rem = s.round(Z, gX, rv)
    S, Q, h = state
    So = S.copy().tolil()
    So.setdiag(0)
    So = So.tocsr()
    for a in (0, 7):
        assert int((z[a] < 0).sum()) == int(rem[a])
        z_ref, _, rem_ref, un = orc.rounding_one_attempt(Z, gX, state, rv[a], randint=lambda Z_, size: np.zeros(size))
        assert int(rem[a]) == rem_ref
        assert np.array_equal(z[a] < 0, un) and np.array_equal(z[a][~un], z_ref[~un].astype(np.int32))
        # feasibility per slot
        for zz in range(0, Z, 7):
            mem = np.where(z[a] == zz)[0]
            sub = So[mem][:, mem]
            assert np.all(np.asarray(sub.sum(axis=0)).ravel() <= h[mem] + 1e-12)
            assert Q[mem][:, mem].nnz == 0
    s.close()


def test_config3_batch_of_resident_instances():
    """Several independent N=2000 handles alive on one device, interleaved calls, each equal to its solo run."""
    n_inst, Z, nit = 4, 32, 3
    states = [er_contention_graph(2000, 0.05, seed=10 + i) for i in range(n_inst)]
    hs = [_lib.Solver(Z, st, nit, 0.04, dtype=_lib.F64) for st in states]
    for it in range(nit):
        for h in hs:
            h.iterate(1, None, seed=5)
    outs = [h.read(_lib.F_XVAL) for h in hs]
    for i, st in enumerate(states):
        solo = _lib.Solver(Z, st, nit, 0.04, dtype=_lib.F64)
        for it in range(nit):
            solo.iterate(1, None, seed=5)
        assert np.array_equal(solo.read(_lib.F_XVAL), outs[i])
        solo.close()
    for h in hs:
        h.close()


def test_edge_cases_small_and_degenerate_graphs():
    # Z = 2 (smallest legal slot count), a user with no interferers at all, an empty association relation
    S, Q, h = er_contention_graph(30, 0.2, seed=3)
    S = S.tolil()
    S[5, :] = 0
    S[:, 5] = 0
    S[5, 5] = 3.7
    S = S.tocsr()
    S.eliminate_zeros()
    Q = Q.tolil()
    Q[5, :] = 0
    Q[:, 5] = 0
    Q = Q.tocsr()
    Q.eliminate_zeros()
    for Qc in (Q, scipy.sparse.csr_matrix((30, 30))):
        state = (S, Qc, h)
        for Z in (2, 3):
            sk = sketches(30, 2 * Z, 3, 4)
            o = orc.MMWOracle(nit=3, eta=0.1)
            o.run(Z, state, lambda i, K, D: sk[i], keep_trace=True, factor=False)
            s = _lib.Solver(Z, state, 3, 0.1)
            s.set_expm(_lib.EXPM_LANCZOS, 16, 1e-13)
            s.iterate(3, sk)
            assert relerr(s.read(_lib.F_XHALF), o.trace["X_half"][-1]) < 1e-9
            assert relerr(s.read(_lib.F_Y), o.trace["Y"][-1]) < 1e-9
            X = s.factor(min(29, 2 * (Z - 1)))
            ref = orc.factor_xavg(o.pattern.csr(o.xavg), min(29, 2 * (Z - 1)))
            assert relerr(orc.projector(X), orc.projector(ref)) < 1e-6
            rv = np.random.default_rng(1).standard_normal((2, Z, X.shape[1]))
            rv /= np.linalg.norm(rv, axis=2, keepdims=True)
            z, rem = s.round(Z, X, rv)
            for a in range(2):
                zo, _, remo, _ = orc.rounding_one_attempt(Z, X, state, rv[a], randint=lambda Zs, size: np.full(size, -1))
                assert int(rem[a]) == remo and np.array_equal(z[a], zo.astype(np.int32))
            s.close()


@pytest.mark.parametrize("eta,nit", [(0.04, 120), (0.4, 60)])
def test_full_size_exponential_meets_the_tolerance_when_stopped_early(eta, nit):
    """journal N=10 003, D=372, fp32: exp(L/2) R of the last iteration against scipy's expm_multiply on the same L and the same
    sketch (first 24 columns: the columns are independent).  The Lanczos steps stop on the a-posteriori estimate, typically one
    step before the a-priori 1-norm bound would; the result must still meet the set tolerance (1e-6) -- the north star's bar is
    1e-5 -- and the estimate must be an upper bound of the error actually made.  eta = 0.4 makes L ten times larger."""
    from scipy.sparse.linalg import expm_multiply
    state, Z = journal_graph(28, 0.0319, 0), 186
    s = _lib.Solver(Z, state, nit + 1, eta, dtype=_lib.F32)
    s.set_expm(_lib.EXPM_LANCZOS, 12, 1e-6)
    s.iterate(nit, None, seed=11)
    ip, ix = s.read_i32(_lib.I_L_INDPTR), s.read_i32(_lib.I_L_INDICES)
    K = s.K
    s.iterate(1, None, seed=11)             # one more iteration: its L, sketch and X_half are read back
    L = scipy.sparse.csr_matrix((s.read(_lib.F_LVAL), ix, ip), shape=(K, K))
    R = s.read(_lib.F_SKETCH)[:, :24]
    got = s.read(_lib.F_XHALF)[:, :24]
    ref = expm_multiply(0.5 * L, R)
    err = relerr(got, ref)
    info = s.read(_lib.F_EXPM_INFO)
    assert err < 2e-6, (err, info)           # tolerance 1e-6 + fp32 rounding of a K x D block (~4e-8 per entry)
    assert info[1] >= 1
    s.close()


@pytest.mark.parametrize("eta,nit,expect_first", [(0.04, 120, True), (0.4, 60, False)])
def test_full_size_first_order_exponential_meets_the_tolerance(eta, nit, expect_first):
    """journal N=10 003, D=372, fp32: the last iteration of a chunked run takes exp(L/2)R as one first-order product on fp16 operands (it
    leaves the fp32 copy with the factor e^mu the product itself drops); against scipy's expm_multiply on the same L and sketch, first
    24 columns.  With eta = 0.4 the norm outgrows what the single fp16 plane of u may carry (ExpmPlan::f16_ok) and what one step may
    (the bound): the chunks go back to Lanczos steps by themselves, and whatever form the last iteration took meets the tolerance."""
    from scipy.sparse.linalg import expm_multiply
    state, Z = journal_graph(28, 0.0319, 0), 186
    s = _lib.Solver(Z, state, nit + 1, eta, dtype=_lib.F32)
    s.set_expm(_lib.EXPM_LANCZOS, 12, 1e-6)
    s.iterate(nit, None, seed=11)
    s.sync()
    first = s.read(_lib.F_DUAL_INFO)[2]
    if expect_first:
        assert first >= nit // 2 and s.read(_lib.F_BLOCKING)[3] == 0, (first, s.read(_lib.F_BLOCKING))
    else:
        assert first < nit // 2, first
    ip, ix = s.read_i32(_lib.I_L_INDPTR), s.read_i32(_lib.I_L_INDICES)
    K = s.K
    L = scipy.sparse.csr_matrix((s.read(_lib.F_LVAL), ix, ip), shape=(K, K))
    R = s.read(_lib.F_SKETCH)
    got = s.read(_lib.F_XHALF)
    ref = expm_multiply(0.5 * L, R[:, :24])
    err = relerr(got[:, :24], ref)
    assert err < 2e-6, err
    if expect_first:
        # what the fused launches of that last iteration left around the exponential, against the oracle's functions on the same state:
        # X on the pattern (the SDDMM on the planes the product's epilogue wrote, the diagonal from its fixed-point row norms, the
        # trace from its shares) and the weights (softmax inside the violation pass, normalised by the LOSS pass)
        from oracle import mmw_oracle as orc
        pat = orc.Pattern(Z, state)
        xref = orc.x_on_pattern(pat, expm_multiply(0.5 * L, R))
        assert relerr(s.read(_lib.F_XVAL), xref) < 1e-4
        assert relerr(s.read(_lib.F_Y), orc.softmax(s.read(_lib.F_E_ACCU))) < 1e-4
    s.close()
