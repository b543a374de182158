"""Full-size parity on the GPU (the BASELINE configurations at their real sizes, not only through properties):

* journal-1pct (N = 10 003, Z = 186, D = 372; the benchmark instance, locality-blocked kernels on): three iterations on
  uploaded sketches against the CPU oracle, every field, fp64 (<= 1e-9) and fp32 (<= 1e-5 on exp(L/2)R, <= 1e-4 on the
  weights and the accumulated loss);
* the epilogue factor at that size and rank (370) against the oracle's svds: spectrum of X_half^T X_half and the projector
  X_half X_half^T on 512 sampled rows;
* configs[4] (N = 50 000): one rounding attempt, EXACT equality of the slot of every user and of the remainder with the
  oracle (mmw.py:124-197, sdp_solver.py:70-101, mmw.py:213-216).
"""
import numpy as np
import pytest
import scipy.sparse

from conftest import relerr
from oracle import mmw_oracle as orc
from sig_sdp_mmw_amd import _lib
from sig_sdp_mmw_amd.graphs import er_contention_graph, journal_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bench_instance():
    state = journal_graph(28, 0.0319, 0)
    Z, nit, eta = 186, 3, 0.04
    K = state[0].shape[0]
    rng = np.random.default_rng(5)
    sk = np.stack([orc.sketch_rows(rng.standard_normal((K, 2 * Z))) for _ in range(nit)])
    o = orc.MMWOracle(nit=nit, eta=eta)
    o.run(Z, state, lambda i, K_, D_: sk[i], keep_trace=True, factor=False)
    return state, Z, nit, eta, sk, o


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_n10k_trajectory_matches_the_oracle_every_field(bench_instance, dtype):
    state, Z, nit, eta, sk, o = bench_instance
    f64 = dtype == "f64"
    s = _lib.Solver(Z, state, nit, eta, dtype=_lib.F64 if f64 else _lib.F32)
    assert s.read(_lib.F_BLOCKING)[0] == 1.0  # the locality-blocked kernels are the ones under test
    bx, bw = (1e-9, 1e-9) if f64 else (1e-5, 1e-4)  # exp(L/2)R ; weights, accumulated loss, X on the pattern
    if f64:
        s.set_expm(_lib.EXPM_LANCZOS, 16, 1e-12)
    for i in range(nit):
        s.iterate(1, sk[i])
        assert relerr(s.read(_lib.F_E_THIS), o.trace["e_this"][i]) < bw, i
        assert relerr(s.read(_lib.F_E_ACCU), o.trace["e_accu"][i]) < bw, i
        assert relerr(s.read(_lib.F_Y), o.trace["Y"][i]) < bw, i
        assert relerr(s.read(_lib.F_LVAL), o.trace["lval"][i]) < bw, i
        assert relerr(s.read(_lib.F_XHALF), o.trace["X_half"][i]) < bx, i
        assert relerr(s.read(_lib.F_XVAL), o.trace["xval"][i]) < bw, i
    # running sums (mmw.py:77-78): nit terms each, the last X / Y excluded
    assert relerr(s.read(_lib.F_XAVG) / nit, o.xavg) < bw
    assert relerr(s.read(_lib.F_YAVG) / nit, o.yavg) < bw
    s.close()


@pytest.mark.timeout(1200)
def test_n10k_factor_rank_370_against_oracle_svds(bench_instance):
    state, Z, nit, eta, sk, o = bench_instance
    K = state[0].shape[0]
    rank = min(K - 1, 2 * (Z - 1))
    assert rank == 370
    s = _lib.Solver(Z, state, nit, eta, dtype=_lib.F64)
    s.set_expm(_lib.EXPM_LANCZOS, 16, 1e-12)
    s.iterate(nit, sk)
    X = s.factor(rank, seed=2)
    xavg = scipy.sparse.csr_matrix((s.read(_lib.F_XAVG) / nit, s.read_i32(_lib.I_L_INDICES), s.read_i32(_lib.I_L_INDPTR)), shape=(K, K))
    ref = orc.factor_xavg(xavg, rank)
    # spectrum: the eigenvalues of X_half^T X_half are the top singular values of Xbar
    ev = np.sort(np.linalg.eigvalsh(X.T @ X))
    ev_ref = np.sort(np.linalg.eigvalsh(ref.T @ ref))
    assert np.max(np.abs(ev - ev_ref)) <= 1e-7 * ev_ref.max()
    # projector on sampled rows (the full K x K product is 800 MB)
    rows = np.random.default_rng(0).choice(K, size=512, replace=False)
    assert relerr(X[rows] @ X.T, ref[rows] @ ref.T) < 1e-6
    s.close()


@pytest.mark.timeout(1200)
def test_n50k_rounding_is_exactly_the_oracles():
    state = er_contention_graph(50000, 0.002, seed=1, hi=1.5)
    K, Z, Dp = 50000, 32, 62
    rng = np.random.default_rng(3)
    gX = rng.standard_normal((K, Dp)) * np.exp(rng.standard_normal((K, 1)) * 0.3)
    rv = rng.standard_normal((2, Z, Dp))
    rv /= np.linalg.norm(rv, axis=2, keepdims=True)
    s = _lib.Solver(Z, state, 1, 0.04, dtype=_lib.F32)
    z, rem = s.round(Z, gX, rv)
    for b in (0, 1):
        z_ref, _, rem_ref, un = orc.rounding_one_attempt(Z, gX, state, rv[b], randint=lambda Z_, size: np.zeros(size))
        assert int(rem[b]) == rem_ref
        assert np.array_equal(z[b] < 0, un)
        assert np.array_equal(z[b][~un], z_ref[~un].astype(np.int32))
    # the same at a slot count where the greedy pass must leave users out
    Zs = 6
    rv2 = rng.standard_normal((1, Zs, Dp))
    rv2 /= np.linalg.norm(rv2, axis=2, keepdims=True)
    z2, rem2 = s.round(Zs, gX, rv2)
    z_ref, _, rem_ref, un = orc.rounding_one_attempt(Zs, gX, state, rv2[0], randint=lambda Z_, size: np.zeros(size))
    assert rem_ref > 0 and int(rem2[0]) == rem_ref
    assert np.array_equal(z2[0] < 0, un) and np.array_equal(z2[0][~un], z_ref[~un].astype(np.int32))
    s.close()
