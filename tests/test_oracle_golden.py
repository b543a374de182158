"""Pin the CPU oracle (oracle/mmw_oracle.py) to the golden vectors captured from the reference.

CPU-only.  Tolerances: the oracle is a re-ordered O(nnz) restatement, so agreement is to
rounding (1e-12 relative is asserted; observed ~1e-15), integers exactly.
"""
import numpy as np
import pytest
import scipy.sparse

from conftest import csr_from, load_golden, relerr, state_from
from oracle import mmw_oracle as orc


def test_state_processing_matches_reference(run_case):
    name, g = run_case
    state = state_from(g)
    Z = int(g["Z"])
    p = orc.Pattern(Z, state)
    assert str(g["ST_format"]) == "csc"  # the quirk the loss phase depends on
    ST = csr_from(g, "ST")
    assert (p.ST != ST).nnz == 0
    np.testing.assert_allclose(p.S_sum, g["S_sum"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(p.norm_H, g["norm_H"], rtol=1e-14, atol=0)
    for mine, ref in [(p.gain_x, "nz_idx_gain_x_ut"), (p.gain_y, "nz_idx_gain_y_ut"),
                      (p.asso_x, "nz_idx_asso_x_ut"), (p.asso_y, "nz_idx_asso_y_ut")]:
        assert np.array_equal(mine, g[ref])
    assert p.C == int(g["C"]) and p.E_asso == int(g["E_asso"])


def test_per_phase_arithmetic_on_injected_inputs(run_case):
    """Each phase fed with the reference's own inputs of that phase (no error compounding)."""
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    p = orc.Pattern(Z, state)
    K = p.K
    e_accu = np.zeros(p.C)
    xval = np.zeros(p.nnzL)
    xval[p.diag_pos] = 1.0
    for i in range(nit):
        for as_exec in (False, True):
            e = orc.violations(p, xval, dual_as_executed=as_exec)
            assert relerr(e, g["e_this"][i]) < 1e-12
        e_accu = g["e_accu"][i]
        Y = orc.softmax(e_accu)
        assert relerr(Y, g["Y"][i]) < 1e-13
        Lref = csr_from(g, "Laccu%d" % i)
        Lprev = csr_from(g, "Laccu%d" % (i - 1)) if i else scipy.sparse.csr_matrix((K, K))
        step = p.csr(orc.loss_values(p, g["Y"][i]) * eta)
        assert abs((Lprev - step) - Lref).max() < 1e-13 * max(1.0, abs(Lref).max())
        # pattern of L_accu is the oracle's fixed pattern
        Lr = Lref.copy()
        Lr.sort_indices()
        assert Lr.nnz <= p.nnzL
        # X on the pattern from the reference's X_half
        xval = orc.x_on_pattern(p, g["X_half_it"][i])
        assert relerr(xval[p.diag_pos], g["X_mdiag"][i]) < 1e-13
        xo = xval.copy()
        xo[p.diag_pos] = 0
        assert abs(p.csr(xo) - csr_from(g, "Xoffdi%d" % i)).max() < 1e-13
        # the seam: same SciPy call on the same inputs is bit-identical
        Xh = orc.expm_half(csr_from(g, "Lhalf%d" % i), g["randv"][i])
        assert np.array_equal(Xh, g["X_half_it"][i])


def test_full_trajectory_with_recorded_sketches(run_case):
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    o = orc.MMWOracle(nit=nit, eta=eta, log_gap=bool(int(g["log_gap"])))
    ok, _ = o.run(Z, state, lambda i, K, D: g["randv"][i], keep_trace=True, factor=False)
    p = o.pattern
    for i in range(nit):
        assert relerr(o.trace["Y"][i], g["Y"][i]) < 1e-11
        assert relerr(o.trace["X_half"][i], g["X_half_it"][i]) < 1e-11
        assert abs(p.csr(o.trace["lval"][i]) - csr_from(g, "Laccu%d" % i)).max() < 1e-12
    assert abs(p.csr(o.xavg) - csr_from(g, "Xavgd")).max() < 1e-12
    if int(g["log_gap"]):
        gap = np.array(o.trace["gap"])
        np.testing.assert_allclose(gap[:, 0], g["gap"][:, 0], rtol=1e-10)
        np.testing.assert_allclose(gap[:, 1], g["gap"][:, 1], rtol=1e-7, atol=1e-9)


def test_sketch_rows_match_reference_draw(run_case):
    name, g = run_case
    K = int(g["S_shape"][0])
    D = int(g["Z"]) * 2
    np.random.seed(int(g["seed"]))
    r = orc.sketch_rows(np.random.randn(K, D))
    assert np.array_equal(r, g["randv"][0])


def test_factor_projector_matches_reference(run_case):
    """svds has sign/rotation freedom: compare X_half X_half^T (SURVEY §8c fixture 4)."""
    name, g = run_case
    Xavg = csr_from(g, "Xavgd")
    K = Xavg.shape[0]
    rank = int(min(K - 1, (int(g["Z"]) - 1) * 2))
    assert g["X_half_ret"].shape == (K, rank)
    mine = orc.factor_xavg(Xavg, rank)
    assert relerr(orc.projector(mine), orc.projector(g["X_half_ret"])) < 1e-8


def test_rounding_attempts_exact(run_case):
    name, g = run_case
    state = state_from(g)
    for pre, Zk, gXk in (("att_", "round_Z", "round_gX"), ("small_att_", "small_Z", "small_gX")):
        Z = int(g[Zk])
        gX = g[gXk]
        for a in range(g[pre + "randv"].shape[0]):
            rv = g[pre + "randv"][a]
            rv = rv / np.linalg.norm(rv, axis=1, keepdims=True)
            left = g[pre + "randint"][a]
            left = left[left >= 0]
            fn = lambda Zs, size: left[:size]  # noqa: E731
            z, _, rem, un = orc.rounding_one_attempt(Z, gX, state, rv, randint=fn)
            assert rem == int(g[pre + "rem"][a])
            assert np.array_equal(z, g[pre + "z"][a])
            if state[0].shape[0] <= 130:
                z2, _, rem2, un2 = orc.rounding_one_attempt_as_executed(Z, gX, state, rv, randint=fn)
                assert rem2 == rem and np.array_equal(z2, z)


def test_rounding_multi_attempt_replay(run_case):
    """sdp_solver.rounding replayed on the same global NumPy stream reproduces z_vec exactly."""
    name, g = run_case
    state = state_from(g)
    np.random.seed(int(g["round_seed"]))
    z, Z, rem = orc.rounding(int(g["round_Z"]), g["round_gX"], state, orc.draw_randv_host)
    assert rem == int(g["round_rem"])
    assert np.array_equal(z, g["round_z_vec"])


def test_expm_seam_large_norms():
    g = load_golden("expm_seam")
    L = csr_from(g, "L")
    for i, s in enumerate(g["scales"]):
        Ls = L.copy()
        Ls.data = Ls.data * s
        X = orc.expm_half(Ls, g["randv%d" % i])
        assert np.array_equal(X, g["X%d" % i])
        if L.shape[0] <= 200:
            import scipy.linalg
            dense = scipy.linalg.expm(Ls.toarray()) @ g["randv%d" % i]
            assert relerr(X, dense) < 1e-12
