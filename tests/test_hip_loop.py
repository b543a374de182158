"""GPU parity of the device-resident MMW loop (through the C-ABI) against the golden vectors of the
reference and against the CPU oracle.

Tolerances: the north star asks <= 1e-5 relative Frobenius on exp(L/2)R; with the Krylov tolerance set
tight the fp64 path is held to 1e-9 on every per-iteration quantity, the fp32 path to 1e-4 / 1e-5.
"""
import numpy as np
import pytest

from conftest import csr_from, load_golden, relerr, state_from
from oracle import mmw_oracle as orc
from sig_sdp_mmw_amd import _lib

pytestmark = pytest.mark.gpu


def pattern_csr(s, vals):
    import scipy.sparse
    return scipy.sparse.csr_matrix((vals, s.read_i32(_lib.I_L_INDICES), s.read_i32(_lib.I_L_INDPTR)), shape=(s.K, s.K))


@pytest.mark.parametrize("method", [_lib.EXPM_LANCZOS, _lib.EXPM_TAYLOR])
def test_trajectory_matches_reference_fp64(run_case, method):
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    s = _lib.Solver(Z, state, nit, eta, dtype=_lib.F64)
    s.set_expm(method, 16, 1e-13)
    diag = s.read_i32(_lib.I_DIAG_POS)
    for i in range(nit):
        s.iterate(1, g["randv"][i])
        assert relerr(s.read(_lib.F_E_THIS), g["e_this"][i]) < 1e-9, (name, i)
        assert relerr(s.read(_lib.F_E_ACCU), g["e_accu"][i]) < 1e-9
        assert relerr(s.read(_lib.F_Y), g["Y"][i]) < 1e-9
        L = pattern_csr(s, s.read(_lib.F_LVAL))
        Lref = csr_from(g, "Laccu%d" % i)
        assert abs(L - Lref).max() < 1e-10 * max(1e-3, abs(Lref).max())
        assert relerr(s.read(_lib.F_XHALF), g["X_half_it"][i]) < 1e-9
        xv = s.read(_lib.F_XVAL)
        assert relerr(xv[diag], g["X_mdiag"][i]) < 1e-9
        xo = xv.copy()
        xo[diag] = 0
        assert abs(pattern_csr(s, xo) - csr_from(g, "Xoffdi%d" % i)).max() < 1e-9
    xavg = pattern_csr(s, s.read(_lib.F_XAVG) / nit)
    assert abs(xavg - csr_from(g, "Xavgd")).max() < 1e-9
    info = s.read(_lib.F_EXPM_INFO)
    assert 1 <= info[1] <= 16 and info[2] == 1
    s.close()


def test_trajectory_fp32_within_tolerance(run_case):
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    s = _lib.Solver(Z, state, nit, eta, dtype=_lib.F32)
    s.set_expm(_lib.EXPM_LANCZOS, 12, 1e-7)
    for i in range(nit):
        s.iterate(1, g["randv"][i])
        assert relerr(s.read(_lib.F_XHALF), g["X_half_it"][i]) < 1e-5, (name, i)  # north-star bar
        assert relerr(s.read(_lib.F_Y), g["Y"][i]) < 1e-4
    s.close()


def test_batched_iterate_equals_stepwise(run_case):
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    a = _lib.Solver(Z, state, nit, eta)
    b = _lib.Solver(Z, state, nit, eta)
    a.iterate(nit, g["randv"][:nit])
    for i in range(nit):
        b.iterate(1, g["randv"][i])
    for f in (_lib.F_Y, _lib.F_LVAL, _lib.F_XVAL, _lib.F_XAVG, _lib.F_YAVG):
        assert np.array_equal(a.read(f), b.read(f))  # bitwise reproducible: no atomics anywhere
    with pytest.raises(_lib.MMWError):
        a.iterate(1, g["randv"][0])  # more than the announced nit
    a.reset(nit)
    a.iterate(nit, g["randv"][:nit])
    assert np.array_equal(a.read(_lib.F_XVAL), b.read(_lib.F_XVAL))
    a.close()
    b.close()


def test_default_tolerance_meets_north_star_bar(run_case):
    """Default Krylov tolerance (order chosen on the device from the one-norm bound)."""
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    for dtype, bar in ((_lib.F64, 1e-7), (_lib.F32, 1e-5)):
        s = _lib.Solver(Z, state, nit, eta, dtype=dtype)
        for i in range(nit):
            s.iterate(1, g["randv"][i])
            assert relerr(s.read(_lib.F_XHALF), g["X_half_it"][i]) < bar
        s.close()


@pytest.mark.parametrize("dtype,tol,bar", [(_lib.F64, 1e-12, 1e-10), (_lib.F64, 1e-6, 1e-5), (_lib.F32, 1e-6, 1e-5)])
@pytest.mark.parametrize("method", [_lib.EXPM_LANCZOS, _lib.EXPM_TAYLOR])
def test_expm_seam_all_norms(dtype, tol, bar, method):
    """mmw.expm_half_randsk seam at one-norms 0.004 ... 12.6 (substeps kick in at the top)."""
    g = load_golden("expm_seam")
    L = csr_from(g, "L")
    for i, sc in enumerate(g["scales"]):
        Ls = L.copy()
        Ls.data = Ls.data * sc
        out, info = _lib.expm_apply(Ls, g["randv%d" % i], dtype=dtype, method=method, max_order=16, tol=tol)
        assert relerr(out, g["X%d" % i]) < bar, (i, info)
        assert info["one_norm"] >= float(g["onenorm%d" % i]) * (1 - 1e-6)
        assert info["one_norm"] <= float(g["onenorm%d" % i]) * 1.0001 + 1e-12
    assert info["substeps"] > 1


def test_expm_zero_matrix_and_odd_widths():
    import scipy.sparse
    rng = np.random.default_rng(0)
    K = 97
    for D in (1, 3, 5, 31, 40, 70, 140, 300):
        B = rng.standard_normal((K, D))
        Z0 = scipy.sparse.csr_matrix((K, K))
        out, info = _lib.expm_apply(Z0, B)
        assert relerr(out, B) < 1e-14
        A = scipy.sparse.random(K, K, 0.1, random_state=1)
        A = (A + A.T) * 0.3
        ref = orc.expm_half(A.tocsr(), B)
        for dt, bar in ((_lib.F64, 1e-9), (_lib.F32, 2e-6)):
            out, info = _lib.expm_apply(A.tocsr(), B, dtype=dt, tol=1e-11 if dt == _lib.F64 else 1e-7, max_order=16)
            assert relerr(out, ref) < bar, (D, dt, info)


def test_device_sketch_rows_are_unit_gaussian_directions():
    from sig_sdp_mmw_amd.graphs import er_contention_graph
    state = er_contention_graph(500, 0.03, 2)
    s = _lib.Solver(20, state, 2, 0.05)
    s.iterate(1, None, seed=123)
    R = s.read(_lib.F_SKETCH)
    assert R.shape == (500, 40)
    np.testing.assert_allclose(np.linalg.norm(R, axis=1), 1.0, rtol=1e-12)
    # isotropy: mean ~ 0, second moment ~ 1/D, no duplicated rows, different per iteration/seed
    assert abs(R.mean()) < 5e-3
    assert abs((R ** 2).mean() - 1 / 40) < 1e-3
    assert len({r.tobytes() for r in R}) == 500
    s.iterate(1, None, seed=123)
    assert not np.array_equal(R, s.read(_lib.F_SKETCH))
    t = _lib.Solver(20, state, 2, 0.05)
    t.iterate(1, None, seed=123)
    assert np.array_equal(R, t.read(_lib.F_SKETCH))  # counter-based: same seed, same iteration -> same draw
    # and the iterate it produced is the oracle's on that sketch
    o = orc.MMWOracle(nit=1, eta=0.05)
    o.run(20, state, lambda i, K, D: R, keep_trace=True, factor=False)
    assert relerr(t.read(_lib.F_XHALF), o.trace["X_half"][0]) < 1e-7
    s.close()
    t.close()


def test_phase_timers_are_filled():
    from sig_sdp_mmw_amd.graphs import er_contention_graph
    s = _lib.Solver(8, er_contention_graph(300, 0.05, 3), 5, 0.05, dtype=_lib.F32)
    s.set_timing(True)
    s.iterate(5, None, seed=1)
    t = s.read(_lib.F_PHASE_US).reshape(5, 4)
    assert np.all(t > 0) and np.all(t[:, 3] >= t[:, :3].max(axis=1))
    s.close()


def test_blocked_and_generic_spmm_agree(run_case, monkeypatch):
    """The LDS-staged locality-blocked SpMM and the generic gather SpMM are two traversals of the same sums."""
    name, g = run_case
    state = state_from(g)
    Z, nit, eta = int(g["Z"]), int(g["nit"]), float(g["eta"])
    outs = {}
    for mode in ("1", "full", "0"):  # half-tile kernel (two workgroups per CU), full-tile kernel, generic gather
        monkeypatch.setenv("MMW_BLOCKING", "0" if mode == "0" else "1")
        if mode == "full":
            monkeypatch.setenv("MMW_FULL_TILE", "1")
        else:
            monkeypatch.delenv("MMW_FULL_TILE", raising=False)
        for dtype in (_lib.F64, _lib.F32):
            s = _lib.Solver(Z, state, nit, eta, dtype=dtype)
            info = s.read(_lib.F_BLOCKING)
            assert info[0] == (0.0 if mode == "0" else 1.0), (name, info)
            s.set_expm(_lib.EXPM_LANCZOS, 16, 1e-13 if dtype == _lib.F64 else 1e-7)
            s.iterate(nit, g["randv"][:nit])
            outs[(mode, dtype)] = (s.read(_lib.F_XHALF), s.read(_lib.F_LVAL))
            s.close()
    for dtype, bar in ((_lib.F64, 1e-12), (_lib.F32, 2e-5)):
        for mode in ("1", "full"):
            assert relerr(outs[(mode, dtype)][0], outs[("0", dtype)][0]) < bar
            assert relerr(outs[(mode, dtype)][1], outs[("0", dtype)][1]) < bar


@pytest.mark.parametrize("kind", ["journal", "er"])
def test_blocking_on_mid_size_graphs(kind):
    from sig_sdp_mmw_amd.graphs import er_contention_graph, journal_graph
    state = journal_graph(14, 0.012, seed=2) if kind == "journal" else er_contention_graph(1500, 0.02, 3)
    K = state[0].shape[0]
    Z, nit = 30, 3
    rng = np.random.default_rng(1)
    sk = np.stack([orc.sketch_rows(rng.standard_normal((K, 2 * Z))) for _ in range(nit)])
    o = orc.MMWOracle(nit=nit, eta=0.05)
    o.run(Z, state, lambda i, K_, D_: sk[i], keep_trace=True, factor=False)
    for dtype, bar in ((_lib.F64, 1e-9), (_lib.F32, 1e-5)):
        s = _lib.Solver(Z, state, nit, 0.05, dtype=dtype)
        s.set_expm(_lib.EXPM_LANCZOS, 16, 1e-12 if dtype == _lib.F64 else 1e-7)
        info = s.read(_lib.F_BLOCKING)
        if kind == "journal":
            assert info[0] == 1.0 and info[2] > 3.0  # geometric graph: real reuse
        s.iterate(nit, sk)
        assert relerr(s.read(_lib.F_XHALF), o.trace["X_half"][-1]) < bar, (kind, dtype, info)
        s.close()


@pytest.mark.parametrize("Z", [2, 3, 17])
def test_blocked_kernels_with_narrow_and_ragged_blocks(Z):
    """D = 2Z = 4, 6, 34 columns: narrower than one 128-byte half tile / not a multiple of it.  The half-tile SpMM and SDDMM
    must mask the columns past the block's width (gathers, deposits, stores) and still match the oracle."""
    from sig_sdp_mmw_amd.graphs import journal_graph
    state = journal_graph(12, 0.012, seed=5)
    K = state[0].shape[0]
    nit = 3
    rng = np.random.default_rng(2)
    sk = np.stack([orc.sketch_rows(rng.standard_normal((K, 2 * Z))) for _ in range(nit)])
    o = orc.MMWOracle(nit=nit, eta=0.05)
    o.run(Z, state, lambda i, K_, D_: sk[i], keep_trace=True, factor=False)
    for dtype, bar in ((_lib.F64, 1e-9), (_lib.F32, 1e-5)):
        s = _lib.Solver(Z, state, nit, 0.05, dtype=dtype)
        s.set_expm(_lib.EXPM_LANCZOS, 16, 1e-12 if dtype == _lib.F64 else 1e-7)
        assert s.read(_lib.F_BLOCKING)[0] == 1.0
        s.iterate(nit, sk)
        assert relerr(s.read(_lib.F_XHALF), o.trace["X_half"][-1]) < bar, (Z, dtype)
        s.close()


# ---- matrix-core SpMM (kernels_mfma.h): the fp32 default on locality-blocked patterns ------------------------------
def _journal_small():
    from sig_sdp_mmw_amd.graphs import journal_graph
    return journal_graph(16, 0.02, seed=4)  # K = 2048, dense enough for row blocks with reuse


@pytest.mark.parametrize("Z", [9, 24, 70])
def test_matrix_core_spmm_matches_the_fp32_kernel_and_the_oracle(Z, monkeypatch):
    """Dpad = 32 / 64 / 160: one, two and five column tiles (partial groups, idle waves)."""
    state = _journal_small()
    K, nit, eta = state[0].shape[0], 4, 0.04
    rng = np.random.default_rng(1)
    sk = np.stack([orc.sketch_rows(rng.standard_normal((K, 2 * Z))) for _ in range(nit)])
    o = orc.MMWOracle(nit=nit, eta=eta)
    o.run(Z, state, lambda i, K_, D_: sk[i], keep_trace=True, factor=False)
    a = _lib.Solver(Z, state, nit, eta, dtype=_lib.F32)
    kind = a.read(_lib.F_SPMM_KIND)
    assert kind[0] == 3.0, "this pattern is expected to run on the matrix-core kernel"
    a.iterate(nit, sk)
    xa = a.read(_lib.F_XHALF)
    assert relerr(xa, o.trace["X_half"][-1]) < 1e-5      # the north-star bar on exp(L/2)R
    assert relerr(a.read(_lib.F_XVAL), o.trace["xval"][-1]) < 1e-4
    monkeypatch.setenv("MMW_NO_MFMA", "1")
    b = _lib.Solver(Z, state, nit, eta, dtype=_lib.F32)
    assert b.read(_lib.F_SPMM_KIND)[0] in (1.0, 2.0)
    b.iterate(nit, sk)
    assert relerr(xa, b.read(_lib.F_XHALF)) < 2e-6        # same result as the fp32 LDS kernel to fp32 rounding
    assert relerr(a.read(_lib.F_XVAL), b.read(_lib.F_XVAL)) < 3e-5   # X on the pattern: matrix-core SDDMM (two-half split) vs fp32 SDDMM
    assert np.array_equal(np.isfinite(a.read(_lib.F_XVAL)), np.ones(a.nnzL, dtype=bool))
    a.close(); b.close()


def test_matrix_core_spmm_steps_aside_when_the_norm_outgrows_the_split(monkeypatch):
    """A large step size makes max_i sum_j |a_ij| exceed what two bf16 halves resolve within the tolerance: the plan says so,
    the optimistic chunk is replayed, and the run continues on the fp32 kernel -- same result as never using the matrix cores."""
    state = _journal_small()
    Z, nit, eta = 12, 24, 2.0
    a = _lib.Solver(Z, state, nit, eta, dtype=_lib.F32)
    a.iterate(nit, None, seed=5)
    xa = a.read(_lib.F_XHALF)
    assert a.read(_lib.F_SPMM_KIND)[1] == 0.0, "the last plan is expected to have ruled the matrix-core kernel out"
    monkeypatch.setenv("MMW_NO_MFMA", "1")
    b = _lib.Solver(Z, state, nit, eta, dtype=_lib.F32)
    b.iterate(nit, None, seed=5)
    assert relerr(xa, b.read(_lib.F_XHALF)) < 1e-5
    assert relerr(a.read(_lib.F_LVAL), b.read(_lib.F_LVAL)) < 1e-5
    a.close(); b.close()


@pytest.mark.parametrize("rows,gt", [("32", "4"), ("32", "12"), ("64", "8"), ("20", "4")])
def test_matrix_core_kernels_in_their_other_shapes(rows, gt, monkeypatch):
    """Row blocks of at most 32 rows (one row tile: the 4-wave SpMM / SDDMM instantiations), wider column groups, blocks that
    leave the second row tile half empty: same result as the fp32 LDS kernels."""
    state = _journal_small()
    Z, nit = 70, 3
    monkeypatch.setenv("MMW_NO_MFMA", "1")
    b = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    b.iterate(nit, None, seed=2)
    xh, xv = b.read(_lib.F_XHALF), b.read(_lib.F_XVAL)
    b.close()
    monkeypatch.delenv("MMW_NO_MFMA")
    monkeypatch.setenv("MMW_MF_ROWS", rows)
    monkeypatch.setenv("MMW_MF_GT", gt)
    a = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    assert a.read(_lib.F_SPMM_KIND)[0] == 3.0
    a.iterate(nit, None, seed=2)
    assert relerr(a.read(_lib.F_XHALF), xh) < 2e-6
    assert relerr(a.read(_lib.F_XVAL), xv) < 3e-5
    a.close()
