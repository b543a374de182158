"""Handle life-cycle on the GPU: replay of an optimistic chunk, rebinding to another slot count, a changed step size,
the opt-in warm start of the binary search, and the environment defaults of the drop-in class."""
import numpy as np
import pytest

from conftest import load_golden, relerr, state_from
from sig_sdp_mmw_amd import _lib
from sig_sdp_mmw_amd.graphs import journal_graph
from sig_sdp_mmw_amd.mmw import mmw

pytestmark = pytest.mark.gpu

FIELDS = (_lib.F_XAVG, _lib.F_E_ACCU, _lib.F_LVAL, _lib.F_YAVG, _lib.F_XVAL, _lib.F_Y)


def test_replayed_chunk_equals_the_synchronous_run(monkeypatch):
    """Without the a-posteriori stop the a-priori Krylov order rises during the run: optimistic chunks launched with too few
    stages are restored from their snapshot and replayed.  The result must be the run that reads the plan every iteration."""
    monkeypatch.setenv("MMW_NO_APOST", "1")
    state = journal_graph(8, 75e-4, seed=1)  # K = 192
    a = _lib.Solver(12, state, 40, 0.05, dtype=_lib.F32)
    a.iterate(40, None, seed=3)
    got = [a.read(f) for f in FIELDS]
    assert a.read(_lib.F_BLOCKING)[3] >= 1, "this configuration is expected to replay at least one chunk"
    a.close()
    monkeypatch.setenv("MMW_SYNC_PLAN", "1")
    b = _lib.Solver(12, state, 40, 0.05, dtype=_lib.F32)
    b.iterate(40, None, seed=3)
    assert b.read(_lib.F_BLOCKING)[3] == 0
    for f, x in zip(FIELDS, got):
        # same launches in the same order once replayed -> the same bits; allow rounding-level slack for the order choice
        assert relerr(x, b.read(f)) < 1e-6, f
    b.close()


@pytest.mark.parametrize("dtype", [_lib.F64, _lib.F32])
def test_set_slots_equals_a_fresh_handle(dtype):
    state = journal_graph(8, 75e-4, seed=2)
    Z1, Z2, nit = 16, 9, 6
    a = _lib.Solver(Z1, state, nit, 0.05, dtype=dtype)
    a.iterate(nit, None, seed=11)  # dirty every buffer at the first width
    a.set_slots(Z2, nit)
    a.iterate(nit, None, seed=12)
    b = _lib.Solver(Z2, state, nit, 0.05, dtype=dtype)
    b.iterate(nit, None, seed=12)
    assert (a.Z, a.D) == (b.Z, b.D) == (Z2, 2 * Z2)
    assert np.array_equal(a.read(_lib.F_NORM_H), b.read(_lib.F_NORM_H))
    for f in (_lib.F_XHALF,) + FIELDS:
        assert np.array_equal(a.read(f), b.read(f)), f
    K = a.K
    assert np.array_equal(a.factor(min(K - 1, 2 * (Z2 - 1)), seed=1), b.factor(min(K - 1, 2 * (Z2 - 1)), seed=1))
    a.close(); b.close()


def test_set_eta_is_honoured_by_a_reused_handle():
    state = journal_graph(8, 75e-4, seed=2)
    a = _lib.Solver(10, state, 5, 0.05)
    a.iterate(5, None, seed=4)
    a.set_eta(0.11)
    a.reset(5)
    a.iterate(5, None, seed=4)
    b = _lib.Solver(10, state, 5, 0.11)
    b.iterate(5, None, seed=4)
    for f in FIELDS:
        assert np.array_equal(a.read(f), b.read(f)), f
    a.close(); b.close()
    # through the class: the reference reads self.eta on every run (mmw.py:137,167)
    alg = mmw(nit=5, eta=0.05, rng="device", seed=1)
    alg.run_with_state(0, 10, state)
    alg.eta = 0.11
    alg._runs = 0
    alg.run_with_state(1, 10, state)
    ref = mmw(nit=5, eta=0.11, rng="device", seed=1)
    ref.run_with_state(0, 10, state)
    assert np.array_equal(alg._dev[2].read(_lib.F_LVAL), ref._dev[2].read(_lib.F_LVAL))
    # an in-place edit of the state's values must not be served from the stale device copy
    S2 = state[0].copy()
    S2.data[:] *= 1.5
    alg._runs = 0
    alg.run_with_state(2, 10, (S2, state[1], state[2]))
    assert not np.array_equal(alg._dev[2].read(_lib.F_LVAL), ref._dev[2].read(_lib.F_LVAL))
    alg.close(); ref.close()


def test_warm_restart_keeps_the_iterate_and_restarts_the_sums():
    state = journal_graph(8, 75e-4, seed=2)
    a = _lib.Solver(12, state, 8, 0.05)
    a.iterate(8, None, seed=4)
    e0, l0, x0, y0 = (a.read(f) for f in (_lib.F_E_ACCU, _lib.F_LVAL, _lib.F_XVAL, _lib.F_Y))
    a.set_slots(11, 4, warm=True)
    assert a.iterations_done == 0 and a.Z == 11
    assert np.array_equal(a.read(_lib.F_E_ACCU), e0) and np.array_equal(a.read(_lib.F_LVAL), l0)
    assert np.array_equal(a.read(_lib.F_XAVG), x0) and np.array_equal(a.read(_lib.F_YAVG), y0)
    a.iterate(4, None, seed=5)
    assert abs(a.read(_lib.F_YAVG).sum() - 4) < 1e-9  # four terms: the kept Y and three new ones
    a.close()


def feasible(state, z_vec, Z):
    S, Q, h = state
    Sd = S.toarray()
    np.fill_diagonal(Sd, 0)
    for zz in range(Z):
        mem = np.where(z_vec == zz)[0]
        if mem.size and (np.any(Sd[np.ix_(mem, mem)].sum(axis=0) > h[mem] + 1e-12) or Q[np.ix_(mem, mem)].nnz):
            return False
    return True


def test_warm_started_search_reaches_a_feasible_colouring_like_the_cold_one():
    from sig_sdp_mmw_amd.binary_search import binary_search_relaxation
    state = journal_graph(10, 75e-4, seed=7)  # K = 300
    res = {}
    for warm in (False, True):
        bs = binary_search_relaxation()
        bs.verbose = False
        alg = mmw(nit=150, eta=0.04, dtype="f32", rng="device", seed=1, warm_start=warm)
        bs.feasibility_check_alg = alg
        np.random.seed(0)
        z_vec, Z, rem = bs.run(state)
        assert rem == 0 and feasible(state, z_vec, Z)
        iters = alg.LOGGED_NP_DATA["mmw_iters"][:, 5]
        res[warm] = (Z, iters)
        alg.close()
    assert np.all(res[False][1] == 150)
    assert res[True][1][0] == 150 and np.all(res[True][1][1:] == 50)  # first probe cold, later ones a third of the iterations
    assert abs(res[True][0] - res[False][0]) <= 1


def test_environment_defaults_put_the_unchanged_harness_on_the_fast_path(monkeypatch):
    monkeypatch.setenv("MMW_DTYPE", "f32")
    monkeypatch.setenv("MMW_RNG", "device")
    monkeypatch.setenv("MMW_EXPM_TOL", "1e-5")
    alg = mmw(nit=20, eta=0.04)  # exactly as sim_script/journal_version/sim_mmw_time.py:34 writes it
    assert (alg.dtype, alg.rng, alg.expm_tol, alg.round_batch) == ("f32", "device", 1e-5, True)
    state = journal_graph(8, 75e-4, seed=1)
    before = np.random.get_state()[2]
    np.random.seed(1)
    pos0 = np.random.get_state()[2]
    ok, X_half = alg.run_with_state(0, 12, state)
    assert np.random.get_state()[2] == pos0  # device sketches: the global NumPy stream is untouched by the loop
    assert alg._dev[2].dtype == _lib.F32 and X_half.shape == (state[0].shape[0], 22)
    alg.close()
    explicit = mmw(nit=20, eta=0.04, dtype="f64", rng="host")  # explicit arguments win over the environment
    assert (explicit.dtype, explicit.rng) == ("f64", "host")


def test_lagged_plans_give_the_exact_plans_result(monkeypatch):
    """Inside a chunk the exponential's plan is extrapolated from last iteration's matrix (row sums taken in k_dual_h, plan made
    by a spare workgroup of k_softmax_b) and verified one iteration later.  The run must agree with the one that plans every
    iteration exactly, to the tolerance of the exponential, without replays on a smoothly growing matrix."""
    state = journal_graph(16, 0.02, seed=4)  # K = 2048
    Z, nit = 24, 60
    a = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    a.iterate(nit, None, seed=9)
    got = [a.read(f) for f in FIELDS]
    assert a.read(_lib.F_BLOCKING)[3] == 0
    a.close()
    monkeypatch.setenv("MMW_NO_LAGGED_PLAN", "1")
    b = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    b.iterate(nit, None, seed=9)
    for f, x in zip(FIELDS, got):
        assert relerr(x, b.read(f)) < 2e-5, f
    b.close()


@pytest.mark.parametrize("dtype,tol", [(_lib.F32, 2e-5), (_lib.F64, 1e-9)])
def test_fused_dual_pass_gives_the_two_pass_softmax(dtype, tol, monkeypatch):
    """Inside a chunk the softmax rides in k_dual_h, shifted by the previous iteration's maximum, and the LOSS pass normalises
    (k_dual_scal folds the sums).  The softmax does not depend on the shift: the run agrees with the two-pass kernels to rounding,
    and with no replay."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 60
    a = _lib.Solver(Z, state, nit, 0.04, dtype=dtype)
    a.iterate(nit, None, seed=9)
    got = [a.read(f) for f in FIELDS]
    assert a.read(_lib.F_BLOCKING)[3] == 0
    a.close()
    monkeypatch.setenv("MMW_NO_FUSED_DUAL", "1")
    b = _lib.Solver(Z, state, nit, 0.04, dtype=dtype)
    b.iterate(nit, None, seed=9)
    for f, x in zip(FIELDS, got):
        assert relerr(x, b.read(f)) < tol, f
    b.close()


def test_fused_dual_pass_is_replayed_when_the_maximum_runs_away(monkeypatch):
    """k_dual_scal distrusts the fused pass's exponentials when e_accu's maximum has moved further from the shift than exp can
    absorb; with the allowance forced negative every chunk is restored and replayed on the two-pass kernels, and the result is
    the run that never took the fused pass."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 30
    monkeypatch.setenv("MMW_NO_LAGGED_PLAN", "1")
    monkeypatch.setenv("MMW_DUAL_GAP", "-1")
    a = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    a.iterate(nit, None, seed=9)
    got = [a.read(f) for f in FIELDS]
    assert a.read(_lib.F_BLOCKING)[3] > 0  # replays
    a.close()
    monkeypatch.delenv("MMW_DUAL_GAP")
    monkeypatch.setenv("MMW_SYNC_PLAN", "1")
    b = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    b.iterate(nit, None, seed=9)
    assert b.read(_lib.F_BLOCKING)[3] == 0
    for f, x in zip(FIELDS, got):
        assert relerr(x, b.read(f)) < 1e-6, f  # as in test_replayed_chunk_equals_the_synchronous_run
    b.close()


def test_profiling_the_shipped_path_does_not_change_it():
    """mmw_set_profile(s, 2) brackets the launches of the shipped path as they are (chunks without readback, sketch and lagged plan
    riding in the LOSS launch): same bits as the unprofiled run, no stand-alone sketch launches; mode 1 (synchronous, every class
    in launches of its own) counts one sketch launch per iteration."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 40
    a = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    a.iterate(nit, None, seed=9)
    plain = [a.read(f) for f in FIELDS]
    a.reset(nit)
    a.set_profile(2)
    a.iterate(nit, None, seed=9)
    kt2 = a.kernel_times()
    for f, x in zip(FIELDS, plain):
        assert np.array_equal(x, a.read(f)), f
    assert kt2["spmm"][1] >= nit and kt2["dual"][1] == nit and kt2["loss"][1] == nit
    assert kt2["sketch"][1] <= 1 + nit // 4  # only a chunk's first iteration draws its sketch in a launch of its own
    a.reset(nit)
    a.set_profile(1)
    a.iterate(nit, None, seed=9)
    kt1 = a.kernel_times()
    assert kt1["sketch"][1] == nit
    a.set_profile(0)
    a.close()


def test_speculative_search_reaches_a_feasible_colouring_like_the_sequential_one():
    """binary_search.speculate (opt-in): while `mid` is solved, the slot count that follows if it is feasible is solved on a second
    handle in a second host thread; roundings run in the search's order.  Same update rules, so the search ends on a feasible
    colouring within one slot of the sequential search's, after no more deciding probes than it."""
    from sig_sdp_mmw_amd.binary_search import binary_search_relaxation
    from sig_sdp_mmw_amd.mmw import mmw
    state = journal_graph(16, 0.02, seed=3)
    res = {}
    for spec in (False, True):
        bs = binary_search_relaxation()
        bs.verbose = False
        bs.speculate = spec
        alg = mmw(nit=60, eta=0.04, dtype="f32", rng="device", seed=2, warm_start=True)
        bs.feasibility_check_alg = alg
        np.random.seed(0)
        z_vec, Z, rem = bs.run(state)
        alg.close()
        assert rem == 0 and z_vec.min() >= 0 and z_vec.max() < Z
        res[spec] = (Z, bs.LOGGED_NP_DATA["bs_search_per_it"].shape[0], bs.LOGGED_NP_DATA.get("bs_speculation"))
    assert abs(res[True][0] - res[False][0]) <= 1
    assert res[True][1] <= res[False][1] + 1
    assert res[True][2] is not None and res[False][2] is None



def test_max_violation_field_is_the_maximum_of_e_this():
    state = journal_graph(8, 75e-4, seed=2)
    a = _lib.Solver(10, state, 6, 0.05, dtype=_lib.F32)
    a.iterate(6, None, seed=4)
    assert a.read(_lib.F_E_MAX)[0] == np.max(a.read(_lib.F_E_THIS))
    a.close()
    # the value mmw_sync fetched behind the run (no device work at the read) is the same number, also across a reset and a replayed chunk
    b = _lib.Solver(10, state, 40, 0.05, dtype=_lib.F32)
    for rounds in range(2):
        b.iterate(25, None, seed=4)
        b.sync()
        assert b.read(_lib.F_E_MAX)[0] == np.max(b.read(_lib.F_E_THIS))
        b.iterate(15, None, seed=4)
        b.sync()
        assert b.read(_lib.F_E_MAX)[0] == np.max(b.read(_lib.F_E_THIS))
        b.reset(40)
    b.close()


@pytest.mark.parametrize("dtype,tol", [(_lib.F32, 2e-5), (_lib.F64, 1e-9)])
def test_chained_chunks_give_the_result_of_chunks_that_restart_exactly(dtype, tol, monkeypatch):
    """A handle that has not replayed lets a chunk's first iteration continue on the lagged plan and the shifted softmax of the
    previous chunk (and sizes its chunks by the room the error estimate leaves).  Several calls of different lengths, so that chunk
    boundaries fall everywhere: same result as with chunks that restart exactly, to the tolerance of the exponential, no replay."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 90
    def run():
        a = _lib.Solver(Z, state, nit, 0.04, dtype=dtype)
        for n in (5, 20, 3, 40, 22):
            a.iterate(n, None, seed=9)
        out = [a.read(f) for f in FIELDS]
        replays = a.read(_lib.F_BLOCKING)[3]
        a.close()
        return out, replays
    got, replays = run()
    assert replays == 0
    monkeypatch.setenv("MMW_NO_CHUNK_CHAIN", "1")
    ref, _ = run()
    for f, x, y in zip(FIELDS, got, ref):
        assert relerr(x, y) < tol, f


def test_row_sums_from_the_matrix_core_sddmm_give_the_separate_pass(monkeypatch):
    """Inside a call the DUAL phase takes the row sums of the off-diagonal X from slabs the matrix-core SDDMM leaves (one slot per
    union-tile group, summed from the accumulators) instead of k_dual_rows, and makes the D / F violations itself.  Same
    numbers as the separate pass to fp32 rounding, in the chunked run and in the run that reads every plan back."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 60
    a = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    assert a.read(_lib.F_SPMM_KIND)[0] == 3.0
    a.iterate(nit, None, seed=9)
    got = [a.read(f) for f in FIELDS] + [a.read(_lib.F_E_THIS)]
    assert a.read(_lib.F_DUAL_INFO)[0] >= nit - 8, "all but the first iteration of a chunk that restarts are expected on the slabs"
    a.close()
    monkeypatch.setenv("MMW_SYNC_PLAN", "1")
    c = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    c.iterate(nit, None, seed=9)
    assert c.read(_lib.F_DUAL_INFO)[0] == nit - 1
    for f, x in zip(FIELDS + (_lib.F_E_THIS,), got):
        assert relerr(x, c.read(f)) < 2e-5, f
    c.close()
    monkeypatch.setenv("MMW_NO_SDDMM_ROWSUMS", "1")
    b = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    b.iterate(nit, None, seed=9)
    assert b.read(_lib.F_DUAL_INFO)[0] == 0
    for f, x in zip(FIELDS + (_lib.F_E_THIS,), got):
        assert relerr(x, b.read(f)) < 2e-5, f
    b.close()


# this graph's matrix norm grows four times faster than the benchmark instance's: the step size that keeps a whole run inside what the
# first-order product's certificate admits (truncation + measured fp16 rounding <= tol, kernels_mfma.h first_verify)
ETA_FIRST = 0.01


def test_first_order_exponential_gives_the_lanczos_run(monkeypatch):
    """While one Lanczos step is accepted with room, a chunk takes exp(L/2)R as ONE product, y = u + (L/2 - mu I)u, certified after the
    fact by ||A'u|| rho/2 e^(2 rho) per column (no scalar launch, no combination).  Both forms meet the same tolerance: the runs agree to
    it, and the first-order run replays nothing."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 60
    a = _lib.Solver(Z, state, nit, ETA_FIRST, dtype=_lib.F32)
    a.iterate(nit, None, seed=9)
    got = [a.read(f) for f in FIELDS]
    info = a.read(_lib.F_DUAL_INFO)
    assert info[2] >= nit // 3, info  # chunks after the first plan readbacks
    assert a.read(_lib.F_BLOCKING)[3] == 0
    a.close()
    monkeypatch.setenv("MMW_NO_FIRST_ORDER", "1")
    b = _lib.Solver(Z, state, nit, ETA_FIRST, dtype=_lib.F32)
    b.iterate(nit, None, seed=9)
    assert b.read(_lib.F_DUAL_INFO)[2] == 0
    for f, x in zip(FIELDS, got):
        assert relerr(x, b.read(f)) < 2e-5, f
    b.close()


def test_first_order_exponential_is_replayed_when_its_bound_misses_the_tolerance(monkeypatch):
    """A tolerance tightened between two calls makes the chunk that was planned on the old one miss its bound: the verification that rides
    in the SDDMM launch raises the replay flag and the chunk is redone with Lanczos steps -- the result is the synchronous run's."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 48
    a = _lib.Solver(Z, state, nit, ETA_FIRST, dtype=_lib.F32)
    a.iterate(32, None, seed=9)
    a.sync()
    first0 = a.read(_lib.F_DUAL_INFO)[2]
    assert first0 > 0 and a.read(_lib.F_BLOCKING)[3] == 0
    a.set_expm(_lib.EXPM_LANCZOS, 12, 1e-9)
    a.iterate(16, None, seed=9)
    a.sync()
    assert a.read(_lib.F_BLOCKING)[3] >= 1, "the chunk planned on the old tolerance is expected to be replayed"
    got = [a.read(f) for f in FIELDS]
    a.close()
    monkeypatch.setenv("MMW_SYNC_PLAN", "1")
    b = _lib.Solver(Z, state, nit, ETA_FIRST, dtype=_lib.F32)
    b.iterate(32, None, seed=9)
    b.set_expm(_lib.EXPM_LANCZOS, 12, 1e-9)
    b.iterate(16, None, seed=9)
    for f, x in zip(FIELDS, got):
        assert relerr(x, b.read(f)) < 2e-5, f
    b.close()


def test_first_order_product_with_the_matrix_in_one_fp16_half(monkeypatch):
    """Early in a run (2 * 2^-12 max_i sum_j |a_ij| <= tol, ExpmPlan::f16a_ok) the first-order product reads the matrix as ONE fp16 half from
    an image of its own: half the bytes of the matrix, one product per tile and k-step.  A small step size keeps the whole run inside
    that gate; the run agrees with the two-halves form to the tolerance both meet."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit, eta = 24, 48, 0.004
    a = _lib.Solver(Z, state, nit, eta, dtype=_lib.F32)
    a.iterate(nit, None, seed=9)
    got = [a.read(f) for f in FIELDS]
    info = a.read(_lib.F_DUAL_INFO)
    assert info[3] >= nit // 3 and info[3] <= info[2], info
    assert a.read(_lib.F_BLOCKING)[3] == 0
    a.close()
    monkeypatch.setenv("MMW_NO_FIRST_A16", "1")
    b = _lib.Solver(Z, state, nit, eta, dtype=_lib.F32)
    b.iterate(nit, None, seed=9)
    ib = b.read(_lib.F_DUAL_INFO)
    assert ib[3] == 0 and ib[2] > 0, ib
    for f, x in zip(FIELDS, got):
        assert relerr(x, b.read(f)) < 2e-5, f
    b.close()


def test_sampled_phase_timers_keep_one_row_per_iteration():
    """mmw_set_timing(S > 1): events in iteration 0 and in one iteration of every S; MMW_F_PHASE_US still has a row per iteration
    (the harness takes means over them, sim_mmw_time.py:48-52) and the iterate is the one of an untimed run."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 40
    ref = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
    ref.iterate(nit, None, seed=5)
    ref.sync()
    want = ref.read(_lib.F_XVAL)
    ref.close()
    rows = {}
    for stride in (1, 8):
        s = _lib.Solver(Z, state, nit, 0.04, dtype=_lib.F32)
        s.set_timing(stride)
        s.iterate(25, None, seed=5)
        s.iterate(nit - 25, None, seed=5)
        s.sync()
        us = s.read(_lib.F_PHASE_US).reshape(nit, 4)
        assert np.all(us > 0) and np.all(np.isfinite(us))
        assert np.allclose(us[:, :3].sum(axis=1), us[:, 3], rtol=0.05, atol=2.0)  # the three phases make the iteration
        assert relerr(s.read(_lib.F_XVAL), want) < 1e-6
        rows[stride] = us
        s.close()
    assert len(np.unique(rows[8][8:16, 3])) == 1 and len(np.unique(rows[1][8:16, 3])) > 1  # a group repeats its sample
    assert 0.5 < rows[8][:, 3].mean() / rows[1][:, 3].mean() < 1.5
