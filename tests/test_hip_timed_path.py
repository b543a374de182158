"""The path bench.py times, under the oracle.

`mmw_iterate(n > 1, NULL, seed)` is the shipped fast path: chunks enqueued without plan readback, lagged plans, the softmax
inside the violation pass, row sums of X from the matrix-core SDDMM, and exp(L/2)R as ONE certified first-order product on
fp16 operands (csrc/mmw_api.hip, `optimistic`).  Every other trajectory test uploads its sketches and therefore runs the
synchronous path.  Here the run draws its sketches on the device; `mmw_sketch` regenerates the Philox block of every
(seed, iteration) -- the generator is counter-based, so the blocks are exactly the ones the chunks multiplied -- and the
CPU oracle (reference loop mmw.py:124-197 restated) follows the same run on them.  Compared: every quantity of the loop
at a chunk boundary in the middle of the run and at its end, the running sums included.

Bars (north star): exp(L/2)R <= 1e-5 relative Frobenius; everything else <= 1e-4.
"""
import numpy as np
import pytest

from conftest import relerr
from oracle import mmw_oracle as orc
from sig_sdp_mmw_amd import _lib
from sig_sdp_mmw_amd.graphs import er_contention_graph, journal_graph

pytestmark = pytest.mark.gpu

FIELDS = [("e_this", _lib.F_E_THIS), ("e_accu", _lib.F_E_ACCU), ("Y", _lib.F_Y), ("lval", _lib.F_LVAL), ("xval", _lib.F_XVAL),
          ("X_half", _lib.F_XHALF)]


def snapshot(s):
    d = {name: s.read(f) for name, f in FIELDS}
    d["xsum"] = s.read(_lib.F_XAVG)
    d["ysum"] = s.read(_lib.F_YAVG)
    return d


def follow(state, Z, nit, n1, eta, seed, dtype=_lib.F32):
    """Device run in two calls (each chunked by the library), then the oracle on the regenerated sketches."""
    s = _lib.Solver(Z, state, nit, eta, dtype=dtype)
    s.set_expm(_lib.EXPM_LANCZOS, 12, 1e-6 if dtype == _lib.F32 else 1e-12)  # bench.py's settings (fp32) / tight for the fp64 bars
    s.iterate(n1, None, seed)
    mid = snapshot(s)
    s.iterate(nit - n1, None, seed)
    end = snapshot(s)
    info = s.read(_lib.F_DUAL_INFO)
    replays = s.read(_lib.F_BLOCKING)[3]
    o = orc.MMWOracle(nit=nit, eta=eta)
    o.run(Z, state, lambda i, K, D: s.sketch(seed, i), keep_trace={n1 - 1, nit - 1}, factor=False)
    s.close()
    return mid, end, o, info, replays


def compare(got, o, idx, last, nit, bars=(1e-5, 1e-4)):
    tr = o.trace
    for name, _ in FIELDS:
        bar = bars[0] if name == "X_half" else bars[1]
        err = relerr(got[name], tr[name][idx])
        assert err < bar, (name, "iteration", tr["iters"][idx], err)
    if last:  # a full run: nit terms each, the last X / Y are not averaged (mmw.py:77-78,203)
        assert relerr(got["xsum"] / nit, o.xavg) < bars[1]
        assert relerr(got["ysum"] / nit, o.yavg) < bars[1]
    else:     # what mmw.py:77-78 hold when the next iteration starts
        assert relerr(got["xsum"], tr["xsum"][idx]) < bars[1]
        assert relerr(got["ysum"], tr["ysum"][idx]) < bars[1]


def test_fast_path_trajectory_small_journal():
    """K ~ 2 k, 60 iterations in two calls: the chunks after the first plan readbacks take the first-order form (a small step size keeps
    the matrix inside what the certificate admits for the whole run: this graph's norm grows four times faster than the benchmark's)."""
    state = journal_graph(16, 0.02, seed=4)
    mid, end, o, info, replays = follow(state, 24, 60, 32, 0.01, seed=9)
    assert info[0] > 0 and info[1] > 0 and info[2] > 0, info  # row sums from the SDDMM, fused softmax, first-order products
    assert replays == 0
    compare(mid, o, 0, False, 60)
    compare(end, o, 1, True, 60)


def test_fast_path_trajectory_fp64_class_default():
    """The class default dtype (fp64) on the chunked device-RNG path: LDS-staged kernels, Lanczos steps stopped a posteriori, lagged plans,
    the softmax inside the violation pass -- against the oracle at fp64 bars (1e-8 / 1e-8)."""
    state = journal_graph(16, 0.02, seed=4)
    mid, end, o, info, replays = follow(state, 24, 40, 20, 0.04, seed=21, dtype=_lib.F64)
    assert info[1] > 0, info  # fused softmax ran
    compare(mid, o, 0, False, 40, bars=(1e-8, 1e-8))
    compare(end, o, 1, True, 40, bars=(1e-8, 1e-8))


@pytest.mark.timeout(1500)
def test_fast_path_trajectory_bench_instance():
    """The benchmark's instance (journal-1pct, N = 10 003, Z = 186, D = 372, fp32), 48 iterations, two calls of 24."""
    state = journal_graph(28, 0.0319, 0)
    mid, end, o, info, replays = follow(state, 186, 48, 24, 0.04, seed=1234)
    assert info[0] > 0 and info[1] > 0, info
    assert info[2] > 0 and info[3] > 0, info  # first-order products, some with the matrix in one fp16 half
    assert replays == 0
    compare(mid, o, 0, False, 48)
    compare(end, o, 1, True, 48)


@pytest.mark.timeout(1500)
def test_fast_path_trajectory_er_1pct():
    """Erdos-Renyi N = 10 000, 1 % (no locality: generic kernels, no matrix cores): chunks, lagged plans and the fused softmax."""
    state = er_contention_graph(10000, 0.01, 0)
    mid, end, o, info, replays = follow(state, 32, 40, 20, 0.04, seed=77)
    assert info[1] > 0, info  # the softmax ran inside the violation pass
    compare(mid, o, 0, False, 40)
    compare(end, o, 1, True, 40)


def test_sketch_regeneration_is_the_block_the_run_multiplied():
    """mmw_sketch(seed, i) against MMW_F_SKETCH (the last iteration's block, regenerated by the handle itself) and against
    exp(0)R = R of a first iteration; another iteration or seed gives another block."""
    state = journal_graph(16, 0.02, seed=4)
    s = _lib.Solver(24, state, 8, 0.0, dtype=_lib.F32)
    s.iterate(1, None, seed=3)
    r0 = s.sketch(3, 0)
    assert np.array_equal(r0, s.read(_lib.F_SKETCH))
    assert relerr(s.read(_lib.F_XHALF), r0) < 1e-6  # eta = 0: L = 0
    s.iterate(5, None, seed=3)
    assert np.array_equal(s.sketch(3, 5), s.read(_lib.F_SKETCH))
    assert not np.array_equal(s.sketch(3, 4), s.sketch(3, 5)) and not np.array_equal(s.sketch(4, 5), s.sketch(3, 5))
    assert np.allclose(np.linalg.norm(r0, axis=1), 1.0, atol=1e-6)
    s.close()


def test_first_order_certificate_counts_the_fp16_rounding(monkeypatch):
    """The certificate of the first-order product includes what the fp16 plane of u lost, measured by the sketch kernel.  Inflating
    that measure (MMW_FV_DU_SCALE) makes the certificate miss: the chunk is replayed with Lanczos steps and the result is the
    synchronous run's."""
    state = journal_graph(16, 0.02, seed=4)
    Z, nit = 24, 48
    monkeypatch.setenv("MMW_FV_DU_SCALE", "200")
    a = _lib.Solver(Z, state, nit, 0.01, dtype=_lib.F32)
    a.iterate(nit, None, seed=9)
    a.sync()
    assert a.read(_lib.F_DUAL_INFO)[2] > 0 and a.read(_lib.F_BLOCKING)[3] >= 1, (a.read(_lib.F_DUAL_INFO), a.read(_lib.F_BLOCKING))
    got = {name: a.read(f) for name, f in FIELDS}
    a.close()
    monkeypatch.delenv("MMW_FV_DU_SCALE")
    monkeypatch.setenv("MMW_SYNC_PLAN", "1")
    b = _lib.Solver(Z, state, nit, 0.01, dtype=_lib.F32)
    b.iterate(nit, None, seed=9)
    for name, f in FIELDS:
        assert relerr(got[name], b.read(f)) < 2e-5, name
    b.close()
