"""The outer bisection (sig_sdp_mmw_amd.binary_search) against the reference's recorded searches:
a stub solver replays the recorded remainders, the search must visit the same (left, right, mid) triples
(binary_search_relaxation.py:44-72).  CPU-only."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, state_from
from sig_sdp_mmw_amd.binary_search import binary_search_relaxation


class Replay:
    """run_with_state / rounding protocol that answers with the reference's recorded remainders."""

    def __init__(self, per_it):
        self.rem_by_call = [int(r[4]) for r in per_it]
        self.calls = []

    def run_with_state(self, it, Z, state):
        self.calls.append(Z)
        return True, np.zeros((state[0].shape[0], 2))

    def rounding(self, Z, gX, state):
        rem = self.rem_by_call[len(self.calls) - 1]
        return np.zeros(state[0].shape[0]), Z, rem


@pytest.mark.parametrize("name", ["env75", "env108"])
def test_search_reproduces_recorded_sequence(name):
    g = load_golden("bs_run")
    state = state_from(g, name + "_")
    per_it = g[name + "_per_it"]
    bs = binary_search_relaxation()
    bs.verbose = False
    bs.feasibility_check_alg = Replay(per_it)
    lb, ub = bs.set_bounds(state)
    assert [lb, ub] == [int(x) for x in g[name + "_bounds"][0]]
    z_vec, Z, rem = bs.run(state)
    mine = bs.LOGGED_NP_DATA["bs_search_per_it"][:, 3:8]
    assert mine.shape == per_it.shape
    assert np.array_equal(mine[:, :5], per_it[:, :5])  # left, right, mid, Z, rem per step
    assert Z == int(g[name + "_Z"]) and rem == int(g[name + "_rem"])
    assert bs.LOGGED_NP_DATA["bs_search"].shape == (1, 9)


def test_forced_bounds():
    g = load_golden("bs_run")
    state = state_from(g, "env75_")
    bs = binary_search_relaxation()
    bs.force_lower_bound = True
    lb, ub = bs.set_bounds(state)
    assert lb == ub == int(np.max(np.diff(state[1].indptr))) + 1
    bs.force_lower_bound = False
    bs.force_full_bound = True
    assert bs.set_bounds(state) == (1, state[0].shape[0])


def test_dropin_overlay_resolves_to_this_solver():
    """`from sim_src.alg.mmw import mmw` with dropin/ first on the path gives the MI355X class."""
    import subprocess
    code = ("import sys; sys.path[:0]=[%r, %r]; import sim_src.alg.mmw as m, sim_src.alg as a; "
            "assert m.mmw.__module__ == 'sig_sdp_mmw_amd.mmw'; assert hasattr(a, 'alg_interface'); print('ok')") % (
        os.path.join(ROOT, "dropin"), ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr
    # the harness writes mmw(nit=150, eta=0.04) (sim_script/journal_version/sim_mmw_time.py:34): the environment alone puts that
    # unchanged call on the device-RNG / fp32 / warm-started path
    code = ("import sys; sys.path[:0]=[%r, %r]; from sim_src.alg.mmw import mmw; a = mmw(nit=150, eta=0.04); "
            "assert (a.dtype, a.rng, a.expm_tol, a.warm_start, a.round_batch) == ('f32', 'device', 1e-6, True, True), (a.dtype, a.rng); print('ok')") % (
        os.path.join(ROOT, "dropin"), ROOT)
    env = dict(os.environ, MMW_DTYPE="f32", MMW_RNG="device", MMW_EXPM_TOL="1e-6", MMW_WARM_START="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


class Threshold:
    """Deterministic stand-in with the protocol the speculative search needs: a slot count is feasible iff it is at least `zmin`;
    every solve is recorded (thread-safe append), `sibling()` shares the record."""
    rng = "device"

    def __init__(self, zmin, log=None):
        self.zmin = zmin
        self.log = [] if log is None else log
        self.closed = False

    def sibling(self):
        return Threshold(self.zmin, self.log)

    def prepare(self, Z, state):
        self.log.append(("prepare", Z))

    def run_with_state(self, it, Z, state):
        self.log.append(("solve", Z))
        return True, np.full((state[0].shape[0], 2), float(Z))

    def rounding(self, Z, gX, state):
        assert gX[0, 0] == float(Z)  # the rounding gets the factor of the slot count it is asked about
        return np.zeros(state[0].shape[0]), Z, (0 if Z >= self.zmin else self.zmin - Z)

    def close(self):
        self.closed = True


@pytest.mark.parametrize("zmin", [2, 3, 7, 11, 12, 13, 20])
def test_speculative_search_decides_like_the_sequential_one(zmin):
    """binary_search.speculate: probes solved ahead are used only when the search arrives at them; the deciding sequence of
    (left, right, mid, remainder) rows is the sequential search's, row for row."""
    g = load_golden("bs_run")
    state = state_from(g, "env75_")
    rows = {}
    for spec in (False, True):
        bs = binary_search_relaxation()
        bs.verbose = False
        bs.speculate = spec
        alg = Threshold(zmin)
        bs.feasibility_check_alg = alg
        z_vec, Z, rem = bs.run(state)
        assert rem == 0 and Z >= zmin
        rows[spec] = bs.LOGGED_NP_DATA["bs_search_per_it"][:, 3:8]  # left, right, mid, Z, rem
        if spec:
            solved = [z for kind, z in alg.log if kind == "solve"]
            assert len(solved) >= rows[spec].shape[0]  # every deciding probe was solved, some more were solved ahead
            assert "bs_speculation" in bs.LOGGED_NP_DATA
    assert np.array_equal(rows[True], rows[False])
