"""Outer bisection on the slot count Z -- the caller of the hot path.

Behavioural restatement of `binary_search_relaxation` (sim_src/alg/binary_search_relaxation.py:8-72):
same attributes (`feasibility_check_alg`, `force_lower_bound`, `force_full_bound`), same bounds
(:13-29: lower = max association degree + 1, upper = max stored row length of S + S^T, + 1), same
update rules (:57-67) and the same log tables (`bs_set_bounds`, `bs_search`, `bs_search_per_it`).
It drives any object with `run_with_state(it, Z, state)` / `rounding(Z, gX, state)`.
"""
import math

import numpy as np

from .stats import STATS_OBJECT


class binary_search_relaxation(STATS_OBJECT):
    def __init__(self):
        self.feasibility_check_alg = None
        self.force_lower_bound = False
        self.force_full_bound = False
        self.verbose = True
        # New, opt-in (the reference probes one slot count at a time, :44-72): while `mid` is being solved, the slot count the
        # search would try next if `mid` turns out feasible is solved on a second handle (own stream, own host thread).  The
        # roundings still run one after the other in the search's order, so the global NumPy stream is consumed deterministically.
        self.speculate = False

    def set_bounds(self, state):
        if hasattr(state, "env") and hasattr(state.env, "bounds") and not (self.force_lower_bound or self.force_full_bound):
            return state.env.bounds()  # a device-resident state (_lib.DeviceState): the same bounds from the device's count pass
        S, Q = state[0], state[1]
        if self.force_lower_bound:
            lb = int(np.max(np.diff(Q.indptr))) + 1
            return lb, lb
        if self.force_full_bound:
            return 1, S.shape[0]
        # The reference zeroes the diagonal of S + S^T with setdiag(0) (:23), which leaves one STORED entry per row whether or
        # not the row had a diagonal before: its count is (off-diagonal entries of the row) + 1, and the bound adds one more.
        sym = (S + S.transpose()).tocsr()
        sym.sum_duplicates()
        rows = np.repeat(np.arange(sym.shape[0]), np.diff(sym.indptr))
        offdiag = np.bincount(rows[sym.indices != rows], minlength=sym.shape[0])
        ub = int(np.max(offdiag)) + 2
        lb = int(np.max(np.diff(Q.indptr))) + 1
        return lb, ub

    def run(self, state):
        tic = self._get_tic()
        left, right = self.set_bounds(state)
        self._add_np_log("bs_set_bounds", 0, np.array([left, right, self._get_tim(tic)]))
        tic = self._get_tic()
        Z, z_vec, rem, it = self.search(left, right, state)
        self._add_np_log("bs_search", 0, np.array([left, right, Z, rem, it, self._get_tim(tic)]))
        return z_vec, Z, rem

    @staticmethod
    def _step(left, right, mid, rem):
        """The update rules of :57-67: (left, right, done) after a probe of `mid` left `rem` users unassigned."""
        if left < right and rem > 0:
            return mid + 1, right, False
        if left + 1 < right and rem == 0:
            return left, mid, False
        if rem == 0:  # left + 1 == right, or left >= right
            return left, right, True
        return left + 1, right + 1, False  # left >= right and rem > 0: the bracket was too tight, slide it up

    def search(self, left, right, state):
        alg = self.feasibility_check_alg
        if self.speculate and hasattr(alg, "sibling") and getattr(alg, "rng", None) == "device":
            return self._search_speculative(left, right, state)
        it = 0
        while True:
            mid = math.floor(float(left + right) / 2.)
            tic = self._get_tic()
            _, gX = alg.run_with_state(it, mid, state)
            t_solve = self._get_tim(tic)
            tic = self._get_tic()
            z_vec, Z, rem = alg.rounding(mid, gX, state)
            t_round = self._get_tim(tic)
            self._add_np_log("bs_search_per_it", it, np.array([left, right, mid, Z, rem, t_solve, t_round]))
            it += 1
            done = False
            if left < right and rem > 0:
                left = mid + 1
            elif left + 1 < right and rem == 0:
                right = mid
            elif rem == 0:  # left + 1 == right, or left >= right
                done = True
            else:  # left >= right and rem > 0: the bracket was too tight, slide it up
                left += 1
                right += 1
            if self.verbose:
                self._printalltime(left, right, mid, Z, rem, "++++++++++++++++++++")
            if done:
                return Z, z_vec, rem, it

    def _search_speculative(self, left, right, state):
        """Two probes in flight: `mid` on the algorithm object, and on its sibling the slot count that follows if `mid` is feasible
        (the common case while the upper bound is loose).  A speculative solve whose premise fails is dropped; the sequence of
        probes that decide the search, their roundings and the log rows are those of `search`."""
        import threading
        import time
        algs = [self.feasibility_check_alg, self.feasibility_check_alg.sibling()]
        ready = {}  # slot count -> (alg index, gX, solve time in us): solved ahead, not yet used
        it = 0
        wasted = 0

        def solve(idx, Z, out):
            t0 = time.time()
            try:
                _, gX = algs[idx].run_with_state(it, Z, state)
                out[Z] = (idx, gX, (time.time() - t0) * 1e6, None)
            except BaseException as e:  # surfaces in the caller's thread
                out[Z] = (idx, None, 0.0, e)

        def prepare(idx, Z, out):  # first round: the second handle's state processing only, under the first probe
            try:
                algs[idx].prepare(Z, state)
            except BaseException as e:
                out[-1] = (idx, None, 0.0, e)

        try:
            while True:
                mid = math.floor(float(left + right) / 2.)
                if mid not in ready:
                    l2, r2, done2 = self._step(left, right, mid, 0)
                    nxt = None if done2 else math.floor(float(l2 + r2) / 2.)
                    out = {}
                    jobs = [threading.Thread(target=solve, args=(0, mid, out))]  # fixed roles: both handles see descending slot counts
                    if nxt is not None and nxt != mid and nxt >= 2:
                        # the first probe is the large one (state processing + the widest solve): a second solve beside it slows
                        # both down by more than it saves; its thread only builds the second handle then
                        jobs.append(threading.Thread(target=prepare if it == 0 and hasattr(algs[1], "prepare") else solve, args=(1, nxt, out)))
                    for j in jobs:
                        j.start()
                    for j in jobs:
                        j.join()
                    for Z, rec in out.items():
                        if rec[3] is not None:
                            raise rec[3]
                    wasted += len(ready)  # solved ahead for a branch the search did not take
                    ready = out
                idx, gX, t_solve, _ = ready.pop(mid)
                tic = self._get_tic()
                z_vec, Z, rem = algs[idx].rounding(mid, gX, state)
                t_round = self._get_tim(tic)
                self._add_np_log("bs_search_per_it", it, np.array([left, right, mid, Z, rem, t_solve, t_round]))
                it += 1
                left, right, done = self._step(left, right, mid, rem)
                if self.verbose:
                    self._printalltime(left, right, mid, Z, rem, "++++++++++++++++++++")
                if done:
                    wasted += len(ready)
                    self._add_np_log("bs_speculation", 0, np.array([it, wasted]))
                    return Z, z_vec, rem, it
                if rem > 0 and ready:  # the premise of what was solved ahead failed
                    wasted += len(ready)
                    ready = {}
        finally:
            algs[1].close()
