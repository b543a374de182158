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

    def set_bounds(self, state):
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

    def search(self, left, right, state):
        alg = self.feasibility_check_alg
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
