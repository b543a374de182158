"""Logging / timing mixin with the surface the reference's harness reads from a solver object.

Mirrors the observable behaviour of `STATS_OBJECT` (sim_src/util.py:112-217): in-memory tables
`LOGGED_NP_DATA[key]` whose rows are `[g_step, step, time(), *payload]` (3 header columns,
util.py:112,149-162), microsecond tic/tim timers (util.py:201-217), `_print` gated by `DEBUG` and the
step window (util.py:168-171), `_printalltime` (util.py:173-174), `save_np` (util.py:136-147).
The harness scripts index these tables by column, e.g. `alg.LOGGED_NP_DATA["mmw_expm"][:,5]`
(sim_script/journal_version/sim_mmw_time.py:48-52), so the layout is part of the drop-in contract.
"""
import os
import pprint
from time import time

import numpy as np

HEADER_COLUMNS = 3


class STATS_OBJECT:
    N_STEP = 0
    DISABLE_ALL_DEBUG = False
    DEBUG_STEP = 100
    DEBUG = False
    LOGGED_CLASS_NAME = None
    PRINT_DIM = 5

    # class-level defaults; every instance gets its own containers on first use
    LOGGED_NP_DATA = {}

    def _tables(self):
        if "LOGGED_NP_DATA" not in self.__dict__:
            self.LOGGED_NP_DATA = {}
        return self.LOGGED_NP_DATA

    def _add_np_log(self, key, step, float_row_data, g_step=0):
        tables = self._tables()
        payload = np.atleast_1d(np.squeeze(np.asarray(float_row_data, dtype=np.float64)))
        if payload.ndim != 1:
            raise ValueError("log rows must be scalars or 1-D")
        width = payload.size + HEADER_COLUMNS
        if key not in tables:
            tables[key] = np.zeros((0, width))
        if tables[key].shape[1] != width:
            raise ValueError("log table %r has %d columns, got a row of %d" % (key, tables[key].shape[1], width))
        row = np.concatenate(([g_step, step, time()], payload))
        tables[key] = np.vstack((tables[key], row))

    def _add_np_log_rows(self, key, steps, payload_rows, g_step=0):
        """Append many rows at once (one per device iteration) without the quadratic vstack."""
        tables = self._tables()
        payload_rows = np.asarray(payload_rows, dtype=np.float64)
        n = payload_rows.shape[0]
        block = np.empty((n, payload_rows.shape[1] + HEADER_COLUMNS))
        block[:, 0] = g_step
        block[:, 1] = np.asarray(steps, dtype=np.float64)
        block[:, 2] = time()
        block[:, HEADER_COLUMNS:] = payload_rows
        if key not in tables:
            tables[key] = block
        else:
            tables[key] = np.vstack((tables[key], block))

    def save_np(self, path, postfix):
        os.makedirs(path, exist_ok=True)
        owner = self.LOGGED_CLASS_NAME or self.__class__.__name__
        for key, table in self._tables().items():
            np.savetxt(os.path.join(path, "%s.%s.%s.txt" % (owner, key, postfix)), table, delimiter=",")

    def save(self, path, postfix):
        pass

    def status(self):
        if self.DEBUG:
            pprint.pprint(vars(self))

    def _debug(self, ENABLE, debug_step=100):
        self.DEBUG = ENABLE
        self.DEBUG_STEP = debug_step

    def _print(self, *args, **kwargs):
        if self.DEBUG and not STATS_OBJECT.DISABLE_ALL_DEBUG and (self.N_STEP % self.DEBUG_STEP) in (0, 1, 2):
            print(("%6d\t" % self.N_STEP) + " ".join(map(str, args)), **kwargs)

    def _printalltime(self, *args, **kwargs):
        print(("%6d\t" % self.N_STEP) + ("%10s\t" % self.__class__.__name__) + " ".join(map(str, args)), **kwargs)

    # ---- microsecond timers keyed by a ticket number ------------------------------------------
    def _get_tic(self):
        if "_tics" not in self.__dict__:
            self._tics = {}
            self._ntic = 0
        self._ntic += 1
        self._tics[self._ntic] = time()
        return self._ntic

    def _get_tim(self, tic_id):
        try:
            t0 = self._tics.pop(tic_id)
        except (KeyError, AttributeError):
            raise Exception("no timer is found.")
        return (time() - t0) * 1e6
