"""Drop-in `mmw` solver class on MI355X.

Same class surface as the reference's `mmw` (sim_src/alg/mmw.py:12-229):

    alg = mmw(nit=150, eta=0.04)                 # + rank_radio, alpha, log_gap
    ok, X_half = alg.run_with_state(it, Z, state)    # state = (S_gain csr, Q_asso csr, h_max)
    z_vec, Z, rem = alg.rounding(Z, X_half, state)
    alg.LOGGED_NP_DATA["mmw_expm"][:, 5]             # per-iteration phase times in us
    alg.LOG_GAP, alg.DEBUG, alg._get_tic(), alg._get_tim(tic)

Every per-iteration quantity lives on the GPU for the whole call; the host calls through the ctypes
C-ABI of include/mmw_hip.h.  There is no CPU path: a missing library or GPU raises `MMWError`.

Extra keyword-only options (the reference has none of them).  `dtype`, `rng` and `expm_tol` left at None take their
defaults from the environment, so that a harness script that constructs `mmw(nit=150, eta=0.04)` unchanged
(sim_script/journal_version/sim_mmw_time.py:34, sim_script/pd_mmw_template.py:26) can be put on the fast path from outside:
  dtype   "f64" (default, the reference's arithmetic) or "f32"                       [$MMW_DTYPE]
  rng     "host": the K x D sketch of every iteration is drawn from the global NumPy stream exactly
          like mmw.py:226-227 and uploaded (a seeded run then consumes `np.random` like the
          reference does, including the one `standard_normal(K)` its `svds` start vector takes);
          "device": Philox sketches generated on the GPU (the fast path).              [$MMW_RNG]
  expm    "lanczos" (default) or "taylor"; expm_tol [$MMW_EXPM_TOL], expm_max_order tune the Krylov order choice.
  device  HIP device index (default: $LOCAL_RANK or 0).
  warm_start  False (default, the reference's behaviour: every run restarts from Y = 1/C, X = I, mmw.py:62-68) or True:
          a run on the state of the previous run continues from that run's (e_accu, L_accu, X, Y) and averages
          `warm_fraction * nit` iterations only -- the probes of one binary search solve neighbouring problems.  [$MMW_WARM_START]
"""
import math
import os

import numpy as np

from . import _lib
from .sdp_solver import sdp_solver
from .stats import STATS_OBJECT

_SKETCH_CHUNK_BYTES = 64 << 20


class mmw(STATS_OBJECT, sdp_solver):
    def __init__(self, nit=100, rank_radio=2, alpha=1., eta=0.1, log_gap=False, *, dtype=None, rng=None,
                 expm="lanczos", expm_tol=None, expm_max_order=12, device=None, seed=0, warm_start=None, warm_fraction=1.0 / 3.0):
        sdp_solver.__init__(self, nit=nit, rank_radio=rank_radio, alpha=alpha)
        self.eta = eta
        self.LOG_GAP = log_gap
        dtype = os.environ.get("MMW_DTYPE", "f64") if dtype is None else dtype
        rng = os.environ.get("MMW_RNG", "host") if rng is None else rng
        if expm_tol is None and os.environ.get("MMW_EXPM_TOL"):
            expm_tol = float(os.environ["MMW_EXPM_TOL"])
        if warm_start is None:
            warm_start = os.environ.get("MMW_WARM_START", "0") not in ("", "0")
        if dtype not in ("f32", "f64") or rng not in ("host", "device") or expm not in ("lanczos", "taylor"):
            raise ValueError("dtype in {f32,f64}, rng in {host,device}, expm in {lanczos,taylor}")
        self.dtype, self.rng, self.expm = dtype, rng, expm
        self._dtype_code = _lib.F32 if dtype == "f32" else _lib.F64
        self.expm_tol = expm_tol if expm_tol is not None else (1e-6 if dtype == "f32" else 1e-9)
        self.expm_max_order = expm_max_order
        self._device_index = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        self.seed = int(seed)
        self._runs = 0
        self.last_expm_info = None
        self.round_batch = rng == "device"  # the fast path batches the rounding attempts too
        self.warm_start = bool(warm_start)
        self.warm_fraction = float(warm_fraction)

    def sibling(self):
        """A second solver with the same settings and its own device handle, stream and sketch seeds (binary_search's opt-in
        speculative mode keeps one probe in flight on each)."""
        other = mmw(nit=self.nit, rank_radio=self.rank_radio, alpha=self.alpha, eta=self.eta, log_gap=self.LOG_GAP, dtype=self.dtype, rng=self.rng,
                    expm=self.expm, expm_tol=self.expm_tol, expm_max_order=self.expm_max_order, device=self._device_index, seed=self.seed + 7919,
                    warm_start=self.warm_start, warm_fraction=self.warm_fraction)
        other.round_batch = self.round_batch
        return other

    def prepare(self, Z, state):
        """State processing only (the device handle for `state`), so that a later run_with_state finds it there."""
        self._device_solver(Z, state, nit=int(self.nit), eta=self.eta, need_loop=True, warm=False)

    def run_with_state(self, bs_iteration, Z, state):
        tic = self._get_tic()
        ret = self._run(Z, state)
        tim = self._get_tim(tic)
        K = self._state_K(state)
        self._add_np_log("mmw_all_it", bs_iteration, np.array([Z, K, tim]))
        return ret

    def _run(self, Z, state):
        sp_tic = self._get_tic()
        K = self._state_K(state)
        nit = int(self.nit)
        warm = self.warm_start and self._same_state(state) and self._dev[2].iterations_done > 0
        if warm:  # continue from the previous probe's iterate and average fewer iterations
            nit = max(1, int(math.ceil(nit * self.warm_fraction)))
        solver = self._device_solver(Z, state, nit=nit, eta=self.eta, need_loop=True, warm=warm)
        self._add_np_log("mmw_iters", 0, np.array([Z, K, nit, 1.0 if warm else 0.0]))
        solver.set_expm(_lib.EXPM_LANCZOS if self.expm == "lanczos" else _lib.EXPM_TAYLOR, self.expm_max_order, self.expm_tol)
        # the reference's per-iteration phase timers (mmw.py:142,170,197,200).  The reference-exact path (host draws, one call per chunk of
        # sketches) times every iteration; the fast path one iteration of every MMW_PHASE_TIMING (default 8; 1: all, 0: none -- rows of zeros)
        stride = 1 if self.rng == "host" else int(os.environ.get("MMW_PHASE_TIMING", "8"))
        solver.set_timing(stride)
        D = solver.D
        self._add_np_log("mmw_state_process", 0, np.array([Z, K, self._get_tim(sp_tic)]))

        self._runs += 1
        dev_seed = (self.seed << 20) + self._runs
        per_call = 1 if self.LOG_GAP else nit
        if self.rng == "host":
            per_call = min(per_call, max(1, _SKETCH_CHUNK_BYTES // (K * D * 8)))
        done = 0
        self.N_STEP = 0
        while done < nit:
            n = min(per_call, nit - done)
            if self.LOG_GAP:
                self._add_np_log("gap", done, solver.gap())  # [max violation at Xbar, K lambda_min(L(Ybar)), gap], mmw.py:116
            if self.rng == "host":
                randv = np.random.randn(n * K, D) / math.sqrt(float(D))  # the draws of mmw.py:226, n iterations at once
                randv = randv / np.linalg.norm(randv, axis=1)[:, None]
                solver.iterate(n, randv)
            else:
                solver.iterate(n, None, seed=dev_seed)
            done += n
            self.N_STEP = done
        solver.sync()
        us = solver.read(_lib.F_PHASE_US).reshape(nit, 4) if stride > 0 else np.zeros((nit, 4))
        steps = np.arange(nit)
        zk = np.tile(np.array([Z, K], dtype=np.float64), (nit, 1))
        for key, col in (("mmw_dual", 0), ("mmw_loss", 1), ("mmw_expm", 2), ("mmw_per_it", 3)):
            self._add_np_log_rows(key, steps, np.hstack((zk, us[:, col:col + 1])))
        self.last_expm_info = solver.read(_lib.F_EXPM_INFO)

        tic_xavg = self._get_tic()
        rank = int(np.min([K - 1, (Z - 1) * self.rank_radio]))
        if self.rng == "host":
            np.random.standard_normal(K)  # the start vector scipy's svds draws at mmw.py:215 (keeps seeded streams aligned)
        # fast path: X_half stays on the device (a DeviceFactor: an array to everything but `rounding`, which reads it in place);
        # MMW_FACTOR_HOST=1 or rng="host" return the NumPy array itself
        resident = self.rng == "device" and os.environ.get("MMW_FACTOR_HOST", "0") in ("", "0")
        X_half = solver.factor(rank, seed=dev_seed, resident=resident)
        self._add_np_log("mmw_xavg", 0, np.array([Z, K, self._get_tim(tic_xavg)]))
        return True, X_half

    @staticmethod
    def expm_half_randsk(L, D):
        """The reference's seam (mmw.py:224-229): exp(L) applied to a row-normalised Gaussian sketch."""
        randv = np.random.randn(L.shape[0], D) / math.sqrt(float(D))
        randv = randv / np.linalg.norm(randv, axis=1)[:, None]
        out, _ = _lib.expm_apply(L, randv, dtype=_lib.F64, method=_lib.EXPM_LANCZOS, max_order=16, tol=1e-12,
                                 device=int(os.environ.get("LOCAL_RANK", "0")))
        return out
