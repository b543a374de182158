"""ctypes binding of include/mmw_hip.h (the only way the Python host reaches the GPU).

There is no CPU fallback: if `libmmw_hip.so` is missing or cannot be loaded this module raises, and
every entry point raises `MMWError` with the library's message on a non-zero status.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmw_hip.so")

F32, F64 = 0, 1
EXPM_LANCZOS, EXPM_TAYLOR = 0, 1

# enum mmw_field / mmw_ifield
F_Y, F_E_ACCU, F_E_THIS, F_LVAL, F_XVAL, F_XAVG, F_YAVG, F_XHALF, F_SKETCH = range(9)
F_S_SUM, F_NORM_H, F_ST_DATA, F_PHASE_US, F_EXPM_INFO, F_FACTOR, F_KERNEL_US, F_BLOCKING, F_SPMM_KIND, F_E_MAX, F_DUAL_INFO = range(9, 20)
KERNEL_CLASSES = ["spmm", "sddmm", "dual", "loss", "krylov_vec", "sketch", "project", "greedy", "factor"]
I_L_INDPTR, I_L_INDICES, I_ST_INDPTR, I_ST_INDICES, I_GAIN_X, I_GAIN_Y, I_ASSO_X, I_ASSO_Y, I_DIAG_POS, I_ASSO_POS = range(10)

EXPORTS = ["mmw_last_error", "mmw_version", "mmw_device_count", "mmw_create", "mmw_destroy", "mmw_sizes", "mmw_set_expm",
           "mmw_set_timing", "mmw_set_profile", "mmw_bench_spmm", "mmw_reset", "mmw_set_slots", "mmw_set_slots_warm", "mmw_set_eta", "mmw_iterate", "mmw_sync", "mmw_sketch", "mmw_read_f64", "mmw_read_i32", "mmw_gap",
           "mmw_factor", "mmw_expm_apply", "mmw_sym_eig", "mmw_round", "mmw_env_create", "mmw_env_destroy", "mmw_env_sizes", "mmw_env_state",
           "mmw_env_evaluate", "mmw_create_from_env", "mmw_env_bounds"]


class MMWError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library once; raise loudly if it is absent (build it with `python __graft_entry__.py`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MMWError("HIP library %s not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                       "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    p_i32 = C.POINTER(C.c_int32)
    p_f64 = C.POINTER(C.c_double)
    L.mmw_last_error.restype = C.c_char_p
    L.mmw_last_error.argtypes = []
    L.mmw_version.restype = C.c_int
    L.mmw_device_count.argtypes = [C.POINTER(C.c_int)]
    L.mmw_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                             p_i32, p_i32, p_f64, p_i32, p_i32, p_f64, p_f64]
    L.mmw_destroy.argtypes = [C.c_void_p]
    L.mmw_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.mmw_set_expm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    L.mmw_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.mmw_set_profile.argtypes = [C.c_void_p, C.c_int]
    L.mmw_bench_spmm.argtypes = [C.c_void_p, C.c_int, C.c_int, p_f64]
    L.mmw_reset.argtypes = [C.c_void_p, C.c_int32]
    L.mmw_set_slots.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.mmw_set_slots_warm.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.mmw_set_eta.argtypes = [C.c_void_p, C.c_double]
    L.mmw_iterate.argtypes = [C.c_void_p, C.c_int32, p_f64, C.c_uint64]
    L.mmw_sync.argtypes = [C.c_void_p]
    L.mmw_sketch.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, p_f64, C.c_int64]
    L.mmw_read_f64.argtypes = [C.c_void_p, C.c_int, p_f64, C.c_int64]
    L.mmw_read_i32.argtypes = [C.c_void_p, C.c_int, p_i32, C.c_int64]
    L.mmw_gap.argtypes = [C.c_void_p, p_f64]
    L.mmw_factor.argtypes = [C.c_void_p, C.c_int32, p_f64, C.c_uint64]
    L.mmw_sym_eig.argtypes = [C.c_int, C.c_int32, p_f64, C.c_double, C.c_int32, p_f64, p_f64, C.POINTER(C.c_int32)]
    L.mmw_expm_apply.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int32, C.c_int32, p_i32, p_i32, p_f64,
                                 p_f64, p_f64, p_f64, C.c_int32, p_f64]
    L.mmw_round.argtypes = [C.c_void_p, C.c_int32, C.c_int32, p_f64, C.c_int32, p_f64, p_i32, p_i32]
    L.mmw_env_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int32, C.c_int32, p_f64, p_f64, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_double]
    L.mmw_env_destroy.argtypes = [C.c_void_p]
    L.mmw_env_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.mmw_env_state.argtypes = [C.c_void_p, p_i32, p_i32, p_f64, p_i32, p_i32, p_f64, p_f64]
    L.mmw_env_evaluate.argtypes = [C.c_void_p, p_f64, C.c_int32, C.c_double, C.c_double, C.c_double, p_f64, p_f64]
    L.mmw_create_from_env.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_double, C.c_int32]
    L.mmw_env_bounds.argtypes = [C.c_void_p, p_i32]
    for name in EXPORTS:
        if name not in ("mmw_last_error",):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise MMWError("mmw_hip status %d: %s" % (rc, lib().mmw_last_error().decode("utf-8", "replace")))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int(0)
    check(lib().mmw_device_count(C.byref(n)))
    return n.value


def canonical_csr(m):
    """scipy CSR with sorted, de-duplicated indices as int32/float64 arrays (copies only when needed)."""
    import scipy.sparse
    m = scipy.sparse.csr_matrix(m)
    if not m.has_canonical_format:
        m = m.copy()
        m.sum_duplicates()
    return _i32(m.indptr), _i32(m.indices), _f64(m.data)


class Solver:
    """Owning wrapper of one `mmw_solver*` handle."""

    def __init__(self, Z, state, nit, eta, rank_radio=2, dtype=F64, device=0):
        S, Q, h = state
        self.K = int(S.shape[0])
        if S.shape != (self.K, self.K) or Q.shape != (self.K, self.K) or len(h) != self.K:
            raise MMWError("state must be (S_gain KxK, Q_asso KxK, h_max[K])")
        sp, si, sx = canonical_csr(S)
        qp, qi, qx = canonical_csr(Q)
        hm = _f64(h)
        self._h = C.c_void_p()
        check(lib().mmw_create(C.byref(self._h), int(device), int(dtype), self.K, int(Z), int(rank_radio), float(eta), int(nit),
                               _pi(sp), _pi(si), _pd(sx), _pi(qp), _pi(qi), _pd(qx), _pd(hm)))
        sz = (C.c_int64 * 10)()
        check(lib().mmw_sizes(self._h, sz))
        (self.K, self.Z, self.D, self.Dpad, self.nnzL, self.nnzST, self.E_gain, self.E_asso, self.C, _) = [int(x) for x in sz]
        self.dtype = dtype
        self._timing = False
        self._timed = 0

    @classmethod
    def from_env(cls, env, Z, nit, eta, rank_radio=2, dtype=F64):
        """The handle for the state a DeviceEnv holds, built on the device without the host round trip (mmw_create_from_env)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        check(lib().mmw_create_from_env(C.byref(self._h), env._h, int(dtype), int(Z), int(rank_radio), float(eta), int(nit)))
        sz = (C.c_int64 * 10)()
        check(lib().mmw_sizes(self._h, sz))
        (self.K, self.Z, self.D, self.Dpad, self.nnzL, self.nnzST, self.E_gain, self.E_asso, self.C, _) = [int(x) for x in sz]
        self.dtype = dtype
        self._timing = False
        self._timed = 0
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            try:
                self._keep_resident_factor()
            finally:
                lib().mmw_destroy(self._h)
                self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def iterations_done(self):
        sz = (C.c_int64 * 10)()
        check(lib().mmw_sizes(self._h, sz))
        return int(sz[9])

    def set_expm(self, method=EXPM_LANCZOS, max_order=12, tol=1e-9):
        check(lib().mmw_set_expm(self._h, int(method), int(max_order), float(tol)))

    def set_timing(self, on):
        """False / 0: off; True / 1: phase events in every iteration; S > 1: in one iteration of every S (rows repeated in between)."""
        check(lib().mmw_set_timing(self._h, int(on)))
        self._timing = bool(on)

    def set_profile(self, on):
        """False / 0: off; True / 1: synchronous mode with exact launch counts; 2: the shipped path as launched."""
        check(lib().mmw_set_profile(self._h, int(on)))

    def kernel_times(self):
        """{class: (total device us, launches)} since set_profile(True)."""
        self.sync()
        v = self.read(F_KERNEL_US, 2 * len(KERNEL_CLASSES))
        return {k: (float(v[2 * i]), int(v[2 * i + 1])) for i, k in enumerate(KERNEL_CLASSES)}

    def bench_spmm(self, blocked=True, reps=20):
        us = C.c_double(0.0)
        check(lib().mmw_bench_spmm(self._h, int(blocked), int(reps), C.byref(us)))
        return us.value

    def reset(self, nit):
        check(lib().mmw_reset(self._h, int(nit)))
        self._timed = 0

    def set_eta(self, eta):
        check(lib().mmw_set_eta(self._h, float(eta)))

    def spmm_kernel_info(self):
        """Which SpMM kernel exp(L/2)R runs on for this handle, and what bounds it (for the benchmark's roofline record)."""
        kind = int(self.read(F_SPMM_KIND, 2)[0])
        return [
            {"name": "k_spmm (generic CSR gather SpMM)", "limiter": "L2 gather rate of the dense rows"},
            {"name": "k_spmm_blk (LDS-staged locality-blocked CSR SpMM, 256-byte tiles)", "limiter": "VALU issue + LDS latency"},
            {"name": "k_spmm_blk2 (LDS-staged locality-blocked CSR SpMM, 128-byte half tiles)",
             "limiter": "VALU issue + LDS latency; operands live in L2 / Infinity Cache, not HBM"},
            {"name": "k_spmm_mfma (locality blocks as dense bf16 hi/lo -- first-order form: fp16 -- products on the matrix cores)",
             "limiter": "gather of the blocks' union rows into LDS at the CU's L2 rate; operands live in L2 / Infinity Cache, not HBM"},
        ][kind]

    def set_slots(self, Z, nit, warm=False):
        """Rebind to another slot count on the same state (keeps pattern, blocking and device copies).
        warm=True continues from the previous probe's iterate (mmw_set_slots_warm)."""
        check((lib().mmw_set_slots_warm if warm else lib().mmw_set_slots)(self._h, int(Z), int(nit)))
        sz = (C.c_int64 * 10)()
        check(lib().mmw_sizes(self._h, sz))
        (self.K, self.Z, self.D, self.Dpad, self.nnzL, self.nnzST, self.E_gain, self.E_asso, self.C, _) = [int(x) for x in sz]
        self._timed = 0

    def iterate(self, n, randv=None, seed=0):
        if self._timing:
            self._timed += int(n)
        if randv is None:
            check(lib().mmw_iterate(self._h, int(n), None, C.c_uint64(int(seed))))
        else:
            r = _f64(randv)
            if r.size != n * self.K * self.D:
                raise MMWError("randv must hold n*K*D = %d values, got %d" % (n * self.K * self.D, r.size))
            check(lib().mmw_iterate(self._h, int(n), _pd(r), C.c_uint64(0)))

    def sync(self):
        check(lib().mmw_sync(self._h))

    def sketch(self, seed, iteration):
        """The (K, D) sketch the device generator draws for `iteration` of a run with `seed` (counter-based: exact, any chunking)."""
        out = np.empty((self.K, self.D), dtype=np.float64)
        check(lib().mmw_sketch(self._h, C.c_uint64(int(seed)), int(iteration), _pd(out), int(out.size)))
        return out

    _LEN = {F_Y: "C", F_E_ACCU: "C", F_E_THIS: "C", F_LVAL: "nnzL", F_XVAL: "nnzL", F_XAVG: "nnzL", F_YAVG: "C",
            F_S_SUM: "K", F_NORM_H: "K", F_ST_DATA: "nnzST"}

    def read(self, which, n=None):
        if n is None:
            if which in (F_XHALF, F_SKETCH):
                n = self.K * self.D
            elif which in (F_EXPM_INFO, F_BLOCKING):
                n = 4
            elif which == F_SPMM_KIND:
                n = 2
            elif which == F_DUAL_INFO:
                n = 4
            elif which == F_E_MAX:
                n = 1
            elif which == F_PHASE_US:
                n = 4 * self._timed_iters()
            else:
                n = getattr(self, self._LEN[which])
        out = np.empty(int(n), dtype=np.float64)
        check(lib().mmw_read_f64(self._h, int(which), _pd(out), int(n)))
        if which in (F_XHALF, F_SKETCH):
            out = out.reshape(self.K, self.D)
        return out

    def _timed_iters(self):
        return self._timed

    _ILEN = {I_L_INDPTR: lambda s: s.K + 1, I_L_INDICES: lambda s: s.nnzL, I_ST_INDPTR: lambda s: s.K + 1,
             I_ST_INDICES: lambda s: s.nnzST, I_GAIN_X: lambda s: s.E_gain, I_GAIN_Y: lambda s: s.E_gain,
             I_ASSO_X: lambda s: s.E_asso, I_ASSO_Y: lambda s: s.E_asso, I_DIAG_POS: lambda s: s.K,
             I_ASSO_POS: lambda s: s.E_asso}

    def read_i32(self, which):
        n = self._ILEN[which](self)
        out = np.empty(int(n), dtype=np.int32)
        check(lib().mmw_read_i32(self._h, int(which), _pi(out), int(n)))
        return out

    def gap(self):
        out = np.empty(3, dtype=np.float64)
        check(lib().mmw_gap(self._h, _pd(out)))
        return out

    def factor(self, rank, seed=0, resident=False):
        """X_half (K, rank) float64.  resident=True: a `DeviceFactor` -- the factor stays on the device, `round` takes it from there,
        and it becomes a NumPy array (one copy out) the moment anything else looks at it."""
        self._keep_resident_factor()
        if resident:
            check(lib().mmw_factor(self._h, int(rank), None, C.c_uint64(int(seed))))
            self._factor_serial = getattr(self, "_factor_serial", 0) + 1
            df = DeviceFactor(self, int(rank), self._factor_serial)
            self._resident = weakref.ref(df)
            return df
        out = np.empty((self.K, int(rank)), dtype=np.float64)
        check(lib().mmw_factor(self._h, int(rank), _pd(out), C.c_uint64(int(seed))))
        self._factor_serial = getattr(self, "_factor_serial", 0) + 1
        return out

    def _keep_resident_factor(self):
        """A resident factor somebody still holds is copied out before the handle overwrites it (next factor) or goes away (close)."""
        ref = getattr(self, "_resident", None)
        df = ref() if ref is not None else None
        if df is not None and df._host is None and getattr(self, "_h", None) is not None and self._h.value:
            np.asarray(df)
        self._resident = None

    def round(self, Z, gX, randv):
        """randv: (nbatch, Z, D') row-normalised; returns (z[nbatch,K] int32 with -1 = unassigned, rem[nbatch])."""
        on_device = isinstance(gX, DeviceFactor) and gX.on_device_of(self)
        if not on_device:
            gX = _f64(gX)
        randv = _f64(randv)
        if randv.ndim == 2:
            randv = randv[None]
        nb, Zr, Dp = randv.shape
        if Zr != Z or tuple(gX.shape) != (self.K, Dp):
            raise MMWError("round: gX must be (K, D') and randv (nbatch, Z, D')")
        z = np.empty((nb, self.K), dtype=np.int32)
        rem = np.empty(nb, dtype=np.int32)
        check(lib().mmw_round(self._h, int(Z), int(Dp), None if on_device else _pd(gX), int(nb), _pd(randv), _pi(z), _pi(rem)))
        return z, rem


class DeviceFactor(np.lib.mixins.NDArrayOperatorsMixin):
    """The X_half of `Solver.factor(resident=True)`: (K, rank) float64 that lives on the device.  The reference hands the factor from
    `run_with_state` straight to `rounding` (binary_search_relaxation.py:50-53): `Solver.round` recognises this object and reads the factor
    where it lies.  For everything else it is an array: `np.asarray`, arithmetic, indexing and attribute access copy it out (once)."""

    def __init__(self, solver, rank, serial):
        self._solver, self._serial, self._host = solver, serial, None
        self.shape, self.ndim, self.dtype = (solver.K, rank), 2, np.dtype(np.float64)

    def on_device_of(self, solver):
        """True while `solver` is the handle that made this factor and has not made another one since."""
        return self._solver is solver and getattr(solver, "_factor_serial", 0) == self._serial and bool(solver._h.value)

    def __array__(self, dtype=None, copy=None):
        if self._host is None:
            if not self.on_device_of(self._solver):
                raise MMWError("this factor was not copied to the host before its handle computed another one (or was closed)")
            self._host = self._solver.read(F_FACTOR, self.shape[0] * self.shape[1]).reshape(self.shape)
        return self._host if dtype is None else self._host.astype(dtype, copy=False)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        return getattr(ufunc, method)(*[np.asarray(x) if isinstance(x, DeviceFactor) else x for x in inputs], **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        conv = lambda x: np.asarray(x) if isinstance(x, DeviceFactor) else x
        return func(*[conv(a) for a in args], **{k: conv(v) for k, v in kwargs.items()})

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return np.asarray(self)[idx]

    def __getattr__(self, name):  # (only reached for what the object itself lacks: T, sum, copy, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(np.asarray(self), name)


class DeviceEnv:
    """Owning wrapper of one `mmw_env*`: the reference's problem generator and scorer on the device (env.py:136-233)."""

    def __init__(self, sta_locs, ap_locs, fre_Hz=4e9, txp_offset=2.0, min_s_n_ratio=0.1, min_sinr=1.0, noise_floor_dbm=-94.0, device=0):
        sta = _f64(sta_locs)
        ap = _f64(ap_locs)
        if sta.ndim != 2 or sta.shape[1] != 2 or ap.ndim != 2 or ap.shape[1] != 2:
            raise MMWError("station and AP locations must be (n, 2) arrays")
        self.K, self.A = int(sta.shape[0]), int(ap.shape[0])
        self._h = C.c_void_p()
        check(lib().mmw_env_create(C.byref(self._h), int(device), self.K, self.A, _pd(sta), _pd(ap), float(fre_Hz), float(txp_offset),
                                   float(min_s_n_ratio), float(min_sinr), float(noise_floor_dbm)))
        sz = (C.c_int64 * 4)()
        check(lib().mmw_env_sizes(self._h, sz))
        self.nnzS, self.nnzQ = int(sz[2]), int(sz[3])

    def state(self):
        """(S_gain csr, Q_asso csr, h_max) as env.generate_S_Q_hmax returns them."""
        import scipy.sparse
        sp = np.empty(self.K + 1, dtype=np.int32); si = np.empty(self.nnzS, dtype=np.int32); sx = np.empty(self.nnzS, dtype=np.float64)
        qp = np.empty(self.K + 1, dtype=np.int32); qi = np.empty(self.nnzQ, dtype=np.int32); qx = np.empty(self.nnzQ, dtype=np.float64)
        h = np.empty(self.K, dtype=np.float64)
        check(lib().mmw_env_state(self._h, _pi(sp), _pi(si), _pd(sx), _pi(qp), _pi(qi), _pd(qx), _pd(h)))
        S = scipy.sparse.csr_matrix((sx, si, sp), shape=(self.K, self.K))
        Q = scipy.sparse.csr_matrix((qx, qi, qp), shape=(self.K, self.K))
        return S, Q, h

    def bounds(self):
        """(lower, upper) slot-count bounds of binary_search_relaxation.py:13-29 for this state, from the device's count pass."""
        out = np.zeros(2, dtype=np.int32)
        check(lib().mmw_env_bounds(self._h, _pi(out)))
        return int(out[0]), int(out[1])

    def device_state(self):
        """The `state` to hand to the solver classes: behaves like the (S_gain, Q_asso, h_max) tuple (materialised on the host only if
        someone indexes it), and lets `mmw` / `binary_search_relaxation` stay on the device (Solver.from_env, bounds())."""
        return DeviceState(self)

    def evaluate(self, z, Z, packet_bit=800, bandwidth=5e6, slot_time=1.25e-4, bler=True):
        """(sinr, bler) per user under the colouring z (env.evaluate_sinr / evaluate_bler)."""
        zz = _f64(z)
        if zz.shape != (self.K,):
            raise MMWError("z must have one slot per user")
        sinr = np.empty(self.K, dtype=np.float64)
        bl = np.empty(self.K, dtype=np.float64) if bler else None
        check(lib().mmw_env_evaluate(self._h, _pd(zz), int(Z), float(packet_bit), float(bandwidth), float(slot_time), _pd(sinr),
                                     _pd(bl) if bler else None))
        return sinr, bl

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().mmw_env_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceState:
    """`state` of a DeviceEnv: a lazy (S_gain, Q_asso, h_max) triple that remembers where it lives."""

    def __init__(self, env):
        self.env = env
        self.K = env.K
        self._host = None

    def host(self):
        if self._host is None:
            self._host = self.env.state()
        return self._host

    def __getitem__(self, i):
        return self.host()[i]

    def __iter__(self):
        return iter(self.host())

    def __len__(self):
        return 3


def sym_eig(G, rel_tol=1e-13, max_sweeps=30, device=0):
    """Eigen-decomposition of a symmetric matrix by the device's block Jacobi; returns (theta unsorted, Q, block sweeps)."""
    G = _f64(G)
    b = G.shape[0]
    if G.shape != (b, b):
        raise ValueError("sym_eig: square matrix expected")
    theta = np.empty(b, dtype=np.float64)
    Q = np.empty((b, b), dtype=np.float64)
    sw = C.c_int32(0)
    check(lib().mmw_sym_eig(int(device), b, _pd(G), float(rel_tol), int(max_sweeps), _pd(theta), _pd(Q), C.byref(sw)))
    return theta, Q, sw.value


def expm_apply(A_csr, B, dtype=F64, method=EXPM_LANCZOS, max_order=12, tol=1e-9, device=0, reps=1):
    """exp(A) B on the device for a symmetric scipy CSR matrix A; returns (out, info dict)."""
    ip, ci, vv = canonical_csr(A_csr)
    B = _f64(B)
    K, D = B.shape
    out = np.empty((K, D), dtype=np.float64)
    info = np.zeros(4, dtype=np.float64)
    us = C.c_double(0.0)
    check(lib().mmw_expm_apply(int(device), int(dtype), int(method), int(max_order), float(tol), K, D, _pi(ip), _pi(ci), _pd(vv),
                               _pd(B), _pd(out), _pd(info), int(reps), C.byref(us)))
    return out, {"one_norm": info[0], "order": int(info[1]), "substeps": int(info[2]), "shift": info[3], "kernel_us": us.value}
