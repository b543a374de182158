/*
 * mmw_hip.h -- C ABI of the MI355X (gfx950) MMW SDP hot path.
 *
 * The reference (zhouyou-gu/sig-sdp-mmw) is pure Python and has no FFI; its boundary for this path
 * is the duck-typed solver protocol
 *
 *     f, gX          = alg.run_with_state(it, Z, state)   sim_src/alg/binary_search_relaxation.py:50
 *     z_vec, Z, rem  = alg.rounding(Z, gX, state)         sim_src/alg/binary_search_relaxation.py:53
 *
 * implemented by class mmw (sim_src/alg/mmw.py:12-229) and sdp_solver.rounding
 * (sim_src/alg/sdp_solver.py:18-107).  The entry points below are what a ctypes binding of that class
 * calls (see INTEGRATION.md for the stub); each one cites the reference lines it replaces.
 *
 * Conventions: C linkage, plain pointers + sizes, no exceptions cross the ABI.  Every function returns
 * 0 on success or a negative status; mmw_last_error() gives the message of the calling thread's last
 * failure.  Host buffers are caller-owned (NumPy arrays); device memory is library-owned.  One handle =
 * one device + one HIP stream.  A handle is not thread-safe; different handles may be used
 * concurrently (the library keeps no mutable global state besides the thread-local error string).
 * All host-side floating point crosses the ABI as float64 whatever the device compute type is.
 */
#ifndef MMW_HIP_H
#define MMW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmw_solver mmw_solver;

#define MMW_OK 0
#define MMW_ERR_ARG (-1)     /* bad argument / malformed state */
#define MMW_ERR_HIP (-2)     /* a HIP runtime call failed (no device, OOM, launch failure) */
#define MMW_ERR_STATE (-3)   /* call out of order (e.g. iterate past nit) */

enum mmw_dtype { MMW_F32 = 0, MMW_F64 = 1 };
enum mmw_expm_method {
    MMW_EXPM_LANCZOS = 0, /* per-column Lanczos, exp of the small tridiagonals on device */
    MMW_EXPM_TAYLOR = 1   /* shifted truncated Taylor (the scheme scipy's expm_multiply runs) */
};

/* selectors for mmw_read_f64 / mmw_read_i32 */
enum mmw_field {
    MMW_F_Y = 0,          /* [C]      dual weights Y (mmw.py:139)                       */
    MMW_F_E_ACCU = 1,     /* [C]      accumulated violations e_accu (mmw.py:137)        */
    MMW_F_E_THIS = 2,     /* [C]      last e_this (mmw.py:136)                          */
    MMW_F_LVAL = 3,       /* [nnzL]   L_accu on the fixed pattern (mmw.py:167)          */
    MMW_F_XVAL = 4,       /* [nnzL]   X on the pattern, diagonal included (mmw.py:183-194) */
    MMW_F_XAVG = 5,       /* [nnzL]   running sum of X (mmw.py:77), not yet divided     */
    MMW_F_YAVG = 6,       /* [C]      running sum of Y (mmw.py:78)                      */
    MMW_F_XHALF = 7,      /* [K*D]    last exp(L/2) R, row-major (mmw.py:180)           */
    MMW_F_SKETCH = 8,     /* [K*D]    last row-normalised sketch R (mmw.py:226-227)     */
    MMW_F_S_SUM = 9,      /* [K]      S_sum  (mmw.py:34)                                */
    MMW_F_NORM_H = 10,    /* [K]      norm_H (mmw.py:39)                                */
    MMW_F_ST_DATA = 11,   /* [nnzST]  values of S_T' in CSR order (mmw.py:28-33)        */
    MMW_F_PHASE_US = 12,  /* [4*iters] per-iteration device us: dual, loss, expm, total (mmw.py:141,169,196,199) */
    MMW_F_EXPM_INFO = 13, /* [4]      last plan: one-norm bound, Krylov order m, substeps, shift mu */
    MMW_F_FACTOR = 14,    /* [K*rank] last factor of the averaged X (mmw.py:213-216)    */
    MMW_F_BLOCKING = 16,  /* [4]      locality blocking: in use (0/1), row blocks, nnz per staged row (reuse); [3] = batches replayed
                             because the device-side Krylov order outgrew the launched stages */
    MMW_F_SPMM_KIND = 17, /* [2]      SpMM kernel of exp(L/2)R on this handle: 0 generic CSR gather, 1 LDS-staged full tiles, 2 half tiles,
                             3 matrix-core (bf16 hi/lo split); [1] = 1 while the last plan allowed the matrix-core kernel */
    MMW_F_E_MAX = 18,     /* [1]      max_c e_c(X) of the last iteration (the largest entry of MMW_F_E_THIS, reduced on the device) */
    MMW_F_DUAL_INFO = 19, /* [4]      iterations since mmw_create {whose DUAL phase took the row sums of X from the matrix-core SDDMM instead of a
                             pass of its own, whose softmax ran inside the violation pass, whose exponential was ONE first-order product,
                             ... with the matrix in one fp16 half} */
    MMW_F_KERNEL_US = 15  /* [2*9]    per kernel class {total device us, launches} since mmw_set_profile(1):
                             spmm, sddmm, dual, loss, krylov vector ops, sketch, projection, greedy, factor */
};
enum mmw_ifield {
    MMW_I_L_INDPTR = 0,   /* [K+1]   */
    MMW_I_L_INDICES = 1,  /* [nnzL]  */
    MMW_I_ST_INDPTR = 2,  /* [K+1]   */
    MMW_I_ST_INDICES = 3, /* [nnzST] */
    MMW_I_GAIN_X = 4,     /* [E_gain] nz_idx_gain_x_ut (mmw.py:56) */
    MMW_I_GAIN_Y = 5,
    MMW_I_ASSO_X = 6,     /* [E_asso] nz_idx_asso_x_ut (mmw.py:57) */
    MMW_I_ASSO_Y = 7,
    MMW_I_DIAG_POS = 8,   /* [K] position of (k,k) in the L pattern */
    MMW_I_ASSO_POS = 9    /* [E_asso] position of (x,y), x<y, in the L pattern */
};

const char* mmw_last_error(void);
int mmw_version(void);
/* number of visible HIP devices; does not create a context */
int mmw_device_count(int* n);

/*
 * mmw_create: mmw._process_state + the prologue of mmw._run (mmw.py:26-41, 46-74).
 * Takes `state` exactly as the reference's caller hands it over: S_gain and Q_asso as canonical CSR
 * (sorted indices, no duplicates; int32 index arrays as scipy stores them), h_max[K].  Builds S_T',
 * S_sum, norm_H, the edge lists and the fixed CSR pattern of L/X in native host code, copies them to
 * `device` once and sets the iterate to the reference's initial point (Y = 1/C, X = I, L = 0).
 * The inputs are not modified (the reference copies them too, mmw.py:28).
 * device == -1 builds the host-side pattern only (no HIP call): such a handle answers mmw_sizes,
 * mmw_read_i32 and the host fields S_SUM / NORM_H / ST_DATA, everything else returns MMW_ERR_STATE.
 */
int mmw_create(mmw_solver** out, int device, int dtype, int32_t K, int32_t Z, int32_t rank_radio, double eta,
               int32_t nit, const int32_t* S_indptr, const int32_t* S_indices, const double* S_data,
               const int32_t* Q_indptr, const int32_t* Q_indices, const double* Q_data, const double* h_max);
int mmw_destroy(mmw_solver* s);

/* out[0..9] = K, Z, D, Dpad, nnzL, nnzST, E_gain, E_asso, C, iterations done */
int mmw_sizes(mmw_solver* s, int64_t out[10]);

/* Krylov scheme for exp(L/2)R, max order per substep (<= 16) and target relative accuracy. */
int mmw_set_expm(mmw_solver* s, int method, int max_order, double tol);
/* 1: record HIP events around every phase of every iteration (fills MMW_F_PHASE_US, the reference's per-iteration timers
 * mmw.py:142,170,197,200); 0: none, iterations run back to back; S > 1: events only in iteration 0 and in one iteration of every S
 * (four event records per iteration are four barrier packets in the chain of dependent launches): MMW_F_PHASE_US still has one row per
 * iteration, the rows of a group of S iterations repeat the group's sample -- the harness takes means (sim_mmw_time.py:48-52). */
int mmw_set_timing(mmw_solver* s, int enabled);

/* 1: bracket every kernel class with HIP events on the solver's stream (fills MMW_F_KERNEL_US), plans read back every iteration and
 * every class in launches of its own, so launch counts are exact; 2: the same brackets around the launches of the shipped path as they
 * are (chunks without readback, the sketch and the lagged plan riding in the LOSS launch); 0: off.  Clears the sums. */
int mmw_set_profile(mmw_solver* s, int enabled);

/* micro-benchmark of the dominant kernel on the handle's pattern and current L values: `reps` launches of the
 * CSR SpMM (blocked = 1: LDS-staged locality-blocked kernel, 0: generic gather kernel); mean device us per launch */
int mmw_bench_spmm(mmw_solver* s, int blocked, int reps, double* avg_us);

/* back to the initial point of mmw.py:62-73 for a fresh run of `nit` iterations on the same (state, Z) */
int mmw_reset(mmw_solver* s, int32_t nit);

/*
 * mmw_set_slots: rebind the handle to another slot count Z on the SAME state (the binary search probes
 * several Z per state, binary_search_relaxation.py:46-50): norm_H and the D = Z*rank_radio wide blocks are
 * rebuilt, the pattern, its locality blocking and the device copies of the state are reused; then mmw_reset(nit).
 */
int mmw_set_slots(mmw_solver* s, int32_t Z, int32_t nit);
/*
 * mmw_set_slots_warm: the same rebinding, but the next run CONTINUES from the previous probe's iterate instead of the
 * reference's initial point (opt-in warm start of the binary search, binary_search_relaxation.py:44-72 calls the solver
 * once per probed Z and the reference restarts each time, mmw.py:62-68): e_accu, L_accu and the last X / Y are kept, the
 * running sums of X and Y restart from them.  Falls back to mmw_set_slots when the handle has not iterated yet.
 */
int mmw_set_slots_warm(mmw_solver* s, int32_t Z, int32_t nit);
/* step size for the iterations that follow (the reference reads self.eta on every run, mmw.py:137,167) */
int mmw_set_eta(mmw_solver* s, double eta);

/*
 * mmw_iterate: `n` passes of the loop body mmw.py:75-200 (averaging, DUAL, LOSS, EXPM), device resident.
 * randv: NULL -> the sketch of every iteration is generated on the device (Philox4x32-10 normals,
 * counter = (seed, iteration, row, column), rows normalised); otherwise n*K*D float64, iteration-major,
 * each block the row-normalised (K,D) sketch the reference would draw at mmw.py:226-227 (parity mode).
 * Returns after the work is enqueued; any read or mmw_sync waits for it.
 */
int mmw_iterate(mmw_solver* s, int32_t n, const double* randv, uint64_t seed);
int mmw_sync(mmw_solver* s);
/*
 * mmw_sketch: the row-normalised (K, D) sketch the device generator draws for `iteration` of a run with `seed` (what
 * np.random.randn + the row normalisation of mmw.py:226-227 are to the reference).  The generator is counter-based (Philox4x32-10
 * keyed by seed, counter = iteration, row, column), so the block is exactly the one mmw_iterate(n, NULL, seed) multiplied in that
 * iteration, in whatever chunk it ran: parity tests hand it to the oracle to follow a device-RNG run.  out: K*D float64, row-major.
 * Waits for enqueued work; does not touch the iterate.
 */
int mmw_sketch(mmw_solver* s, uint64_t seed, int32_t iteration, double* out, int64_t n);

int mmw_read_f64(mmw_solver* s, int which, double* out, int64_t n);
int mmw_read_i32(mmw_solver* s, int which, int32_t* out, int64_t n);

/*
 * mmw_gap: the LOG_GAP branch mmw.py:79-117 at the current averages (call before iteration i with
 * i+1 terms accumulated): out = { max_c e_c(Xbar), K*lambda_min(L(Ybar)), difference }.
 */
int mmw_gap(mmw_solver* s, double out[3]);

/*
 * mmw_factor: the epilogue mmw.py:202-216.  Xbar = (sum of X)/nit on the pattern, top-`rank`
 * (by |eigenvalue|) invariant subspace by block Krylov iteration on the device,
 * X_half[K,rank] = V sqrt(|lambda|), columns in ascending |lambda| like svds.  out: K*rank float64, or NULL: the factor stays on
 * the device (mmw_round takes it from there with gX == NULL; mmw_read_f64(MMW_F_FACTOR) copies it out when asked) -- the caller
 * binary_search_relaxation.py:50-53 only hands it from run_with_state to rounding.
 */
int mmw_factor(mmw_solver* s, int32_t rank, double* out, uint64_t seed);

/*
 * mmw_expm_apply: the stand-alone seam mmw.expm_half_randsk (mmw.py:224-229) minus the draw:
 * out[K,D] = exp(A) B for a symmetric CSR matrix A (pattern arbitrary) and a dense block B.
 * info (may be NULL): one-norm bound, order m, substeps, shift.  `reps` > 1 repeats the device work
 * (benchmarking); kernel_us (may be NULL) receives the mean device time of one application.
 */
int mmw_expm_apply(int device, int dtype, int method, int max_order, double tol, int32_t K, int32_t D,
                   const int32_t* indptr, const int32_t* indices, const double* data, const double* B,
                   double* out, double info[4], int32_t reps, double* kernel_us);

/*
 * mmw_sym_eig: the dense symmetric eigensolve inside mmw_factor's Rayleigh-Ritz step, stand-alone (the reference gets it from
 * LAPACK inside scipy.sparse.linalg.eigsh / svds, mmw.py:206-212): G = Q diag(theta) Q^T for a symmetric b x b matrix (row-major
 * float64) by block Jacobi on the device.  Stops when the off-diagonal Frobenius norm is below rel_tol * max|diag| * sqrt(b)
 * or after max_sweeps block sweeps (*sweeps, may be NULL, receives the count).  theta is not sorted.
 */
int mmw_sym_eig(int device, int32_t b, const double* G, double rel_tol, int32_t max_sweeps, double* theta, double* Q, int32_t* sweeps);

/*
 * mmw_round: one sdp_solver.rounding_one_attempt (sdp_solver.py:27-107) per projection batch entry.
 * gX[K,Dp] and randv[nbatch,Z,Dp] (row-normalised, sdp_solver.py:48-49) in float64.  For every batch
 * entry: inprod = randv gX^T on the fp64 matrix cores, per-user slot preference order, the greedy
 * feasibility assignment in descending ||gX_k|| order.  z_out[nbatch,K] gets the slot or -1 for a user
 * left unassigned (the caller draws those, sdp_solver.py:104-105); rem_out[nbatch] the count.
 * gX == NULL: the K x Dp factor mmw_factor computed last on this handle, read where it lies on the device.
 */
int mmw_round(mmw_solver* s, int32_t Zr, int32_t Dp, const double* gX, int32_t nbatch, const double* randv,
              int32_t* z_out, int32_t* rem_out);

/*
 * The producer and the consumer either side of the path, on the device.
 *
 * mmw_env_create: env._compute_txp / _compute_state / generate_S_Q_hmax (sim_src/env/env.py:136-196) for K stations at
 * sta_xy[K][2] and A access points at ap_xy[A][2] (the caller draws the station drop, env.py:59): path loss
 * 20 log10(f/1e6) - 12 + 28 log10(d + 1) dB, transmit power so that the strongest AP receives txp_offset * min_sinr over the
 * noise floor, receive powers below min_s_n_ratio dropped, association = strongest AP; S_gain = rx[:, asso] without explicit
 * zeros, Q_asso = same-AP relation without diagonal, h_max = diag(S_gain) / min_sinr - 1, all as CSR built on the device.
 * mmw_env_sizes: out = {K, A, nnz(S_gain), nnz(Q_asso)}; mmw_env_state copies the CSR arrays (caller-sized from mmw_env_sizes)
 * and h_max[K] to the host -- exactly the `state` mmw_create takes.
 * mmw_env_evaluate: env.evaluate_sinr / evaluate_bler (env.py:198-233) for the colouring z_vec[K] (float64 slot numbers as
 * rounding returns them) with Z slots: sinr_out[K] after the one-survivor rule for users of one AP sharing a slot, and, when
 * bler_out is not NULL, the finite-blocklength block error rate (env.py:107-111) per user.
 */
typedef struct mmw_env mmw_env;
int mmw_env_create(mmw_env** out, int device, int32_t K, int32_t A, const double* sta_xy, const double* ap_xy, double fre_Hz,
                   double txp_offset, double min_s_n_ratio, double min_sinr, double noise_floor_dbm);
int mmw_env_destroy(mmw_env* e);
int mmw_env_sizes(mmw_env* e, int64_t out[4]);
int mmw_env_state(mmw_env* e, int32_t* S_indptr, int32_t* S_indices, double* S_data, int32_t* Q_indptr, int32_t* Q_indices,
                  double* Q_data, double* h_max);
int mmw_env_evaluate(mmw_env* e, const double* z_vec, int32_t Z, double packet_bit, double bandwidth, double slot_time,
                     double* sinr_out, double* bler_out);
/*
 * mmw_create_from_env: mmw_create for the state this generator holds, WITHOUT the host round trip (env.generate_S_Q_hmax ->
 * mmw._process_state, sim_src/env/env.py:168-196 -> sim_src/alg/mmw.py:26-57): the transpose S_T', the association mask, the symmetric
 * pattern of L / X with its per-entry weights, mirrors and pair ids, the edge lists and the row statistics are built by device
 * kernels straight from the generator's receive powers (csrc/pattern_device.h; rows come out sorted, no sort and no atomic), and the
 * rounding's view of the state from its CSR of S_gain.  The handle is the one mmw_create makes from mmw_env_state's arrays: same
 * pattern, same lists (every mmw_read_i32 field equal), S_sum / norm_H / ST_DATA bit-identical.  Only the traversal order of the
 * locality blocking differs: the generator knows the stations' coordinates, so the row blocks are runs of a spatial order instead
 * of patches grown along an RCM order of the pattern (products agree to rounding, not bit for bit; MMW_ENV_RCM=1 forces the latter).
 * The handle does not keep a reference to `env` after the call returns.
 * mmw_env_bounds: the bisection's bounds of binary_search_relaxation.py:13-29 for this state, out = {lower, upper}, from the same
 * count pass (no host matrices).
 */
int mmw_create_from_env(mmw_solver** out, mmw_env* env, int dtype, int32_t Z, int32_t rank_radio, double eta, int32_t nit);
int mmw_env_bounds(mmw_env* e, int32_t out[2]);

#ifdef __cplusplus
}
#endif
#endif /* MMW_HIP_H */
