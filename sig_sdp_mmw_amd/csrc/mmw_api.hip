// C-ABI of the MI355X MMW hot path (include/mmw_hip.h) and the device-resident solver behind it.
#include <chrono>
#include <cstring>
#include <map>
#include <queue>
#include <memory>
#include <atomic>
#include <thread>

#include "blocking.h"
#include "env_device.h"
#include "expm_engine.h"
#include "kernels_loop.h"
#include "pattern.h"
#include "pattern_device.h"
#include "runtime.h"
#include "solver_extras.h"

using namespace mmw;

struct mmw_env {
    mmw::EnvDevice e;
};
struct mmw_solver {
    virtual ~mmw_solver() {}
    virtual int sizes(int64_t out[10]) = 0;
    virtual int set_expm(int method, int max_order, double tol) = 0;
    virtual int set_timing(int enabled) = 0;
    virtual int set_profile(int enabled) = 0;
    virtual int bench_spmm(int blocked, int reps, double* avg_us) = 0;
    virtual int reset(int32_t nit) = 0;
    virtual int set_slots(int32_t Z, int32_t nit, int warm) = 0;
    virtual int set_eta(double eta) = 0;
    virtual int iterate(int32_t n, const double* randv, uint64_t seed) = 0;
    virtual int sync() = 0;
    virtual int sketch(uint64_t seed, int32_t iteration, double* out, int64_t n) = 0;
    virtual int read_f64(int which, double* out, int64_t n) = 0;
    virtual int read_i32(int which, int32_t* out, int64_t n) = 0;
    virtual int gap(double out[3]) = 0;
    virtual int factor(int32_t rank, double* out, uint64_t seed) = 0;
    virtual int round(int32_t Zr, int32_t Dp, const double* gX, int32_t nbatch, const double* randv, int32_t* z_out,
                      int32_t* rem_out) = 0;
};

namespace {

template <typename T> BlockingLimits blocking_limits() { return BlockingLimits{blk_max_entries<T>(), (int)sizeof(BlkMeta<T>)}; }

template <typename T> struct Solver final : mmw_solver {
    int device = 0;
    bool host_only = false;
    hipStream_t st = nullptr;
    HostPattern H;
    int K = 0, Z = 0, D = 0, rank_radio = 2, nit = 0, iter = 0;
    double eta = 0.1;
    bool timing = false;
    bool kt_shipped = false;  // set_profile(2)
    bool kt_exact() const { return kt.on && !kt_shipped; }  // set_profile(1): synchronous plans, every kernel class in launches of its own
    // pattern on the device
    DevBuf<int> d_indptr, d_col, d_pid, d_mirror, d_diag, d_apos, d_lrow;
    DevBuf<T> d_sab, d_sba, d_h, d_ssum, d_invn, d_cH;
    // iterate state
    DevBuf<T> lval, xval, xavg, Y, yavg, e_accu, e_this, rsum, Xh, drow;
    DevBuf<double> max_part, sum_part, scal, tr_part, stage64, out64;
    DevBuf<T> wH;  // Y_H / norm_H
    DevBuf<T> yun;  // the fused DUAL pass's unnormalised exponentials (see iterate_impl)
    const bool fuse_dual = getenv("MMW_NO_FUSED_DUAL") == nullptr;
    // how far e_accu's maximum may run ahead of the fused pass's shift before its exponentials are distrusted (exp overflows T
    // near 88 / 709); MMW_DUAL_GAP is for the tests, which force the replay with it
    const double dual_gap = getenv("MMW_DUAL_GAP") ? atof(getenv("MMW_DUAL_GAP")) : (sizeof(T) == 4 ? 60.0 : 600.0);
    static constexpr int LOSS_GRID_MAX = 4096;
    // locality blocking (blocking.h)
    HostBlocking HB;
    DevBuf<int> b_rowptr, b_order, b_unptr, b_uncols, b_bptr, b_bpos, b_bepos;
    DevBuf<unsigned short> b_lidx, b_selfli, b_sdla, b_sdlb;
    DevBuf<int> b_sdptr, b_sdepos, b_desc, b_unfixed;
    bool sddmm_blk = false;
    DevBuf<int> b_sd2ptr, b_sd2epos, b_sd2items;
    int sd2_nitems = 0;
    DevBuf<unsigned> b_sd2ab;
    bool sddmm_blk2 = false;  // half-tile SDDMM (k_sddmm_blk2)
    int64_t sketch_done_for = -1;  // iteration whose sketch the last SDDMM launch already drew into the start block
    uint64_t sketch_done_seed = 0;
    int sketch_done_slabs = 0;
    DevBuf<T> lval_blk;
    bool lblk_stale = false;         // lval_blk lags lval (the matrix-core kernel ran the last products)
    bool lagged_plan = getenv("MMW_NO_LAGGED_PLAN") == nullptr;
    bool lagged_missed = false;  // an extrapolated plan of this run did not cover its matrix: the run's matrix outgrows the extrapolation, exact plans until the next reset
    DevBuf<ExpmPlan> sn_plan;        // plan (with its history) at the start of the pending chunk
    DevBuf<int> b_kbase, b_fpos, b_mdesc, b_munfixed, b_morder;  // matrix-core SpMM: its row blocks, CSR entry -> fragment image position
    DevBuf<unsigned> afrag;          // the matrix as bf16 hi << 16 | lo words in MFMA fragment order
    DevBuf<int> b_tbase, b_tptr;  // matrix-core SDDMM: pattern entries by 32 x 32 output tile
    // X in the matrix-core SDDMM's tile order (kernels_mfma.h): xs_val / xs_avg hold X and its running sum while x_tiles is set, the
    // CSR-ordered xval / xavg otherwise; b_e2w maps a CSR entry to its slot, b_xasso an association pair
    DevBuf<T> xs_val, xs_avg;
    DevBuf<int> b_e2w, b_xasso;
    size_t n_xs = 0;        // slots: undirected edges + K
    bool x_tiles = false;   // which pair of buffers holds the iterate's X
    bool sn_tiles = false;  // ... and which the pending chunk's snapshot was taken from
    DevBuf<unsigned short> b_trc, b_tmask, xh_planes;
    DevBuf<long long> rsfx;  // [2K] 2^-40 fixed-point totals: [0, K) row sums of the off-diagonal X, left by the matrix-core SDDMM; [K, 2K) row norms of
                             // y = exp(L/2)R from the first-order product (kernels_mfma.h).  Zeroed by every LOSS pass.
    DevBuf<double> tr1_part; // trace shares of the first-order product's workgroups (zero where none works)
    const bool first_enabled = getenv("MMW_NO_FIRST_ORDER") == nullptr;
    const bool first_a16_enabled = getenv("MMW_NO_FIRST_A16") == nullptr;
    const double fv_du_scale = getenv("MMW_FV_DU_SCALE") ? atof(getenv("MMW_FV_DU_SCALE")) : 1.0;  // tests: inflates the measured rounding of the fp16 plane (forced miss)
    // the rounding of the first-order product's fp16 plane: measured by the sketch kernel (default), or the format's worst case
    const bool fv_measure = getenv("MMW_FV_WORSTCASE") == nullptr;
    double plane_rounding() const { return fv_measure ? F16_PLANE_EXPECT : 1.02 * F16_UNIT; }
    DevBuf<unsigned short> afrag16;  // the matrix as ONE fp16 half, for the first-order product while 2 * 2^-12 absn <= tol (holes zero; an image of its own)
    bool first_a16_guess = false;    // the chunk being enqueued takes that form
    long long n_first16_iters = 0;
    bool first_guess = false;  // the chunk being enqueued takes the first-order exponential (first_order_ok at its start)
    int age0 = 0;              // iterations L_accu had accumulated when this run started (a warm restart continues it): the matrix's norm and
                               // every estimate derived from it grow with age() = age0 + iter, not with the run's own counter
    int age() const { return age0 + iter; }
    // Growth of the matrix's norm bound over the coming `ahead` iterations, as a ratio: at least linear in the age, and at least what
    // the last two plans read back in this run showed (after a warm restart with fewer slots the violations -- and with them the
    // increments of L -- are larger than the age suggests), with a factor 1.5 on that slope.
    double rho_prev = 0.0, rho_last = 0.0;
    int age_prev = -1, age_last = -1;
    bool warm_fresh = false;  // no chunk of this warm-started run has been settled yet: its first chunk is short and carries a spare step
    void note_plan() {  // a plan has just been read back (settle): remember the bound and the age it belongs to
        if (age_last >= 0 && age() > age_last) { rho_prev = rho_last; age_prev = age_last; }
        rho_last = eng.last.rho; age_last = age();
    }
    double growth_ratio(int ahead) const {
        double r = (double)(age() + ahead + 1) / (double)std::max(age(), 1);
        if (age_prev >= 0 && age_last > age_prev && rho_last > 0.0 && rho_last > rho_prev) {
            const double slope = (rho_last - rho_prev) / (double)(age_last - age_prev);
            r = std::max(r, (rho_last + 1.5 * slope * (double)(ahead + 1)) / rho_last);
        }
        return r;
    }
    long long n_first_iters = 0;
    bool rs_last = false;    // the last iteration enqueued left rsfx for the X the next one starts from
    const bool rs_enabled = getenv("MMW_NO_SDDMM_ROWSUMS") == nullptr;
    const bool fv_in_sddmm = !(getenv("MMW_FV_IN_SDDMM") && atoi(getenv("MMW_FV_IN_SDDMM")) == 0);  // where the first-order certificate's workgroups run
    long long n_rs_iters = 0, n_fused_iters = 0;  // MMW_F_DUAL_INFO
    bool sddmm_mfma = false;
    size_t afrag_n = 0;
    int blocking_mode = 1;  // 1: use when profitable, 0: never
    // optimistic (no per-iteration readback) batches: snapshot for the rare replay
    DevBuf<T> sn_lval, sn_xval, sn_xavg, sn_Y, sn_yavg, sn_eaccu;
    bool pending = false;
    // the last chunk ran the shipped path to its end, was settled without a violation and nothing has touched the iterate since: the next
    // chunk's first iteration may continue on the lagged plan and the shifted softmax instead of restarting them exactly
    bool chain_ok = false;
    bool plan_seen = false;  // eng.last holds a plan read back in this run (settle)
    int pend_iter0 = 0, pend_n = 0, m_guess = 3;
    size_t pend_events0 = 0;  // phase-timer events recorded before the pending chunk
    uint64_t pend_seed = 0;
    int replays = 0;
    DevBuf<double> emax_d;
    double emax_h = 0.0;
    int emax_enq_iter = -1, emax_iter = -1;  // iteration count the enqueued / fetched maximum violation belongs to
    bool exact_plans_only = false;  // a cautious second attempt at a discarded chunk is running (settle)
    const bool cautious_replay = !(getenv("MMW_CAUTIOUS_REPLAY") && atoi(getenv("MMW_CAUTIOUS_REPLAY")) == 0);
    ExpmEngine<T> eng;
    Extras<T> extras;
    KernelTimers kt;
    std::vector<hipEvent_t> events;  // 4 per timed iteration
    std::vector<hipEvent_t> event_pool;  // events of earlier runs, kept for reuse
    std::vector<double> phase_us;
    struct PhaseSample { int it; double us[4]; };
    std::vector<PhaseSample> phase_samples;  // the iterations that carried events (set_timing)
    std::vector<int> ev_iter;                // iteration of every group of four pending events
    int timing_stride = 1;
    uint64_t last_seed = 0;
    bool last_was_rng = false;

    ~Solver() override {
        if (blk_thread.joinable()) blk_thread.join();  // it works on this handle's members
        if (host_only) return;
        (void)hipSetDevice(device);
        for (auto e : events) (void)hipEventDestroy(e);
        for (auto e : event_pool) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
    }

    PatternDev<T> pat() const {
        PatternDev<T> P;
        P.K = K; P.Z = Z; P.E_asso = (int)H.E_asso(); P.C = (int)H.C(); P.nnzL = (int)H.nnzL();
        P.indptr = d_indptr.p; P.col = d_col.p; P.pid = d_pid.p; P.mirror = d_mirror.p; P.diag_pos = d_diag.p;
        P.asso_pos = d_apos.p; P.sab = d_sab.p; P.sba = d_sba.p; P.h_max = d_h.p; P.S_sum = d_ssum.p;
        P.inv_norm_H = d_invn.p; P.cH = d_cH.p;
        if (x_tiles) { P.e2w = b_e2w.p; P.xasso = b_xasso.p; P.xdiag_base = (int)HB.m_nedges; }
        return P;
    }

    int init(int dev, int32_t K_, int32_t Z_, int32_t rr, double eta_, int32_t nit_, const int32_t* Sp, const int32_t* Si,
             const double* Sx, const int32_t* Qp, const int32_t* Qi, const double* Qx, const double* h) {
        device = dev;
        const bool verbose = getenv("MMW_VERBOSE") != nullptr;
        auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_0 = tnow();
        // The first kernel launch of a process loads the library's code object (~0.15 s): start it on a helper thread now, under
        // the host-side pattern build.
        static std::atomic<bool> module_loading{false};
        std::thread warm_thread;
        if (!host_only && !module_loading.exchange(true))
            warm_thread = std::thread([dev]() {
                if (hipSetDevice(dev) != hipSuccess) return;
                float* p = nullptr;
                if (hipMalloc((void**)&p, 256 * sizeof(float)) != hipSuccess) return;
                hipLaunchKernelGGL((k_fill<float>), dim3(1), dim3(BLOCK), 0, (hipStream_t) nullptr, (size_t)256, p, 0.0f);
                (void)hipDeviceSynchronize();
                (void)hipFree(p);
            });
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{warm_thread};
        {   // whether the matrix-core blocking is wanted depends on the block's padded width only
            BlockLayout lay0;
            std::string lerr;
            blk_want_mf = sizeof(T) == 4 && !getenv("MMW_NO_MFMA") && make_layout(Z_ * rr, V16<T>::N, lay0, lerr) == MMW_OK &&
                          (double)K_ * lay0.Dpad * 4.0 < 4.0e9;
        }
        const char* blk_env = getenv("MMW_BLOCKING");
        const bool start_blk = !host_only && !(blk_env && blk_env[0] == '0');
        double t_struct = 0.0;
        std::string err = build_pattern(H, K_, Z_, Sp, Si, Sx, Qp, Qi, Qx, h, [&]() {
            t_struct = tnow();
            if (start_blk) blk_thread = std::thread([this]() { host_blockings(); });
        });
        if (!err.empty() && blk_thread.joinable()) blk_thread.join();
        const double t_1 = tnow();
        if (!err.empty()) return fail(MMW_ERR_ARG, "mmw_create: " + err);
        K = K_; Z = Z_; rank_radio = rr; eta = eta_; nit = nit_;
        D = Z * rank_radio;
        if (host_only) {  // device == -1: pattern inspection only (CPU tests of the host logic)
            std::string lerr;
            if (make_layout(D, V16<T>::N, eng.lay, lerr) != MMW_OK) return fail(MMW_ERR_ARG, lerr);
            if (getenv("MMW_HOST_BLOCKING")) {  // developer aid: build the locality blocking on the host and print its statistics
                const double t0 = tnow();
                build_blocking(HB, K, H.l_indptr, H.l_indices, blocking_limits<T>());
                build_sd_tables(HB, K, H.l_indptr, H.l_indices);
                fprintf(stderr, "[mmw] host blocking %.1f ms: usable %d half-tile %d blocks %d rows/block %.1f union/block %.1f reuse %.2f entries %lld (nnz %lld, +%.1f%% padding) sd2_rounds %d\n",
                        (tnow() - t0) * 1e3, (int)HB.usable, (int)HB.fits_half_tile, HB.nb(), (double)K / std::max(1, HB.nb()),
                        (double)HB.un_cols.size() / std::max(1, HB.nb()), HB.reuse, (long long)HB.nent, (long long)H.nnzL(),
                        100.0 * ((double)HB.nent / (double)H.nnzL() - 1.0), HB.sd2_rounds);
                const double t1 = tnow();
                build_mfma_blocking(HB, K, H.l_indptr, H.l_indices, getenv("MMW_MF_ROWS") ? atoi(getenv("MMW_MF_ROWS")) : 64);
                fprintf(stderr, "[mmw] matrix-core blocking %.1f ms: ok %d blocks %d rows/block %.1f reuse %.2f row tiles %d k-steps %d\n", (tnow() - t1) * 1e3,
                        (int)HB.fits_mfma, HB.nbm(), (double)K / std::max(1, HB.nbm()), HB.m_reuse, HB.mfma_mt, HB.kbase.empty() ? 0 : HB.kbase.back());
                if (getenv("MMW_HOST_BLOCKING_HIST")) {  // k-steps of every block, in launch order
                    for (int b = 0; b < HB.nbm(); ++b) fprintf(stderr, "%d:%d ", HB.m_desc[(size_t)b * 8 + 1], HB.kbase[b + 1] - HB.kbase[b]);
                    fprintf(stderr, "\n");
                }
            }
            if (getenv("MMW_CHECK_BLOCKING")) {  // CPU tests: build the blocking and check its invariants
                const BlockingLimits lim = blocking_limits<T>();
                if (HB.order.empty()) build_blocking(HB, K, H.l_indptr, H.l_indices, lim);
                build_sd_tables(HB, K, H.l_indptr, H.l_indices);
                if (!HB.order.empty() && !HB.blk_rowptr.empty() && HB.blk_rowptr.back() == K) {
                    const std::string berr = verify_blocking(HB, K, H.l_indptr, H.l_indices, lim);
                    if (!berr.empty()) return fail(MMW_ERR_STATE, "blocking invariant violated: " + berr);
                    for (int mrows : {64, 32, 7}) {
                        build_mfma_blocking(HB, K, H.l_indptr, H.l_indices, mrows);
                        const std::string merr = verify_mfma_blocking(HB, K, H.l_indptr, H.l_indices);
                        if (!merr.empty()) return fail(MMW_ERR_STATE, "blocking invariant violated: " + merr);
                    }
                }
            }
            return MMW_OK;
        }
        MMW_HIP(hipSetDevice(device));
        MMW_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        MMW_TRY(d_indptr.upload(H.l_indptr, st));
        MMW_TRY(d_col.upload(H.l_indices, st));
        MMW_TRY(d_pid.upload(H.pid, st));
        MMW_TRY(d_mirror.upload(H.mirror, st));
        MMW_TRY(d_diag.upload(H.diag_pos, st));
        MMW_TRY(d_apos.upload(H.asso_pos, st));
        {
            std::vector<int32_t> lrow((size_t)H.nnzL());
            for (int k = 0; k < K; ++k)
                for (int e = H.l_indptr[k]; e < H.l_indptr[k + 1]; ++e) lrow[e] = k;
            MMW_TRY(d_lrow.upload(lrow, st));
        }
        MMW_TRY(d_sab.upload_cast(H.sab, st));
        MMW_TRY(d_sba.upload_cast(H.sba, st));
        MMW_TRY(d_h.upload_cast(H.h_max, st));
        MMW_TRY(d_ssum.upload_cast(H.S_sum, st));
        std::vector<double> invn(K);
        for (int k = 0; k < K; ++k) invn[k] = 1.0 / H.norm_H[k];
        MMW_TRY(d_invn.upload_cast(invn, st));
        MMW_TRY(d_cH.upload_cast(H.cH, st));
        return init_common(t_0, t_1, t_struct, /*env=*/nullptr);
    }

    // ---- mmw_create_from_env: the state never leaves the device.  The generator's receive powers are turned into the pattern, its
    // per-entry arrays, the edge lists and the row statistics by the kernels of pattern_device.h; the host gets the row pointers (from
    // the count pass's prefix sums), the column indices (the blockings read them) and three K-vectors.  The lists that only the
    // API's read fields hand out stay on the device until asked for (ensure_host_lists).
    struct EnvLists {  // device copies kept for ensure_host_lists
        DevBuf<int> st_ptr, st_idx, gain_x, gain_y, asso_x, asso_y, gu_ptr, qu_ptr, so_ptr;
        DevBuf<double> st_val, s_sum, sq_sum;
        bool host_done = true;  // false: H's list vectors are still empty
    } envl;
    int ensure_host_lists() {
        if (envl.host_done) return MMW_OK;
        MMW_HIP(hipSetDevice(device));
        const size_t nst = (size_t)H.n_st, ng = (size_t)H.n_gain, na = (size_t)H.n_asso;
        H.st_indices.resize(nst); H.st_data.resize(nst);
        H.gain_x.resize(ng); H.gain_y.resize(ng); H.asso_x.resize(na); H.asso_y.resize(na);
        H.diag_pos.resize(K); H.asso_pos.resize(na);
        MMW_TRY(copy_d2h(H.st_indices.data(), envl.st_idx.p, nst * sizeof(int32_t), st));
        MMW_TRY(copy_d2h(H.st_data.data(), envl.st_val.p, nst * sizeof(double), st));
        MMW_TRY(copy_d2h(H.gain_x.data(), envl.gain_x.p, ng * sizeof(int32_t), st));
        MMW_TRY(copy_d2h(H.gain_y.data(), envl.gain_y.p, ng * sizeof(int32_t), st));
        MMW_TRY(copy_d2h(H.asso_x.data(), envl.asso_x.p, na * sizeof(int32_t), st));
        MMW_TRY(copy_d2h(H.asso_y.data(), envl.asso_y.p, na * sizeof(int32_t), st));
        MMW_TRY(copy_d2h(H.diag_pos.data(), d_diag.p, (size_t)K * sizeof(int32_t), st));
        MMW_TRY(copy_d2h(H.asso_pos.data(), d_apos.p, na * sizeof(int32_t), st));
        envl.host_done = true;
        return MMW_OK;
    }
    // Row order of a geometric instance: boustrophedon strips about one block wide (blocking.h: consecutive runs of it are compact patches)
    static std::vector<int32_t> spatial_order(int K, const std::vector<double>& xy, int rows_per_block) {
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (int k = 0; k < K; ++k) {
            x0 = std::min(x0, xy[2 * k]); x1 = std::max(x1, xy[2 * k]);
            y0 = std::min(y0, xy[2 * k + 1]); y1 = std::max(y1, xy[2 * k + 1]);
        }
        const double area = std::max((x1 - x0) * (y1 - y0), 1e-300);
        const double w = std::max(std::sqrt(area * (double)rows_per_block / (double)std::max(K, 1)) * 0.9, 1e-300);  // a block is ~ w x w
        std::vector<std::pair<std::pair<int64_t, double>, int32_t>> key(K);
        for (int k = 0; k < K; ++k) {
            const int64_t strip = (int64_t)((xy[2 * k] - x0) / w);
            key[k] = {{strip, (strip & 1) ? -xy[2 * k + 1] : xy[2 * k + 1]}, k};
        }
        std::sort(key.begin(), key.end());
        std::vector<int32_t> ord(K);
        for (int k = 0; k < K; ++k) ord[k] = key[k].second;
        return ord;
    }
    EnvDevice* env_src = nullptr;  // during init_env only
    int init_env(int dev, EnvDevice& E, int32_t Z_, int32_t rr, double eta_, int32_t nit_) {
        device = dev;
        auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_0 = tnow();
        if (Z_ < 2) return fail(MMW_ERR_ARG, "mmw_create_from_env: Z must be >= 2 (the constraints divide by Z-1)");
        if (E.K < 2) return fail(MMW_ERR_ARG, "mmw_create_from_env: K must be >= 2");
        MMW_HIP(hipSetDevice(device));
        MMW_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        MMW_TRY(E.pattern_inputs());  // (cached in the generator: rxT, positions in the AP lists, the count pass)
        K = E.K; Z = Z_; rank_radio = rr; eta = eta_; nit = nit_;
        D = Z * rank_radio;
        H.K = K; H.Z = Z;
        const int A = E.A;
        const int32_t* c6 = E.h_cnt6.data();
        // prefix sums of the count pass
        std::vector<int32_t> st_ptr(K + 1, 0), gu_ptr(K + 1, 0), qu_ptr(K + 1, 0), so_ptr(K + 1, 0);
        H.l_indptr.assign(K + 1, 0);
        int maxdeg = 1, maxq = 1;
        for (int k = 0; k < K; ++k) {
            H.l_indptr[k + 1] = H.l_indptr[k] + c6[k];
            st_ptr[k + 1] = st_ptr[k] + c6[(size_t)K + k];
            gu_ptr[k + 1] = gu_ptr[k] + c6[(size_t)2 * K + k];
            qu_ptr[k + 1] = qu_ptr[k] + c6[(size_t)3 * K + k];
            const int so_len = (E.h_sptr[k + 1] - E.h_sptr[k]) - c6[(size_t)4 * K + k];
            so_ptr[k + 1] = so_ptr[k] + so_len;
            maxdeg = std::max(maxdeg, so_len);
            maxq = std::max(maxq, E.h_qptr[k + 1] - E.h_qptr[k]);
            if ((int64_t)H.l_indptr[k] + c6[k] > (int64_t)INT32_MAX) return fail(MMW_ERR_ARG, "mmw_create_from_env: pattern too large for int32 indexing");
        }
        env_maxdeg = maxdeg; env_maxq = maxq;
        const size_t nnz = (size_t)H.l_indptr[K], nst = (size_t)st_ptr[K], ng = (size_t)gu_ptr[K], na = (size_t)qu_ptr[K];
        H.n_st = (int64_t)nst; H.n_gain = (int64_t)ng; H.n_asso = (int64_t)na;
        H.st_indptr = st_ptr;
        MMW_TRY(d_indptr.upload(H.l_indptr, st));
        MMW_TRY(envl.st_ptr.upload(st_ptr, st)); MMW_TRY(envl.gu_ptr.upload(gu_ptr, st)); MMW_TRY(envl.qu_ptr.upload(qu_ptr, st)); MMW_TRY(envl.so_ptr.upload(so_ptr, st));
        MMW_TRY(d_col.alloc(nnz)); MMW_TRY(d_lrow.alloc(nnz)); MMW_TRY(d_sab.alloc(nnz)); MMW_TRY(d_sba.alloc(nnz)); MMW_TRY(d_pid.alloc(nnz)); MMW_TRY(d_mirror.alloc(nnz));
        MMW_TRY(d_diag.alloc(K)); MMW_TRY(d_apos.alloc(na));
        MMW_TRY(envl.st_idx.alloc(nst)); MMW_TRY(envl.st_val.alloc(nst));
        MMW_TRY(envl.gain_x.alloc(ng)); MMW_TRY(envl.gain_y.alloc(ng)); MMW_TRY(envl.asso_x.alloc(na)); MMW_TRY(envl.asso_y.alloc(na));
        MMW_TRY(envl.s_sum.alloc(K)); MMW_TRY(envl.sq_sum.alloc(K));
        PatOut<T> O;
        O.l_ptr = d_indptr.p; O.st_ptr = envl.st_ptr.p; O.gu_ptr = envl.gu_ptr.p; O.qu_ptr = envl.qu_ptr.p; O.appos = E.appos.p;
        O.l_idx = d_col.p; O.lrow = d_lrow.p; O.sab = d_sab.p; O.sba = d_sba.p; O.pid = d_pid.p; O.diag_pos = d_diag.p;
        O.st_idx = envl.st_idx.p; O.st_val = envl.st_val.p;
        O.gain_x = envl.gain_x.p; O.gain_y = envl.gain_y.p; O.asso_x = envl.asso_x.p; O.asso_y = envl.asso_y.p; O.asso_pos = d_apos.p;
        hipLaunchKernelGGL((k_pat_fill<T>), dim3(grid_rows(K)), dim3(BLOCK), 0, st, K, A, E.P.thr, E.rx.p, E.rxT.p, E.asso.p, O);
        MMW_HIP(hipGetLastError());
        // the column indices first: the host-side blockings start on them while the device finishes the rest
        H.l_indices.resize(nnz);
        MMW_TRY(copy_d2h(H.l_indices.data(), d_col.p, nnz * sizeof(int32_t), st));
        const double t_struct = tnow();
        {   // whether the matrix-core blocking is wanted depends on the block's padded width only
            BlockLayout lay0;
            std::string lerr;
            blk_want_mf = sizeof(T) == 4 && !getenv("MMW_NO_MFMA") && make_layout(Z_ * rr, V16<T>::N, lay0, lerr) == MMW_OK && (double)K * lay0.Dpad * 4.0 < 4.0e9;
        }
        const char* blk_env = getenv("MMW_BLOCKING");
        if (!(blk_env && blk_env[0] == '0')) {
            if (!getenv("MMW_ENV_RCM")) {  // (MMW_ENV_RCM=1: the pattern-only order of the CSR entry point, for comparisons)
                HB.rcm_cache = spatial_order(K, E.h_sta, 64);
                HB.grow = false;
            }
            blk_thread = std::thread([this]() { host_blockings(); });
        }
        hipLaunchKernelGGL(k_pat_mirror, dim3(grid_elems(nnz)), dim3(BLOCK), 0, st, nnz, d_indptr.p, d_col.p, d_lrow.p, d_mirror.p);
        hipLaunchKernelGGL(k_pat_rowstats, dim3(grid_elems((size_t)K)), dim3(BLOCK), 0, st, K, envl.st_ptr.p, envl.st_val.p, envl.s_sum.p, envl.sq_sum.p);
        MMW_HIP(hipGetLastError());
        H.S_sum.resize(K); H.sq_sum.resize(K); H.h_max.resize(K);
        MMW_TRY(copy_d2h(H.S_sum.data(), envl.s_sum.p, (size_t)K * sizeof(double), st));
        MMW_TRY(copy_d2h(H.sq_sum.data(), envl.sq_sum.p, (size_t)K * sizeof(double), st));
        MMW_TRY(copy_d2h(H.h_max.data(), E.h_max.p, (size_t)K * sizeof(double), st));
        H.norm_H.assign(K, 0.0);
        H.cH.assign(K, 0.0);
        {
            const std::string err = update_slots(H, Z);
            if (!err.empty()) {
                if (blk_thread.joinable()) blk_thread.join();
                return fail(MMW_ERR_ARG, "mmw_create_from_env: " + err);
            }
        }
        MMW_TRY(d_h.upload_cast(H.h_max, st));
        MMW_TRY(d_ssum.upload_cast(H.S_sum, st));
        std::vector<double> invn(K);
        for (int k = 0; k < K; ++k) invn[k] = 1.0 / H.norm_H[k];
        MMW_TRY(d_invn.upload_cast(invn, st));
        MMW_TRY(d_cH.upload_cast(H.cH, st));
        envl.host_done = false;
        const double t_1 = tnow();
        env_src = &E;
        const int rc = init_common(t_0, t_1, t_struct, &E);
        env_src = nullptr;
        return rc;
    }
    int env_maxdeg = 1, env_maxq = 1;
    // the rounding's view of the state, from the generator's own CSR of S_gain (diagonal dropped) and Q
    int init_extras_env(EnvDevice& E) {
        MMW_TRY(extras.init_device(st, K, &kt, env_maxdeg, env_maxq));
        std::vector<int32_t> so_ptr_h((size_t)K + 1);
        MMW_TRY(copy_d2h(so_ptr_h.data(), envl.so_ptr.p, so_ptr_h.size() * sizeof(int32_t), st));
        const size_t nso = (size_t)so_ptr_h[K];
        MMW_TRY(extras.so_indptr.upload(so_ptr_h, st));
        MMW_TRY(extras.so_indices.alloc(nso)); MMW_TRY(extras.so_data.alloc(nso)); MMW_TRY(extras.so_hmax.alloc(nso));
        MMW_TRY(extras.q_indptr.alloc((size_t)K + 1)); MMW_TRY(extras.q_indices.alloc((size_t)E.nnzQ)); MMW_TRY(extras.h_max.alloc(K));
        hipLaunchKernelGGL(k_pat_so_fill, dim3(grid_rows(K)), dim3(BLOCK), 0, st, K, (const int*)E.s_ptr.p, (const int*)E.s_idx.p, (const double*)E.s_val.p,
                           (const int*)extras.so_indptr.p, (const double*)E.h_max.p, extras.so_indices.p, extras.so_data.p, extras.so_hmax.p);
        MMW_HIP(hipGetLastError());
        MMW_HIP(hipMemcpyAsync(extras.q_indptr.p, E.q_ptr.p, ((size_t)K + 1) * sizeof(int), hipMemcpyDeviceToDevice, st));
        MMW_HIP(hipMemcpyAsync(extras.q_indices.p, E.q_idx.p, (size_t)E.nnzQ * sizeof(int), hipMemcpyDeviceToDevice, st));
        MMW_HIP(hipMemcpyAsync(extras.h_max.p, E.h_max.p, (size_t)K * sizeof(double), hipMemcpyDeviceToDevice, st));
        MMW_HIP(hipStreamSynchronize(st));
        return MMW_OK;
    }

    // ---- everything after the pattern is on the device: the iterate's buffers, the engine, the blockings, the rounding side
    int init_common(double t_0, double t_1, double t_struct, EnvDevice* env) {
        const bool verbose = getenv("MMW_VERBOSE") != nullptr;
        auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const size_t nnz = (size_t)H.nnzL(), C = (size_t)H.C();
        MMW_TRY(lval.alloc(nnz)); MMW_TRY(xval.alloc(nnz)); MMW_TRY(xavg.alloc(nnz));
        MMW_TRY(Y.alloc(C)); MMW_TRY(yavg.alloc(C)); MMW_TRY(e_accu.alloc(C)); MMW_TRY(e_this.alloc(C));
        MMW_TRY(rsum.alloc(K)); MMW_TRY(drow.alloc(K));
        MMW_TRY(max_part.alloc(ROW_GRID_MAX)); MMW_TRY(sum_part.alloc(4 * (size_t)std::max(2048, ROW_GRID_MAX))); MMW_TRY(scal.alloc(8));
        MMW_TRY(tr_part.alloc(ROW_GRID_MAX)); MMW_TRY(wH.alloc(K));
        MMW_TRY(eng.init(st, K, D, d_indptr.p, d_col.p, lval.p));
        kt.st = st;
        eng.kt = &kt;
        eng.max_order = 12;
        eng.tol = sizeof(T) == 4 ? 1e-6 : 1e-9;
        MMW_TRY(Xh.alloc(eng.bs));
        MMW_TRY(prealloc());
        const double t_2 = tnow();
        MMW_TRY(setup_blocking());
        const double t_3 = tnow();
        if (verbose) fprintf(stderr, "[create] pattern %.1f ms (structure after %.1f), uploads+alloc %.1f ms, blocking %.1f ms\n", (t_1 - t_0) * 1e3, (t_struct - t_0) * 1e3, (t_2 - t_1) * 1e3, (t_3 - t_2) * 1e3);
        size_t big = std::max(std::max(nnz, C), eng.bs);
        MMW_TRY(out64.alloc(big));
        MMW_TRY(stage64.alloc((size_t)K * D));
        MMW_HIP(hipStreamSynchronize(st));
        if (env) MMW_TRY(init_extras_env(*env));
        else MMW_TRY(extras.init(this->st, &H, K, &kt));
        return reset(nit);
    }

    BlkDev blkdev() const {
        BlkDev B;
        B.nb = HB.nb(); B.rowptr = b_rowptr.p; B.order = b_order.p; B.un_ptr = b_unptr.p; B.un_cols = b_uncols.p;
        B.bptr = b_bptr.p; B.lidx = b_lidx.p; B.self_li = b_selfli.p; B.desc = b_desc.p; B.un_fixed = b_unfixed.p;
        B.half_tile = HB.fits_half_tile && (double)K * eng.lay.Dpad * sizeof(T) < 4.0e9 && !getenv("MMW_FULL_TILE");  // 32-bit byte offsets
        return B;
    }
    // Buffers the loop, the factor and the rounding would otherwise allocate on first use (hipMalloc is a synchronous driver call
    // of 0.1 - 3 ms, and the first probe of a search pays all of them inside its timed phases): reserved here, while the
    // blocking thread is still at work and this thread would only wait for it.
    int prealloc() {
        const size_t nnz = (size_t)H.nnzL(), C = (size_t)H.C();
        for (DevBuf<T>* b : {&sn_lval, &sn_xval, &sn_xavg}) MMW_TRY(b->alloc(nnz));
        for (DevBuf<T>* b : {&sn_Y, &sn_yavg, &sn_eaccu, &yun}) MMW_TRY(b->alloc(C));
        MMW_TRY(sn_plan.alloc(1));
        // Krylov basis: the first iterations of a run ask for 2 - 3 steps before the a-posteriori estimate settles on fewer; growing
        // the basis there costs an allocation, a copy and two device synchronisations each time
        if ((double)eng.bs * sizeof(T) * 4.0 < 8.0e9) MMW_TRY(eng.ensure_blocks(std::min(4, eng.max_order + 1)));
        if (blk_want_mf && (eng.lay.Dpad % 32) == 0) {
            MMW_TRY(xh_planes.alloc(2 * eng.bs));
            MMW_TRY(eng.reserve_planes());
        }
        const int rank = std::min(K - 1, (Z - 1) * rank_radio);  // what the host class asks mmw_factor for (mmw.py:206)
        if (rank >= 1) {
            MMW_TRY(extras.fac_reserve(st, K, rank, blk_want_mf));
            if ((size_t)10 * K * Z * sizeof(double) <= ((size_t)2 << 30)) MMW_TRY(extras.round_reserve(K, Z, rank, 10));  // sdp_solver.rounding's 10 attempts
        }
        return MMW_OK;
    }
    // The host side of both blockings: one RCM order, then the two block builders on two threads (they fill disjoint parts of HB).
    // Reads only the pattern's structure, so init() starts it while build_pattern is still making the mirrors and edge lists.
    std::thread blk_thread;
    bool blk_want_mf = false;
    void host_blockings() {
        const int Kp = H.K;
        bool rows_ok = true;
        for (int k = 0; k < Kp && rows_ok; ++k) rows_ok = H.l_indptr[k + 1] - H.l_indptr[k] <= BLK_UNION;
        if (rows_ok && HB.rcm_cache.size() != (size_t)Kp) HB.rcm_cache = rcm_order(Kp, H.l_indptr, H.l_indices);  // (a handle made from the generator brings a spatial order)
        const int mrows = getenv("MMW_MF_ROWS") ? atoi(getenv("MMW_MF_ROWS")) : 64;
        std::thread mf_thread;
        if (blk_want_mf && rows_ok) mf_thread = std::thread([&]() { build_mfma_blocking(HB, Kp, H.l_indptr, H.l_indices, std::min(64, std::max(1, mrows))); });
        build_blocking(HB, Kp, H.l_indptr, H.l_indices, blocking_limits<T>());
        if (mf_thread.joinable()) mf_thread.join();
    }
    // Entry tables of the LDS-staged SDDMM kernels: built and uploaded on first need (a handle whose SDDMM runs on the matrix
    // cores never asks; blocking.h, build_sd_tables)
    bool sd_up = false;
    int ensure_sd() {
        if (sd_up || !HB.usable) return MMW_OK;
        sd_up = true;
        build_sd_tables(HB, K, H.l_indptr, H.l_indices);
        // the block records carry every block's first slot of the half-tile SDDMM: refreshed in place (the kernels' argument
        // structs hold this buffer's address)
        if (b_desc.n != HB.desc.size()) return fail(MMW_ERR_STATE, "internal: block records changed size");
        MMW_HIP(hipMemcpyAsync(b_desc.p, HB.desc.data(), HB.desc.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (HB.sd_max <= SD_ROUNDS * BLK_THREADS) {
            MMW_TRY(b_sdptr.upload(HB.sd_ptr, st)); MMW_TRY(b_sdla.upload(HB.sd_la, st)); MMW_TRY(b_sdlb.upload(HB.sd_lb, st));
            MMW_TRY(b_sdepos.upload(HB.sd_epos, st));
            const size_t shb = (size_t)BLK_UNION_ROWS * BLK_TILE_BYTES;
            MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sddmm_blk<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
            sddmm_blk = true;
        }
        if ((double)K * eng.lay.Dpad * sizeof(T) < 4.0e9 && !getenv("MMW_FULL_TILE")) {
            MMW_TRY(b_sd2ptr.upload(HB.sd2_ptr, st)); MMW_TRY(b_sd2ab.upload(HB.sd2_ab, st)); MMW_TRY(b_sd2epos.upload(HB.sd2_epos, st));
            {   // Work items.  A workgroup is a latency chain whose length is its number of rounds, and the launch lasts as long
                // as its longest workgroup; the resident slots the row blocks leave free are used to cut the longest items in two
                // (each half stages the union again).
                const int cus = device_cus();
                const int per_cu = std::max(1, std::min(2048 / SD2_THREADS, 163840 / std::max(1, HB.un8_max * B2_ROW_BYTES + 128)));
                const size_t slots = (size_t)per_cu * (size_t)cus;
                struct It { int rb, k0, k1; };
                auto len = [](const It& a) { return a.k1 - a.k0; };
                auto less = [&](const It& a, const It& b) { return len(a) != len(b) ? len(a) < len(b) : a.rb > b.rb; };
                std::priority_queue<It, std::vector<It>, decltype(less)> pq(less);
                for (int b = 0; b < HB.nb(); ++b) pq.push({b, 0, (HB.sd2_ptr[b + 1] - HB.sd2_ptr[b]) / SD2_THREADS});
                while ((pq.size() < slots && len(pq.top()) >= 2) || len(pq.top()) > sd2_rounds<T>()) {
                    const It t = pq.top();
                    pq.pop();
                    const int mid = t.k0 + (len(t) + 1) / 2;
                    pq.push({t.rb, t.k0, mid});
                    pq.push({t.rb, mid, t.k1});
                }
                std::vector<int32_t> items;
                while (!pq.empty()) {  // longest first
                    items.push_back(pq.top().rb); items.push_back(pq.top().k0); items.push_back(pq.top().k1);
                    pq.pop();
                }
                sd2_nitems = (int)(items.size() / 3);
                MMW_TRY(b_sd2items.upload(items, st));
            }
            MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sddmm_blk2<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        std::max(HB.un8_max * B2_ROW_BYTES, 65536)));
            sddmm_blk2 = true;
        }
        MMW_HIP(hipStreamSynchronize(st));  // the uploads read host vectors
        return MMW_OK;
    }
    int setup_blocking() {
        const char* env = getenv("MMW_BLOCKING");
        if (env && env[0] == '0') blocking_mode = 0;
        if (!blocking_mode) return MMW_OK;
        if (blk_thread.joinable()) blk_thread.join();  // started under the pattern build (init)
        else host_blockings();
        {
            const bool want_mf = blk_want_mf;
            if (want_mf && getenv("MMW_VERBOSE"))
                fprintf(stderr, "[mmw] matrix-core blocking: ok %d blocks %d rows/block %.1f reuse %.2f row tiles %d k-steps %d\n", (int)HB.fits_mfma, HB.nbm(),
                        (double)K / std::max(1, HB.nbm()), HB.m_reuse, HB.mfma_mt, HB.kbase.empty() ? 0 : HB.kbase.back());
        }
        if (getenv("MMW_VERBOSE"))
            fprintf(stderr, "[mmw] blocking: usable %d half-tile %d blocks %d rows/block %.1f union/block %.1f entries %lld (nnz %lld, +%.1f%% padding) sd_max %d\n",
                    (int)HB.usable, (int)HB.fits_half_tile, HB.nb(), (double)K / std::max(1, HB.nb()), (double)HB.un_cols.size() / std::max(1, HB.nb()),
                    (long long)HB.nent, (long long)H.nnzL(), 100.0 * ((double)HB.nent / (double)H.nnzL() - 1.0), HB.sd_max);
        if (!HB.usable) return MMW_OK;
        MMW_TRY(b_rowptr.upload(HB.blk_rowptr, st)); MMW_TRY(b_order.upload(HB.order, st)); MMW_TRY(b_unptr.upload(HB.un_ptr, st));
        MMW_TRY(b_uncols.upload(HB.un_cols, st)); MMW_TRY(b_bptr.upload(HB.bptr, st)); MMW_TRY(b_bpos.upload(HB.bpos, st));
        MMW_TRY(b_bepos.upload(HB.bepos, st)); MMW_TRY(b_lidx.upload(HB.lidx, st)); MMW_TRY(b_selfli.upload(HB.self_li, st)); MMW_TRY(b_desc.upload(HB.desc, st)); MMW_TRY(b_unfixed.upload(HB.un_fixed, st));
        MMW_TRY(lval_blk.alloc((size_t)HB.nent));
        if (sizeof(T) == 4 && HB.fits_mfma) {
            MMW_TRY(b_kbase.upload(HB.kbase, st));
            MMW_TRY(b_fpos.upload(HB.fpos, st));
            MMW_TRY(b_mdesc.upload(HB.m_desc, st));
            MMW_TRY(b_munfixed.upload(HB.m_unfixed, st));
            MMW_TRY(b_morder.upload(HB.m_order, st));
            afrag_n = (size_t)HB.kbase[HB.nbm()] * HB.mfma_mt * 512;
            MMW_TRY(afrag.alloc(afrag_n));
            MMW_HIP(hipMemsetAsync(afrag.p, 0, afrag_n * sizeof(unsigned), st));
            MMW_TRY(afrag16.alloc(afrag_n));
            MMW_HIP(hipMemsetAsync(afrag16.p, 0, afrag_n * sizeof(unsigned short), st));
            eng.use_mfma = true;
            eng.mf.nb = HB.nbm();
            eng.mf.desc = b_mdesc.p;
            eng.mf.un_fixed = b_munfixed.p;
            eng.mf.order = b_morder.p;
            eng.mf.kbase = b_kbase.p;
            eng.mf.afrag = afrag.p;
            eng.mf_mt = HB.mfma_mt;
            if (!getenv("MMW_NO_MFMA_SDDMM")) {
                MMW_TRY(b_tbase.upload(HB.m_tbase, st)); MMW_TRY(b_tptr.upload(HB.m_tptr, st)); MMW_TRY(b_trc.upload(HB.m_trc, st));
                MMW_TRY(b_e2w.upload(HB.m_e2w, st));
                {   // slot of every association pair: the slot of its upper entry
                    const size_t na = (size_t)H.E_asso();
                    MMW_TRY(b_xasso.alloc(na));
                    if (na) hipLaunchKernelGGL(k_gather_idx, dim3(grid_elems(na)), dim3(BLOCK), 0, st, na, (const int*)d_apos.p, (const int*)b_e2w.p, b_xasso.p);
                    MMW_HIP(hipGetLastError());
                }
                n_xs = (size_t)HB.m_nedges + (size_t)K;
                MMW_TRY(xs_val.alloc(n_xs));
                MMW_TRY(xs_avg.alloc(n_xs));
                MMW_TRY(b_tmask.upload(HB.m_tmask, st));
                MMW_TRY(rsfx.alloc((size_t)2 * K));
                sddmm_mfma = true;
            }
            if ((size_t)HB.nbm() > (size_t)MAX_PART && HB.nbm() > HB.nb()) {
                MMW_TRY(eng.partial.alloc((size_t)HB.nbm() * eng.lay.Dpad));
                MMW_TRY(eng.partial_o2.alloc((size_t)HB.nbm() * eng.lay.Dpad));
            }
        }
        if (!sddmm_mfma) MMW_TRY(ensure_sd());
        MMW_HIP(hipStreamSynchronize(st));
        extras.fac.set_blocking(blkdev(), b_bepos.p, HB.nent);
        if (eng.use_mfma) extras.fac.set_mfma(eng.mf, HB.mfma_mt, b_fpos.p, afrag_n, (int64_t)H.nnzL());
        eng.blk_stale = &lblk_stale;
        eng.blk_refresh = [this]() -> int {
            hipLaunchKernelGGL((k_gather_blocked<T>), dim3(grid_elems((size_t)HB.nent)), dim3(BLOCK), 0, st, (size_t)HB.nent, b_bepos.p, lval.p, lval_blk.p);
            MMW_HIP(hipGetLastError());
            return MMW_OK;
        };
        return eng.enable_blocking(blkdev(), lval_blk.p);
    }

    int sizes(int64_t out[10]) override {
        out[0] = K; out[1] = Z; out[2] = D; out[3] = eng.lay.Dpad; out[4] = H.nnzL(); out[5] = H.nnzST();
        out[6] = H.E_gain(); out[7] = H.E_asso(); out[8] = H.C(); out[9] = iter;
        return MMW_OK;
    }
    int set_expm(int method, int max_order, double tol) override {
        if (method != MMW_EXPM_LANCZOS && method != MMW_EXPM_TAYLOR) return fail(MMW_ERR_ARG, "unknown expm method");
        if (max_order < 1 || max_order > MAX_ORDER) return fail(MMW_ERR_ARG, "max_order must be in [1,16]");
        if (!(tol > 0)) return fail(MMW_ERR_ARG, "tol must be positive");
        eng.method = method; eng.max_order = max_order; eng.tol = tol;
        return MMW_OK;
    }
    int set_timing(int enabled) override {
        timing = enabled != 0;
        timing_stride = enabled > 1 ? enabled : 1;
        return MMW_OK;
    }
    // Phase timers of every iteration cost the loop four event records per iteration (each one a barrier packet between two launches of
    // the dependent chain).  With a stride S > 1 only iteration 0 and the iterations i = S/2 (mod S) carry events; every other row of
    // MMW_F_PHASE_US repeats the sample of its group of S iterations (the harness takes means over the rows, sim_mmw_time.py:48-52).
    bool timed_iteration() const { return timing && (timing_stride <= 1 || iter == 0 || iter % timing_stride == timing_stride / 2); }
    // diagnostic: per-workgroup phase stamps written by a blocked kernel (16 slots per workgroup, slot 9 = end,
    // 10 = HW_ID, 11 = XCC_ID): mean time per phase and how many workgroups were resident per CU
    int dump_stamps(const unsigned long long* dev) {
            std::vector<unsigned long long> h((size_t)16 * 8192);
            MMW_HIP(hipMemcpyAsync(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            MMW_HIP(hipStreamSynchronize(st));
            double acc[10] = {0};
            int cnt = 0;
            unsigned long long tmin = ~0ull, tmax = 0;
            for (int w = 0; w < 8192; ++w) {
                const unsigned long long* q = &h[(size_t)w * 16];
                if (!q[0] || !q[9]) continue;
                ++cnt;
                tmin = std::min(tmin, q[0]);
                tmax = std::max(tmax, q[9]);
                for (int k = 1; k < 10; ++k) if (q[k] && q[k - 1]) acc[k] += (double)(q[k] - q[k - 1]);
            }
            fprintf(stderr, "[stamps] %d workgroups, span %.1f us; mean us per phase:", cnt, (double)(tmax - tmin) * 0.01);
            for (int k = 1; k < 10; ++k) fprintf(stderr, " p%d=%.2f", k, acc[k] / std::max(cnt, 1) * 0.01);
            {
                double g = 0; int c = 0;
                for (int w = 0; w < 8192; ++w) {
                    const unsigned long long* q = &h[(size_t)w * 16];
                    if (q[4] && q[12]) { g += (double)(q[12] - q[4]); ++c; }
                }
                if (c) fprintf(stderr, " gather-issue(p5 part)=%.2f", g / c * 0.01);
            }
            fprintf(stderr, "\n");
            // residency: workgroups whose [start, end) intervals overlap on the same (XCC, SE, SH, CU)
            std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
            double wgdur = 0;
            for (int w = 0; w < 8192; ++w) {
                const unsigned long long* q = &h[(size_t)w * 16];
                if (!q[0] || !q[9] || !q[10]) continue;
                const unsigned long long cu = ((q[11] & 0xF) << 16) | (q[10] & 0xFF00);  // xcc | se, sh, cu bits of HW_ID
                ev[cu].push_back({q[0], +1});
                ev[cu].push_back({q[9], -1});
                wgdur += (double)(q[9] - q[0]);
            }
            double t1 = 0, t2 = 0, t3 = 0;
            for (auto& kv : ev) {
                auto& v = kv.second;
                std::sort(v.begin(), v.end());
                int live = 0;
                for (size_t i = 0; i + 1 < v.size(); ++i) {
                    live += v[i].second;
                    const double dt = (double)(v[i + 1].first - v[i].first);
                    if (live == 1) t1 += dt; else if (live == 2) t2 += dt; else if (live >= 3) t3 += dt;
                }
            }
            {
                std::vector<double> dur;
                for (int w = 0; w < 8192; ++w) {
                    const unsigned long long* q = &h[(size_t)w * 16];
                    if (q[0] && q[9]) dur.push_back((double)(q[9] - q[0]) * 0.01);
                }
                std::sort(dur.begin(), dur.end());
                {   // the ten longest workgroups: when they started, how long they ran, where
                    std::vector<std::pair<double, int>> byd;
                    for (int w = 0; w < 8192; ++w) {
                        const unsigned long long* q = &h[(size_t)w * 16];
                        if (q[0] && q[9]) byd.push_back({(double)(q[9] - q[0]) * 0.01, w});
                    }
                    std::sort(byd.rbegin(), byd.rend());
                    for (size_t i = 0; i < byd.size() && i < 10; ++i) {
                        const unsigned long long* q = &h[(size_t)byd[i].second * 16];
                        fprintf(stderr, "[stamps]   wg %4d start +%.1f us dur %.1f us first-tile %.1f us cu %llx\n", byd[i].second, (double)(q[0] - tmin) * 0.01,
                                byd[i].first, q[5] && q[4] ? (double)(q[5] - q[4]) * 0.01 : 0.0, ((q[11] & 0xF) << 16) | (q[10] & 0xFF00));
                    }
                    double late = 0; int nl = 0;
                    for (auto& pr : byd) { const unsigned long long* q = &h[(size_t)pr.second * 16]; const double st0 = (double)(q[0] - tmin) * 0.01; if (st0 > 5.0) { late += st0; ++nl; } }
                    fprintf(stderr, "[stamps]   %d workgroups started later than +5 us (mean +%.1f us)\n", nl, nl ? late / nl : 0.0);
                }
                if (!dur.empty())
                    fprintf(stderr, "[stamps] workgroup us: min %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f max %.1f\n", dur.front(), dur[dur.size() / 4],
                            dur[dur.size() / 2], dur[dur.size() * 3 / 4], dur[dur.size() * 95 / 100], dur.back());
            }
            if (!ev.empty())
                fprintf(stderr, "[stamps] %zu distinct CUs; mean workgroup %.2f us; per CU: %.1f us with 1 resident, %.1f us with 2, %.1f us with 3+\n", ev.size(),
                        wgdur / std::max(cnt, 1) * 0.01, t1 / ev.size() * 0.01, t2 / ev.size() * 0.01, t3 / ev.size() * 0.01);
        return MMW_OK;
    }
    // SpMM micro-benchmark on the current L values: Tm = 0.5 * L * start_block, `reps` launches
    int bench_spmm(int blocked, int reps, double* avg_us) override {
        if (host_only) return fail(MMW_ERR_STATE, "host-only handle");
        MMW_HIP(hipSetDevice(device));
        if (blocked && !HB.usable) return fail(MMW_ERR_STATE, "no locality blocking for this pattern");
        MMW_TRY(sync());
        hipLaunchKernelGGL((k_sketch_rng<T>), dim3(grid_rows(K)), dim3(BLOCK), 0, st, K, D, eng.lay.Dpad, 99ull, 0u, eng.start_block(), (double*)nullptr);
        const bool keep = eng.use_blk, keep_mf = eng.use_mfma;
        eng.use_blk = blocked != 0;
        if (blocked == 2 && !eng.use_mfma) return fail(MMW_ERR_STATE, "no matrix-core SpMM for this handle (fp32, blocks of <= 32 rows)");
        eng.use_mfma = blocked == 2;
        DevBuf<unsigned long long> stamps;
        const bool want_stamps = blocked && getenv("MMW_STAMPS");
        if (want_stamps) {
            MMW_TRY(stamps.alloc((size_t)16 * 8192));
            MMW_HIP(hipMemsetAsync(stamps.p, 0, (size_t)16 * 8192 * sizeof(unsigned long long), st));
        }
        hipEvent_t e0, e1;
        MMW_HIP(hipEventCreate(&e0));
        MMW_HIP(hipEventCreate(&e1));
        const bool lz = getenv("MMW_BENCH_LANCZOS") != nullptr;  // time the Lanczos epilogue (alpha partials) instead of the plain product
        const unsigned short* pl = nullptr;
        if (blocked == 2) {  // the planes are the producer's job: outside the timed launches
            eng.planes_ready[0] = false;
            MMW_TRY(eng.make_planes(0));
            pl = eng.planes_of(0);
        }
        // MMW_BENCH_FIRST: the first-order product as the loop launches it (fp16 operands and its whole epilogue; the operands are whatever the
        // last iteration left -- only the time is of interest)
        const bool fo = blocked == 2 && getenv("MMW_BENCH_FIRST") != nullptr && sizeof(T) == 4 && rsfx.p != nullptr;
        int ntr1 = 0;
        if (fo) {
            if (xh_planes.n < 2 * eng.bs) MMW_TRY(xh_planes.alloc(2 * eng.bs));
            const size_t need = (size_t)eng.first_grid_max();
            if (tr1_part.n < need) {
                MMW_TRY(tr1_part.alloc(need));
                MMW_HIP(hipMemsetAsync(tr1_part.p, 0, need * sizeof(double), st));
            }
            hipLaunchKernelGGL(k_plane_f16, dim3(grid_elems(eng.bs / 4)), dim3(BLOCK), 0, st, eng.bs / 4, reinterpret_cast<const float4*>(eng.start_block()),
                               reinterpret_cast<uint2*>(eng.planes_of(0)));
        }
        auto one = [&]() {
            if (fo) {
                eng.planes_ready[0] = true;
                eng.planes0_f16 = true;
                return eng.apply_first((T*)nullptr, 0.5, 1, true, xh_planes.p, rsfx.p + K, tr1_part.p, &ntr1);
            }
            return lz ? eng.template launch_spmm<SPMM_LANCZOS>(eng.start_block(), eng.Tm.p, nullptr, 0.5, 0.0, 1.0, nullptr, 0, pl)
                      : eng.template launch_spmm<SPMM_PLAIN>(eng.start_block(), eng.Tm.p, nullptr, 0.5, 0.0, 1.0, nullptr, 0, pl);
        };
        int rc = one();  // warm
        MMW_HIP(hipEventRecord(e0, st));
        for (int r = 0; r < reps && rc == MMW_OK; ++r) rc = one();
        MMW_HIP(hipEventRecord(e1, st));
        MMW_HIP(hipStreamSynchronize(st));
        if (want_stamps && blocked == 2) {  // matrix-core kernel: per-wave phase clocks
            g_mf_stamps = stamps.p;
            rc = one();
            g_mf_stamps = nullptr;
            std::vector<unsigned long long> h((size_t)16 * 8192);
            MMW_HIP(hipMemcpyAsync(h.data(), stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            MMW_HIP(hipStreamSynchronize(st));
            double sum[5] = {0}, steps = 0, life_max = 0, epi = 0;
            int n = 0;
            for (size_t w = 0; w < h.size() / 8; ++w) {
                const unsigned long long* q = &h[w * 8];
                if (!q[4]) continue;
                ++n;
                for (int k = 0; k < 5; ++k) sum[k] += (double)q[k];
                steps += (double)q[5];
                life_max = std::max(life_max, (double)q[4]);
                epi += (double)q[7];
            }
            {   // the slowest twentieth of the waves, and lifetime against the block's k-steps
                std::vector<std::pair<unsigned long long, size_t>> byl;
                for (size_t w = 0; w < h.size() / 8; ++w) if (h[w * 8 + 4]) byl.push_back({h[w * 8 + 4], w});
                std::sort(byl.rbegin(), byl.rend());
                const size_t top = std::max<size_t>(1, byl.size() / 20);
                double ts[8] = {0};
                for (size_t i = 0; i < top && i < byl.size(); ++i) for (int k = 0; k < 8; ++k) ts[k] += (double)h[byl[i].second * 8 + k];
                if (!byl.empty())
                    fprintf(stderr, "[mf stamps] slowest %zu waves: chunks %.1f, prologue %.0f, wait+barrier %.0f, issue %.0f, products %.0f, epilogue %.0f, lifetime %.0f\n", top, ts[5] / top,
                            ts[0] / top, ts[1] / top, ts[2] / top, ts[3] / top, ts[7] / top, ts[4] / top);
                double lo = 0, hi = 0; int nlo = 0, nhi = 0;
                for (auto& pr : byl) { const unsigned long long* q = &h[pr.second * 8]; if (q[5] <= 6) { lo += (double)q[4]; ++nlo; } else if (q[5] >= 9) { hi += (double)q[4]; ++nhi; } }
                fprintf(stderr, "[mf stamps] lifetime of waves with <= 6 chunks: %.0f (%d waves); with >= 9 chunks: %.0f (%d waves)\n", nlo ? lo / nlo : 0.0, nlo, nhi ? hi / nhi : 0.0, nhi);
            }
            if (n)
                fprintf(stderr, "[mf stamps] %d waves, %.1f k-steps each; shader clocks per wave: prologue %.0f, wait+barrier %.0f (%.0f/step), issue %.0f (%.0f/step), "
                                "products %.0f (%.0f/step), epilogue %.0f, lifetime %.0f (max %.0f)\n", n, steps / n, sum[0] / n, sum[1] / n, sum[1] / steps,
                        sum[2] / n, sum[2] / steps, sum[3] / n, sum[3] / steps, epi / n, sum[4] / n, life_max);
        } else if (want_stamps) {
            g_blk_stamps = stamps.p;
            rc = one();
            g_blk_stamps = nullptr;
            MMW_TRY(dump_stamps(stamps.p));
        }
        eng.use_blk = keep;
        eng.use_mfma = keep_mf;
        float ms = 0;
        MMW_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (avg_us) *avg_us = ms * 1e3 / (reps > 0 ? reps : 1);
        return rc;
    }
    int set_profile(int enabled) override {
        if (host_only) return fail(MMW_ERR_STATE, "host-only handle");
        MMW_TRY(sync());
        kt.on = enabled != 0;
        kt_shipped = enabled == 2;  // 2: time the launches of the shipped path (chunks without readback, riding workgroups) as they are
        kt.attach = kt_shipped && !getenv("MMW_KT_MARKERS");  // ... the matrix-core product by the events its launch carries itself
        eng.kt_exact = kt.on && !kt_shipped;
        kt.clear();
        return MMW_OK;
    }

    // same state, new slot count: only the Z-dependent scalars and the D-wide blocks change
    int set_eta(double eta_) override {
        if (!(eta_ >= 0.0)) return fail(MMW_ERR_ARG, "eta must be non-negative");
        if (!host_only) {
            MMW_HIP(hipSetDevice(device));
            MMW_TRY(settle());  // a pending chunk was enqueued with the old step size; a replay must use it too
        }
        eta = eta_;
        return MMW_OK;
    }
    int set_slots(int32_t Z_, int32_t nit_, int warm) override {
        if (host_only) return fail(MMW_ERR_STATE, "this handle was created with device -1 (host pattern only)");
        MMW_HIP(hipSetDevice(device));
        MMW_TRY(settle());
        MMW_HIP(hipStreamSynchronize(st));
        if (warm && iter == 0) warm = 0;  // nothing to continue from
        std::string err = update_slots(H, Z_);
        if (!err.empty()) return fail(MMW_ERR_ARG, "mmw_set_slots: " + err);
        Z = Z_;
        D = Z * rank_radio;
        std::vector<double> invn(K);
        for (int k = 0; k < K; ++k) invn[k] = 1.0 / H.norm_H[k];
        MMW_TRY(d_invn.upload_cast(invn, st));
        MMW_TRY(d_cH.upload_cast(H.cH, st));
        MMW_TRY(eng.resize(D));
        MMW_TRY(Xh.alloc(eng.bs));
        const size_t nnz = (size_t)H.nnzL(), C = (size_t)H.C();
        MMW_TRY(out64.alloc(std::max(std::max(nnz, C), eng.bs)));
        MMW_TRY(stage64.alloc((size_t)K * D));
        return warm ? restart_warm(nit_) : reset(nit_);
    }
    // Warm start of the next probe of the binary search (opt-in; the reference restarts every probe from Y = 1/C, X = I,
    // mmw.py:62-68): the accumulated violations e_accu, the accumulated loss L_accu and the last X / Y are kept, the
    // running sums restart from that X / Y, the iteration counter from zero.
    int restart_warm(int32_t nit_) {
        if (nit_ < 1) return fail(MMW_ERR_ARG, "nit must be >= 1");
        nit = nit_;
        age0 += iter;
        iter = 0;
        emax_enq_iter = emax_iter = -1;
        warm_fresh = true;
        lagged_missed = false;
        age_prev = age_last = -1;
        pending = false;
        chain_ok = false;
        if (eng.viol_d.p) MMW_TRY(eng.clear_violation());
        MMW_TRY(eng.reset_plan_history(true));
        const size_t nnz = (size_t)H.nnzL(), C = (size_t)H.C();
        if (x_tiles) MMW_HIP(hipMemcpyAsync(xs_avg.p, xs_val.p, n_xs * sizeof(T), hipMemcpyDeviceToDevice, st));
        else MMW_HIP(hipMemcpyAsync(xavg.p, xval.p, nnz * sizeof(T), hipMemcpyDeviceToDevice, st));
        MMW_HIP(hipMemcpyAsync(yavg.p, Y.p, C * sizeof(T), hipMemcpyDeviceToDevice, st));
        phase_us.clear();
        phase_samples.clear();
        return MMW_OK;
    }
    int reset(int32_t nit_) override {
        if (host_only) return fail(MMW_ERR_STATE, "this handle was created with device -1 (host pattern only)");
        MMW_HIP(hipSetDevice(device));
        if (nit_ < 1) return fail(MMW_ERR_ARG, "nit must be >= 1");
        nit = nit_;
        iter = 0;
        emax_enq_iter = emax_iter = -1;
        age0 = 0;
        warm_fresh = false;
        age_prev = age_last = -1;
        pending = false;
        chain_ok = false;
        plan_seen = false;
        lagged_missed = false;
        m_guess = 3;
        if (eng.viol_d.p) MMW_TRY(eng.clear_violation());
        MMW_TRY(eng.reset_plan_history(false));
        const size_t nnz = (size_t)H.nnzL(), C = (size_t)H.C();
        MMW_HIP(hipMemsetAsync(lval.p, 0, nnz * sizeof(T), st));
        if (lval_blk.p) MMW_HIP(hipMemsetAsync(lval_blk.p, 0, (size_t)HB.nent * sizeof(T), st));
        if (afrag.p) MMW_HIP(hipMemsetAsync(afrag.p, 0, afrag_n * sizeof(unsigned), st));
        if (afrag16.p) MMW_HIP(hipMemsetAsync(afrag16.p, 0, afrag_n * sizeof(unsigned short), st));
        eng.last_mfma_ok = true;
        lblk_stale = false;
        x_tiles = false;  // the initial point is written in CSR order; the first matrix-core SDDMM call moves it
        MMW_HIP(hipMemsetAsync(xval.p, 0, nnz * sizeof(T), st));
        MMW_HIP(hipMemsetAsync(xavg.p, 0, nnz * sizeof(T), st));
        MMW_HIP(hipMemsetAsync(e_accu.p, 0, C * sizeof(T), st));
        MMW_HIP(hipMemsetAsync(e_this.p, 0, C * sizeof(T), st));
        hipLaunchKernelGGL((k_set_identity<T>), dim3(grid_elems(K)), dim3(BLOCK), 0, st, K, d_diag.p, xval.p, xavg.p);
        const T y0 = (T)(1.0 / (double)C);
        hipLaunchKernelGGL((k_fill<T>), dim3(grid_elems(C)), dim3(BLOCK), 0, st, C, Y.p, y0);
        hipLaunchKernelGGL((k_fill<T>), dim3(grid_elems(C)), dim3(BLOCK), 0, st, C, yavg.p, y0);
        MMW_HIP(hipGetLastError());
        phase_us.clear();
        phase_samples.clear();
        return MMW_OK;
    }

    int record(int slot) {
        if (!timed_iteration()) return MMW_OK;
        if (slot == 0) ev_iter.push_back(iter);
        hipEvent_t e;
        if (!event_pool.empty()) {  // events are kept across runs: creating four per iteration cost the class path ~20 us per iteration
            e = event_pool.back();
            event_pool.pop_back();
        } else
            MMW_HIP(hipEventCreate(&e));
        MMW_HIP(hipEventRecord(e, st));
        events.push_back(e);
        (void)slot;
        return MMW_OK;
    }
    int flush_events() {
        if (events.empty()) return MMW_OK;
        MMW_HIP(hipStreamSynchronize(st));
        for (size_t i = 0; i + 3 < events.size(); i += 4) {
            float a = 0, b = 0, c = 0, t = 0;
            MMW_HIP(hipEventElapsedTime(&a, events[i], events[i + 1]));
            MMW_HIP(hipEventElapsedTime(&b, events[i + 1], events[i + 2]));
            MMW_HIP(hipEventElapsedTime(&c, events[i + 2], events[i + 3]));
            MMW_HIP(hipEventElapsedTime(&t, events[i], events[i + 3]));
            phase_samples.push_back({ev_iter[i / 4], {a * 1e3, b * 1e3, c * 1e3, t * 1e3}});
        }
        for (auto e : events) event_pool.push_back(e);
        events.clear();
        ev_iter.clear();
        // one row per iteration done: its own sample, else the sample of its group of `timing_stride` iterations, else the nearest one
        phase_us.clear();
        if (phase_samples.empty()) return MMW_OK;
        std::sort(phase_samples.begin(), phase_samples.end(), [](const PhaseSample& x, const PhaseSample& y) { return x.it < y.it; });
        const int S = std::max(1, timing_stride);
        for (int i = 0; i < iter; ++i) {
            const int want = (i == 0 || S <= 1) ? i : (i / S) * S + S / 2;
            const PhaseSample* best = nullptr;
            for (const PhaseSample& q : phase_samples) {
                if (q.it == want) { best = &q; break; }
                if (q.it == 0 && i != 0 && phase_samples.size() > 1) continue;  // the first iteration of a run is not like the others
                if (!best || std::abs(q.it - i) < std::abs(best->it - i)) best = &q;
            }
            for (int k = 0; k < 4; ++k) phase_us.push_back(best->us[k]);
        }
        return MMW_OK;
    }

    // CSR-ordered copies of X and its running sum for whoever needs them (API reads, the factor, the gap, the kernels of the other SDDMM
    // forms) while the iterate keeps them in tile order; the tile buffers stay the iterate's
    int x_csr_view() {
        if (!x_tiles) return MMW_OK;
        const size_t nnz = (size_t)H.nnzL();
        hipLaunchKernelGGL((k_x_tiles_to_csr<T>), dim3(grid_elems(nnz)), dim3(BLOCK), 0, st, nnz, b_e2w.p, xs_val.p, xval.p, xs_avg.p, xavg.p);
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    int x_to_csr() {
        MMW_TRY(x_csr_view());
        x_tiles = false;
        return MMW_OK;
    }
    int x_to_tiles() {
        if (x_tiles) return MMW_OK;
        if (!b_e2w.p || n_xs == 0) return fail(MMW_ERR_STATE, "internal: no tile order on this handle");
        const size_t nnz = (size_t)H.nnzL();
        hipLaunchKernelGGL((k_x_csr_to_tiles<T>), dim3(grid_elems(nnz)), dim3(BLOCK), 0, st, nnz, b_e2w.p, xval.p, xs_val.p, xavg.p, xs_avg.p);
        MMW_HIP(hipGetLastError());
        x_tiles = true;
        return MMW_OK;
    }
    int copy_state(bool save) {
        const size_t nnz = (size_t)H.nnzL(), C = (size_t)H.C();
        if (save) sn_tiles = x_tiles;
        else x_tiles = sn_tiles;  // the snapshot goes back into the buffers it was taken from
        DevBuf<T>* snap[6] = {&sn_lval, &sn_xval, &sn_xavg, &sn_Y, &sn_yavg, &sn_eaccu};
        DevBuf<T>* live[6] = {&lval, x_tiles ? &xs_val : &xval, x_tiles ? &xs_avg : &xavg, &Y, &yavg, &e_accu};
        const size_t nx = x_tiles ? n_xs : nnz;
        const size_t len[6] = {nnz, nx, nx, C, C, C};
        CopySet<T> cs;
        for (int i = 0; i < 6; ++i) {
            if (snap[i]->n < len[i]) MMW_TRY(snap[i]->alloc(len[i]));
            cs.dst[i] = save ? snap[i]->p : live[i]->p;
            cs.src[i] = save ? live[i]->p : snap[i]->p;
            cs.n[i] = len[i];
        }
        if (sn_plan.n < 1) MMW_TRY(sn_plan.alloc(1));
        cs.plan_dst = save ? sn_plan.p : eng.plan_d.p;
        cs.plan_src = save ? eng.plan_d.p : sn_plan.p;
        hipLaunchKernelGGL((k_copy_state<T>), dim3(256, 6), dim3(BLOCK), 0, st, cs);  // one launch instead of seven copies
        MMW_HIP(hipGetLastError());
        if (!save && lval_blk.p) lblk_stale = true;  // rebuilt from the restored values when the fp32 kernel next needs it
        if (!save && afrag.p) {  // the fragment image follows the restored values
            hipLaunchKernelGGL((k_refrag<T>), dim3(grid_elems(nnz)), dim3(BLOCK), 0, st, nnz, lval.p, b_fpos.p, afrag.p);
            MMW_HIP(hipGetLastError());
        }
        return MMW_OK;
    }
    // a batch enqueued without plan readbacks is verified here; a violated batch is replayed synchronously
    // A chunk that ends with a violation is discarded and run again from its snapshot.  The second attempt is still a chunk without
    // readbacks, but a cautious one: Lanczos steps (no first-order form) at the a-priori order plus one, an exact plan in front of every
    // exponential (no extrapolated ones), the softmax in its two passes for the first iteration.  Only if that one is refused as well --
    // or the matrix has outgrown the 16-bit split of the matrix-core product, which only the per-iteration readback steps around --
    // do the iterations run synchronously (~2.5x the time per iteration).  Hard probes of a bisection (slot counts at the edge of
    // feasibility: the matrix grows faster than any history predicts) took 3 replays per 150 iterations, 29 ms instead of 13.
    int restore_pending() {
        chain_ok = false;
        emax_enq_iter = emax_iter = -1;  // (the reduction enqueued behind the discarded chunk saw its e_this)
        MMW_TRY(eng.clear_violation());
        MMW_TRY(copy_state(false));
        iter = pend_iter0;
        if (timing) {  // drop the timers of the discarded chunk (earlier chunks keep theirs)
            MMW_HIP(hipStreamSynchronize(st));
            for (size_t i = pend_events0; i < events.size(); ++i) event_pool.push_back(events[i]);
            events.resize(std::min(events.size(), pend_events0));
            ev_iter.resize(events.size() / 4);
        }
        return MMW_OK;
    }
    void say_replay(int viol, const char* how) const {
        if (!getenv("MMW_VERBOSE")) return;
        const ExpmPlan& p = eng.last;
        union { unsigned u; float f; } c1, fe;
        c1.u = p.conv[std::max(0, std::min(p.m_eff, MAX_ORDER))]; fe.u = p.first_est;
        fprintf(stderr, "[replay] iterations %d..%d (Z %d) %s: reason bits %d (1 order, 2 lagged plan, 4 plan, 8 softmax, 16 operands, 32 first-order certificate); launched m %d, first-order %d, one-half %d; "
                        "plan m %d m_eff %d apriori %d rho %.3g absn %.3g est %.2e first_est %.2e tol %.1e\n",
                pend_iter0, pend_iter0 + pend_n - 1, (int)H.Z, how, viol, m_guess, (int)first_guess, (int)first_a16_guess, p.m, p.m_eff, p.m_apriori, p.rho, p.absn, (double)c1.f, (double)fe.f, p.tol);
    }
    int settle() {
        if (!pending) return MMW_OK;
        pending = false;
        int viol = 0;
        MMW_TRY(eng.fetch_plan(&viol));
        if (!viol) {
            plan_seen = true;
            note_plan();
            m_guess = next_launch_order();
            return MMW_OK;
        }
        ++replays;
        const bool operands_only = (viol & VIOL_OPERANDS) && !first_guess;  // the bf16 split's gate: the fp32 kernel has to take over
        if (viol & VIOL_LAGGED) lagged_missed = true;  // (er-50k: a second chunk missed the same way 32 iterations later)
        say_replay(viol, "discarded");
        MMW_TRY(restore_pending());
        if (cautious_replay && !operands_only && eng.method == MMW_EXPM_LANCZOS && pend_n > 1) {
            first_guess = false;
            first_a16_guess = false;
            m_guess = std::min(eng.max_order, std::max(std::max(eng.last.m_apriori, eng.last.m_eff), m_guess) + 1);
            exact_plans_only = true;
            const int rc = iterate_impl(pend_n, nullptr, pend_seed, true);
            exact_plans_only = false;
            MMW_TRY(rc);
            viol = 0;
            MMW_TRY(eng.fetch_plan(&viol));
            if (!viol) {
                plan_seen = true;
                note_plan();
                m_guess = next_launch_order();
                return MMW_OK;
            }
            ++replays;
            say_replay(viol, "cautious attempt discarded");
            MMW_TRY(restore_pending());
        }
        return iterate_impl(pend_n, nullptr, pend_seed, false);
    }
    int iterate(int32_t n, const double* randv, uint64_t seed) override {
        if (host_only) return fail(MMW_ERR_STATE, "this handle was created with device -1 (host pattern only)");
        MMW_HIP(hipSetDevice(device));
        if (n < 0) return fail(MMW_ERR_ARG, "n must be >= 0");
        MMW_TRY(settle());
        if (iter + n > nit) return fail(MMW_ERR_STATE, "mmw_iterate: more iterations than announced to mmw_create/mmw_reset");
        const bool optimistic = randv == nullptr && n > 1 && !kt_exact() && !getenv("MMW_SYNC_PLAN");  // profiling mode 1 counts exact launches
        if (!optimistic) {
            chain_ok = false;
            MMW_TRY(iterate_impl(n, randv, seed, false));
            return enqueue_emax();
        }
        // Chunks enqueued without plan readbacks.  Each chunk starts from a device snapshot; before the next one starts the
        // plan of the previous is looked at (one sync): a chunk that needed more steps than were launched is restored and
        // replayed with per-iteration readback, and the launch order follows the device.  The last chunk is settled by the
        // next call.  Chunks are short while L still grows fast (its norm is proportional to the iteration count): half as many iterations as have
        // been done (4 ... 32); as many as have been done (8 ... 32) while one Lanczos step is accepted with a factor 2 to spare.
        int left = n;
        while (left > 0) {
            MMW_TRY(settle());
            // ... or as many as the last settled plan's estimate leaves room for (room_iterations)
            int cap = lagged_ok() ? std::max(8, std::min(32, age())) : std::max(4, std::min(32, age() / 2));
            // (holding the chunk to the run's age until two plans have shown how fast the matrix grows would spare the hard probes of a
            // bisection one discarded chunk -- at slot counts near infeasibility the norm grew 16x over iterations 4..35, not the 9x of a
            // linear law -- but costs every run one more readback in its first 32 iterations: measured, not kept)
            if (chain_ok && age() >= 4) cap = std::max(cap, std::min(32, room_iterations()));
            if (warm_fresh) cap = 8;
            int chunk = std::min(left, cap);
            if (left - chunk == 1) ++chunk;  // no trailing chunk of one iteration: it would run synchronously and break the chain of chunks
            if (plan_seen) m_guess = next_launch_order(chunk);  // before the first readback of a run: the default set by reset()
            first_guess = plan_seen && first_order_ok(chunk);  // (requires that the last plan read back stopped after one step)
            if (first_guess) m_guess = 1;
            first_a16_guess = first_guess && first_a16_enabled && afrag16.p != nullptr && first_order_ok(chunk, true);
            if (warm_fresh) {  // the plan at hand belongs to the previous probe's slot count: one spare step, no first-order form
                m_guess = std::min(eng.max_order, std::max(2, eng.last.m_eff + 1));
                first_guess = false;
            }
            MMW_TRY(copy_state(true));
            pend_iter0 = iter; pend_n = chunk; pend_seed = seed; pend_events0 = events.size();
            MMW_TRY(iterate_impl(chunk, nullptr, seed, chunk > 1));
            warm_fresh = false;
            pending = chunk > 1;
            left -= chunk;
        }
        return enqueue_emax();
    }
    // The objective record's one number -- the largest violation of the last iteration (MMW_F_E_MAX) -- is reduced right behind the call's
    // work and copied out by the mmw_sync that waits for it anyway: reading it afterwards is free (its launch + copy + wait were a third
    // of what a 20-step timed region spends on its record).  A replay of the last chunk changes `iter` back and forth but ends at the same
    // e_this only after re-running, so the value is tied to the iteration count AND dropped whenever a chunk is discarded.
    int enqueue_emax() {
        if (iter <= 0) return MMW_OK;
        if (!emax_d.p) MMW_TRY(emax_d.alloc(1));
        hipLaunchKernelGGL((k_max_of<T>), dim3(1), dim3(1024), 0, st, (size_t)H.C(), e_this.p, emax_d.p);
        MMW_HIP(hipGetLastError());
        emax_enq_iter = iter;
        emax_iter = -1;
        return MMW_OK;
    }
    // Steps to launch without reading the plan back: what the last application used, plus one spare step unless its
    // estimate met the tolerance with a factor 8 to spare (L grows by a fraction of itself per iteration; the plan is looked
    // at every 16 iterations; a batch that needs more anyway is replayed from its snapshot).
    // `ahead`: iterations the launch order has to hold for (the coming chunk).  The estimate after m steps grows like ||L||^(2m) and
    // ||L|| like the iteration count: no spare step only if the estimate, grown over the chunk, still meets the tolerance with a
    // factor 2 (and never without the factor 8 at the moment of the readback).
    // The coming chunk of `ahead` iterations may take the exponential as ONE product, y = u + (L/2 - mu I) u (ExpmEngine::apply_first):
    // the last plan read back holds the bound that form would have met (first_est, from k_lz_scalars or from the form's own check); it
    // grows like rho * q ~ t^2, and the same margins as for dropping the spare Lanczos step apply.
    // The certificate (kernels_mfma.h, first_verify) adds to that truncation bound what the fp16 operands lose: the plane of u at its
    // measured rounding (F16_PLANE_EXPECT predicts it) and the matrix image at c_A (one fp16 half: 2^-11; hi + lo: 2^-21), both times the
    // row-sum bound absn, which grows linearly.  a16: the chunk would read the matrix as one half.
    bool first_order_ok(int ahead, bool a16 = false) const {
        const ExpmPlan& p = eng.last;
        if (!first_enabled || sizeof(T) != 4 || !p.apost || p.m_eff != 1 || p.first_est == 0u) return false;
        union { unsigned u; float f; } e;
        e.u = p.first_est;
        const double g1 = growth_ratio(ahead), absn_g = p.absn * g1;
        if (!(absn_g < 0.03)) return false;  // the entries times 2^20 stay inside fp16's range
        if (a16 ? !((F16_PLANE_EXPECT + F16_UNIT) * absn_g <= p.tol) : !(F16_PLANE_EXPECT * absn_g <= p.tol)) return false;  // ExpmPlan::f16a_ok / f16_ok over the chunk
        // every iteration of the form is certified (a miss costs a replay of the chunk, nothing else): the truncation bound has to meet the
        // tolerance with a factor 4 now, and the whole predicted bound with a tenth to spare after the growth over the chunk
        const double rounding = std::exp(p.rho * g1) * (absn_g * (plane_rounding() + (a16 ? F16_UNIT : F16_CA_TWO)) + F16_SUBNORMAL_ROW);
        return (double)e.f <= p.tol / 4.0 && (double)e.f * g1 * g1 + rounding <= 0.9 * p.tol;
    }
    int next_launch_order(int ahead = 0) const {
        const ExpmPlan& p = eng.last;
        if (p.m_eff <= 0) return std::min(eng.max_order, p.m + 1);
        int spare = 1;
        if (p.apost && p.m_eff >= 1 && p.m_eff <= MAX_ORDER) {
            union { unsigned u; float f; } e;
            e.u = p.conv[p.m_eff];
            const double grow = std::pow(growth_ratio(ahead), 2.0 * p.m_eff);
            if ((double)e.f <= p.tol / 8.0 && (double)e.f * grow <= p.tol / 2.0) spare = 0;
        }
        if (p.m_eff >= p.m_apriori) spare = 0;  // the a-priori order is never exceeded
        return std::min(eng.max_order, p.m_eff + spare);
    }
    // How many more iterations one Lanczos step should stay accepted: its error estimate grows about quadratically with the norm of
    // L, which grows linearly with the iteration count, so est(t + c) ~ est(t) ((t + c) / t)^2 <= tol gives c <= t (sqrt(tol / est) - 1);
    // half of that.  0 unless the last plan read back stopped after one step.
    int room_iterations() const {
        const ExpmPlan& p = eng.last;
        if (!p.apost || p.m_eff != 1) return 0;
        union { unsigned u; float f; } e;
        e.u = p.conv[1];
        if (!((double)e.f > 0.0)) return 32;
        double c = 0.5 * (double)age() * (std::sqrt(p.tol / (double)e.f) - 1.0);
        if (age_prev >= 0 && age_last > age_prev && rho_last > rho_prev && rho_last > 0.0)  // ... or with the slope the last two plans showed (growth_ratio)
            c = std::min(c, 0.5 * (std::sqrt(p.tol / (double)e.f) - 1.0) * rho_last * (double)(age_last - age_prev) / (1.5 * (rho_last - rho_prev)));
        return c > 32.0 ? 32 : (c < 0.0 ? 0 : (int)c);
    }
    // the last plan read back accepted ONE Lanczos step with a factor 8 to spare (where the order is already rising -- the graphs
    // without locality -- a long chunk launched with too few stages is a long replay: measured 5 187 -> 2 686 it/s at er-5pct-2k)
    bool plan_has_room() const {
        const ExpmPlan& p = eng.last;
        if (!p.apost || p.m_eff != 1) return false;
        union { unsigned u; float f; } e;
        e.u = p.conv[1];
        return (double)e.f <= p.tol / 8.0;
    }
    // Lagged planning pays where one Lanczos step is accepted with room to spare (its extrapolated norm bound is ~1/t larger than
    // the exact one, which must not cost a second product: on graphs without locality a product is 10x the two kernels saved).
    bool lagged_ok() const {
        const ExpmPlan& p = eng.last;
        if (!lagged_plan || lagged_missed || !p.apost || p.m_eff != 1) return false;
        union { unsigned u; float f; } e;
        e.u = p.conv[1];
        return (double)e.f <= p.tol / 2.0;
    }
    int sketch_slabs() const { static const int cap = getenv("MMW_SK_SLABS") ? atoi(getenv("MMW_SK_SLABS")) : 256; return std::min(grid_rows(K), cap); }  // few slabs for the start-norm reduction
    int launch_sketch(hipStream_t s, uint64_t seed, uint32_t it, bool planes_f16 = false) {
        const bool lz = eng.method == MMW_EXPM_LANCZOS;
        const int Dpad = eng.lay.Dpad;
        unsigned short* pl = eng.start_planes();
        hipLaunchKernelGGL((k_sketch_rng<T>), dim3(sketch_slabs()), dim3(BLOCK), lz ? (size_t)(WAVES_PER_BLOCK + 1) * Dpad * sizeof(double) : 0, s, K, D, Dpad,
                           seed, it, eng.start_block(), lz ? eng.partial_sq.p : (double*)nullptr, pl, planes_f16 ? 1 : 0,
                           planes_f16 && lz && fv_measure ? eng.partial_du.p : (double*)nullptr);
        eng.planes_ready[0] = pl != nullptr;
        eng.planes0_f16 = planes_f16 && pl != nullptr;
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    int iterate_impl(int32_t n, const double* randv, uint64_t seed, bool optimistic) {
        // X on the pattern comes from the matrix-core SDDMM in this call (decided below, per iteration, by the same expression): it
        // writes -- and the DUAL phase then reads -- X in tile order; every other SDDMM form works on the CSR order
        const bool sd_mf_call = sizeof(T) == 4 && sddmm_mfma && eng.use_blk && eng.method == MMW_EXPM_LANCZOS && (eng.lay.Dpad % 32) == 0 && b_e2w.p != nullptr;
        if (n > 0) MMW_TRY(sd_mf_call ? x_to_tiles() : x_to_csr());
        const PatternDev<T> P = pat();
        const T* const xcur = x_tiles ? xs_val.p : xval.p;  // the X the DUAL phase reads (the layout does not change inside a call)
        const int gr = grid_rows(K);
        // the DUAL pass's grid: its workgroups stride over the row pairs, and the slabs it leaves (maxima, softmax sums, |L| row sums) are
        // folded by one workgroup afterwards.  One resident round of workgroups (five per CU at the pass's 86 registers) instead of one per
        // eight rows: half the slabs to fold and no second round's tail -- DUAL 21.0 -> 20.0 us per step at the benchmark (640: 22.3; 1920: 20.2)
        static const int dual_cap = getenv("MMW_DUAL_GRID") ? atoi(getenv("MMW_DUAL_GRID")) : 0;
        const int gd = std::min(gr, dual_cap > 0 ? dual_cap : 5 * device_cus());
        const int C = (int)H.C();
        const int gc = grid_elems((size_t)C);
        static const int loss_grid_cap = getenv("MMW_LOSS_GRID") ? atoi(getenv("MMW_LOSS_GRID")) : LOSS_GRID_MAX;
        const int gl = (int)std::min<size_t>(((size_t)H.nnzL() + BLOCK - 1) / BLOCK, (size_t)std::max(1, std::min(loss_grid_cap, LOSS_GRID_MAX)));  // LOSS: one thread per stored entry, grid-stride
        const int Dpad = eng.lay.Dpad;
        int m_launch = optimistic ? m_guess : 0;
        // from the plan the last settled chunk ended on; not in the first chunk after a warm restart: with another slot count the matrix grows
        // at another rate than the history the extrapolation rests on (its bound was missed and the chunk replayed, measured)
        const bool lag_chunk = optimistic && lagged_ok() && !warm_fresh && !exact_plans_only;
        const bool chain = optimistic && chain_ok;          // this chunk continues the previous one (see chain_ok)
        chain_ok = false;
        // the plan is chained only while a single step is accepted with a factor 8 to spare: near a change of order an exact plan at
        // the start of every chunk keeps the a-priori order down (er-1pct: 5 127 it/s with it, 4 495 without)
        const bool chain_plan = chain && plan_has_room();
        bool xavg_deferred = false;
        // the row sums of X the DUAL phase starts from: left by the last matrix-core SDDMM (this call's previous iteration, or the chunk
        // this one continues), otherwise taken by k_dual_rows
        bool rs_ok = chain && rs_last && rs_enabled && rsfx.p != nullptr;
        FirstVerify fv_pending;  // the first-order exponential of the previous iteration of this call still waits for its check
        rs_last = false;
        // drawing the next sketch in extra workgroups of the SDDMM launch paid off with 8-wave SDDMM workgroups (+3.7 %); with
        // 16-wave ones (two per CU, every wave slot taken) it costs 1.5 %, so it is opt-in
        const bool fuse_sketch = !kt_exact() && !timing && getenv("MMW_FUSED_SKETCH") != nullptr;
        sketch_done_for = -1;  // whatever an earlier batch left in the start block is not trusted
        for (int it = 0; it < n; ++it) {
            const int acc = (iter + 1 < nit) ? 1 : 0;  // the last X / Y are not averaged (mmw.py:77-78,203)
            MMW_TRY(record(0));
            // ---- DUAL
            MMW_TRY(kt.begin(KT_DUAL));
            const long long* rs_it = rs_ok ? rsfx.p : nullptr;
            const FirstVerify fv = fv_pending;
            fv_pending = FirstVerify{};
            if (rs_it) ++n_rs_iters;
            if (!rs_it) hipLaunchKernelGGL((k_dual_rows<T>), dim3(gr), dim3(BLOCK), 0, st, P, xcur, rsum.p, e_this.p);
            // Lagged planning inside a chunk (not the first iteration of a run, a replay or after a change of the iterate, which plan exactly): k_dual_h also takes the row sums of the
            // L it walks over anyway -- last iteration's -- and one extra workgroup of k_softmax_b turns them into this iteration's plan
            // (extrapolated bounds, checked by the next plan): k_rowsums + k_plan leave the critical path.
            const bool lagged_it = optimistic && (it > 0 || chain_plan) && eng.method == MMW_EXPM_LANCZOS && lag_chunk;
            PlanArgs pa;
            if (lagged_it) {
                pa.plan = eng.plan_d.p; pa.part = eng.row_part.p; pa.viol = eng.viol_d.p; pa.tol = eng.tol; pa.K = K; pa.method = eng.method;
                pa.max_order = eng.max_order; pa.np = gd; pa.m_launch = m_launch; pa.apost = eng.apost() ? 1 : 0; pa.iter_seen = age() - 1;
            }
            // Inside a chunk (not its first iteration) the softmax rides in k_dual_h, shifted by the previous iteration's maximum
            // instead of this one's: one small workgroup then folds the sums, and the LOSS pass normalises where it reads
            // (kernels_loop.h, k_dual_h / k_dual_scal).  Two launches of the dependent chain fewer.
            const bool fused_dual = optimistic && (it > 0 || chain) && fuse_dual;
            if (fused_dual) {
                ++n_fused_iters;
                if (yun.n < (size_t)C) MMW_TRY(yun.alloc((size_t)C));
                // MMW_DUAL_STAMPS=1 (developer aid): per-wave phase clocks of the last iteration's launch, printed to stderr
                DevBuf<unsigned long long> dh_stamps;
                const bool want_dst = it + 1 == n && getenv("MMW_DUAL_STAMPS") != nullptr;
                if (want_dst) {
                    MMW_TRY(dh_stamps.alloc((size_t)gd * WAVES_PER_BLOCK * 8));
                    MMW_HIP(hipMemsetAsync(dh_stamps.p, 0, (size_t)gd * WAVES_PER_BLOCK * 8 * sizeof(unsigned long long), st));
                }
                hipLaunchKernelGGL((k_dual_h<T>), dim3(gd), dim3(BLOCK), 0, st, P, rsum.p, e_this.p, e_accu.p, eta, max_part.p,
                                   (const T*)(lagged_it ? lval.p : nullptr), 0.5, eng.row_part.p, (const double*)(scal.p + 4), yun.p, wH.p, sum_part.p,
                                   rs_it, xcur, FirstVerify{}, dh_stamps.p);
                if (want_dst) {
                    std::vector<unsigned long long> h((size_t)gd * WAVES_PER_BLOCK * 8);
                    MMW_HIP(hipMemcpyAsync(h.data(), dh_stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                    MMW_HIP(hipStreamSynchronize(st));
                    double sum[5] = {0}, slow[5] = {0};
                    std::vector<std::pair<unsigned long long, size_t>> byl;
                    int nw = 0;
                    for (size_t w = 0; w < h.size() / 8; ++w) {
                        const unsigned long long* q = &h[w * 8];
                        if (!q[5]) continue;
                        ++nw;
                        const unsigned long long p1 = q[1] ? q[1] : q[0], p2 = q[2] ? q[2] : p1, p3 = q[3], p4 = q[4];
                        sum[0] += (double)(p1 - q[0]); sum[1] += (double)(p2 - p1); sum[2] += (double)(p3 - p2); sum[3] += (double)(p4 - p3); sum[4] += (double)(q[5] - p4);
                        byl.push_back({q[5] - q[0], w});
                    }
                    std::sort(byl.rbegin(), byl.rend());
                    const size_t top = std::max<size_t>(1, byl.size() / 20);
                    for (size_t i = 0; i < top && i < byl.size(); ++i) {
                        const unsigned long long* q = &h[byl[i].second * 8];
                        const unsigned long long p1 = q[1] ? q[1] : q[0], p2 = q[2] ? q[2] : p1;
                        slow[0] += (double)(p1 - q[0]); slow[1] += (double)(p2 - p1); slow[2] += (double)(q[3] - p2); slow[3] += (double)(q[4] - q[3]); slow[4] += (double)(q[5] - q[4]);
                    }
                    if (nw)
                        fprintf(stderr, "[dual stamps] %d workgroups, %d waves; clocks per wave: row pointers %.0f, rows (entries + gathers + sums) %.0f, rows' tails %.0f, violation part %.0f, fold + stores %.0f; "
                                        "slowest twentieth: %.0f / %.0f / %.0f / %.0f / %.0f\n",  // (the counters of different XCDs share no origin: no launch-wide span)
                                gd, nw, sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw, slow[0] / top, slow[1] / top, slow[2] / top, slow[3] / top, slow[4] / top);
                }
                hipLaunchKernelGGL(k_dual_scal, dim3(1 + fv.nwg), dim3(DSCAL_THREADS), 0, st, sum_part.p, max_part.p, gd, scal.p,
                                   dual_gap, eng.viol_d.p, fv);
            } else {
                hipLaunchKernelGGL((k_dual_h<T>), dim3(gd + fv.nwg), dim3(BLOCK), 0, st, P, rsum.p, e_this.p, e_accu.p, eta, max_part.p,
                                   (const T*)(lagged_it ? lval.p : nullptr), 0.5, eng.row_part.p, (const double*)nullptr, (T*)nullptr, (T*)nullptr,
                                   (double*)nullptr, rs_it, xcur, fv);
                hipLaunchKernelGGL((k_softmax_a<T>), dim3(gc), dim3(BLOCK), 0, st, P, e_accu.p, Y.p, max_part.p, gd, sum_part.p);
                hipLaunchKernelGGL((k_softmax_b<T>), dim3(gc + (lagged_it ? 1 : 0)), dim3(BLOCK), 0, st, C, Y.p, yavg.p, acc, sum_part.p, gc, scal.p,
                                   K + (int)H.E_asso(), d_invn.p, wH.p, pa, max_part.p, gd);
            }
            MMW_TRY(kt.end());
            MMW_TRY(record(1));
            // ---- LOSS
            MMW_TRY(kt.begin(KT_LOSS));
            // the X of the previous iteration of this chunk is added to the running sum inside this pass (xavg_deferred), and
            // this iteration's sketch is drawn by leading workgroups of the same launch (VALU work under a memory-bound pass)
            const bool rs_zeroed = rs_enabled && rsfx.p != nullptr && sddmm_mfma;  // the coming SDDMM may add its row sums to zeroed totals
            // the exponential of this iteration as one first-order product (decided per chunk, first_order_ok)
            const bool sketch_have = !randv && sketch_done_for == (int64_t)iter && sketch_done_seed == seed;
            // (a sketch an earlier launch already drew came without the fp16 plane and the measure of its rounding: no first-order form then)
            const bool first_it = optimistic && first_guess && m_launch == 1 && !randv && rs_zeroed && sizeof(T) == 4 && eng.mfma_now() &&
                                  eng.method == MMW_EXPM_LANCZOS && eng.use_blk && (Dpad % 32) == 0 && !sketch_have;
            SketchArgs<T> skl{};
            const bool lz_m = eng.method == MMW_EXPM_LANCZOS;
            // (with the per-iteration phase events of mmw_set_timing on as well: the draw then counts into the LOSS phase's microseconds
            // instead of the exponential's -- the reference draws inside mmw.py:172-181 -- and the iteration's total is unchanged; a launch of
            // its own cost the class path 14 us per iteration)
            if (!randv && !sketch_have && !kt_exact() && !getenv("MMW_NO_LOSS_SKETCH")) {
                skl.nblocks = sketch_slabs(); skl.K = K; skl.D = D; skl.seed = seed; skl.iter = (uint32_t)iter;
                skl.R = eng.start_block();
                skl.colsq_part = lz_m ? eng.partial_sq.p : nullptr;
                skl.planes = eng.start_planes();
                skl.planes_f16 = first_it ? 1 : 0;
                skl.dusq_part = first_it && fv_measure ? eng.partial_du.p : nullptr;
                eng.planes_ready[0] = skl.planes != nullptr;
                eng.planes0_f16 = first_it && skl.planes != nullptr;
                sketch_done_for = (int64_t)iter; sketch_done_seed = seed; sketch_done_slabs = skl.nblocks;
            }
            // the blocked copy of L feeds the fp32 LDS kernel only: while the matrix-core kernel runs the products it is left stale
            const bool mf_it = eng.mfma_now() && eng.method == MMW_EXPM_LANCZOS;
            if (mf_it) lblk_stale = true;
            const PlanArgs pl_loss = fused_dual ? pa : PlanArgs{};  // the fused pass has no softmax pass B to lend the planning a workgroup
            hipLaunchKernelGGL((k_loss<T>), dim3(gl + skl.nblocks + (pl_loss.plan ? 1 : 0)), dim3(BLOCK), skl.nblocks && lz_m ? (size_t)(WAVES_PER_BLOCK + 1) * Dpad * sizeof(double) : 0,
                               st, P, d_lrow.p, fused_dual ? yun.p : Y.p, wH.p, scal.p, lval.p, eta,
                               (const int*)(eng.use_blk && !mf_it ? b_bpos.p : nullptr), lval_blk.p,
                               (const T*)(xavg_deferred ? xval.p : nullptr), xavg_deferred ? xavg.p : (T*)nullptr, skl, Dpad,
                               (const int*)(eng.use_mfma ? b_fpos.p : nullptr), afrag.p, fused_dual ? Y.p : (T*)nullptr, yavg.p, acc, pl_loss,
                               rs_zeroed ? rsfx.p : (long long*)nullptr, rs_zeroed ? (first_it ? 2 * K : K) : 0, first_it ? 1 : 0,
                               first_it && first_a16_guess ? afrag16.p : (unsigned short*)nullptr);
            xavg_deferred = false;
            MMW_TRY(kt.end());
            MMW_TRY(record(2));
            // ---- EXPM + X on the pattern
            const bool sketch_rode = !randv && sketch_done_for == (int64_t)iter && sketch_done_seed == seed;  // nothing to launch, nothing to time
            if (!sketch_rode) MMW_TRY(kt.begin(KT_SKETCH));
            if (randv) {
                MMW_TRY(copy_h2d(stage64.p, randv + (size_t)it * K * D, (size_t)K * D * sizeof(double), st));  // page-locked staging (runtime.h)
                hipLaunchKernelGGL((k_import_block<T>), dim3(grid_elems(eng.bs)), dim3(BLOCK), 0, st, K, D, Dpad, stage64.p, eng.start_block());
                eng.planes_ready[0] = false;  // an uploaded sketch is split by a pass of its own
                last_was_rng = false;
            } else {
                const bool have = sketch_done_for == (int64_t)iter && sketch_done_seed == seed;  // drawn by the previous SDDMM launch
                if (!have) MMW_TRY(launch_sketch(st, seed, (uint32_t)iter, first_it));
                eng.start_colsq_ready = eng.method == MMW_EXPM_LANCZOS;  // the Lanczos start norms come out of the sketch kernel
                eng.npart_start = have ? sketch_done_slabs : sketch_slabs();
                sketch_done_for = -1;
                last_was_rng = true;
                last_seed = seed;
            }
            if (!sketch_rode) MMW_TRY(kt.end());
            MMW_HIP(hipGetLastError());
            // X on the pattern runs on the matrix cores too when the exponential did: the combination then also writes y's planes
            const bool sd_mf = sd_mf_call;
            eng.out_planes = nullptr;
            if constexpr (sizeof(T) == 4) {
                if (sd_mf) {
                    if (xh_planes.n < 2 * eng.bs) MMW_TRY(xh_planes.alloc(2 * eng.bs));
                    eng.out_planes = xh_planes.p;
                }
            }
            // inside a chunk only the SDDMM reads X_half; the chunk's last iteration leaves the fp32 copy the API hands out
            eng.planes_only = eng.out_planes != nullptr && optimistic && it + 1 < n && !getenv("MMW_KEEP_XHALF");
            eng.rownorm_d = drow.p;  // the Lanczos combination also emits the row norms and the trace slabs
            eng.rownorm_part = tr_part.p;
            eng.plan_iter = age();
            int ntr1 = 0;
            if (first_it) {
                const size_t need = (size_t)eng.first_grid_max();
                if (tr1_part.n < need) {
                    MMW_TRY(tr1_part.alloc(need));
                    MMW_HIP(hipMemsetAsync(tr1_part.p, 0, need * sizeof(double), st));
                }
                MMW_TRY(eng.apply_first(eng.planes_only ? (T*)nullptr : Xh.p, 0.5, m_launch, lagged_it, xh_planes.p, rsfx.p + K, tr1_part.p, &ntr1,
                                        first_a16_guess ? afrag16.p : (const unsigned short*)nullptr));
                ++n_first_iters;
                if (first_a16_guess) ++n_first16_iters;
            } else
                MMW_TRY(eng.apply(Xh.p, 0.5, m_launch, lagged_it));
            MMW_TRY(kt.begin(KT_SDDMM));
            if (eng.method != MMW_EXPM_LANCZOS)
                hipLaunchKernelGGL((k_rownorm2<T>), dim3(gr), dim3(BLOCK), 0, st, K, Dpad, Xh.p, drow.p, tr_part.p);
            bool sd_done = false;
            rs_ok = false;
            if constexpr (sizeof(T) == 4) {
                if (sd_mf) {
                    SdMfmaDev SM;
                    SM.tbase = b_tbase.p; SM.tptr = b_tptr.p; SM.trc = b_trc.p; SM.nedges = (int)HB.m_nedges;
                    const long long* dfx = first_it ? rsfx.p + K : nullptr;
                    const double* trp = first_it ? tr1_part.p : tr_part.p;
                    const int ntr = first_it ? ntr1 : gr;
                    // the certificate of this iteration's first-order exponential rides in this launch (8 more columns of workgroups)
                    FirstVerify fv_now;
                    if (first_it) {
                        fv_now.plan = eng.plan_d.p; fv_now.viol = eng.viol_d.p; fv_now.o2 = eng.partial_o2.p; fv_now.n_o2 = eng.mf.nb;
                        fv_now.u2 = eng.partial_sq.p; fv_now.du2 = fv_measure ? eng.partial_du.p : nullptr; fv_now.rows = K; fv_now.n_u2 = eng.npart_start; fv_now.Dpad = Dpad;
                        fv_now.nwg = Dpad / FV_COLS;
                        fv_now.cA = first_a16_guess ? F16_UNIT : F16_CA_TWO;
                        fv_now.du_scale = fv_du_scale;
                    }
                    const bool fv_rides = first_it && fv_in_sddmm;
                    const dim3 grid((HB.nbm() + 7) / 8 * 8 + (fv_rides ? 8 : 0), (HB.m_ntile_max + SDM_GT - 1) / SDM_GT);
                    SM.tmask = b_tmask.p;
                    long long* rs_out = rs_zeroed ? rsfx.p : nullptr;  // this iteration's LOSS pass zeroed the totals
                    // MMW_SD_STAMPS=1 (developer aid): per-wave phase clocks of the last iteration's launch, printed to stderr
                    DevBuf<unsigned long long> sdm_stamps;
                    const size_t n_st = (size_t)grid.x * grid.y * 16 * 8;
                    const bool want_st = it + 1 == n && getenv("MMW_SD_STAMPS") != nullptr;
                    if (want_st) {
                        MMW_TRY(sdm_stamps.alloc(n_st));
                        MMW_HIP(hipMemsetAsync(sdm_stamps.p, 0, n_st * sizeof(unsigned long long), st));
                    }
                    static const int sd_nb = getenv("MMW_SD_NB") ? atoi(getenv("MMW_SD_NB")) : 2;  // chunks resident per workgroup (3: measured 1 % slower)
#define MMW_SDM_LAUNCH(MT, NB)                                                                                                               \
    do {                                                                                                                                     \
        MMW_TRY(set_max_lds(reinterpret_cast<const void*>(&k_sddmm_mfma<MT, NB>), sdm_lds_bytes<MT, NB>()));                                 \
        hipLaunchKernelGGL((k_sddmm_mfma<MT, NB>), grid, dim3(256 * MT), (sdm_lds_bytes<MT, NB>()), st, eng.mf, SM, K, Dpad,                 \
                           reinterpret_cast<const char*>(xh_planes.p), drow.p, trp, ntr, xs_val.p, xs_avg.p, acc, rs_out, dfx, sdm_stamps.p,  \
                           fv_rides ? fv_now : FirstVerify{});                                                                               \
    } while (0)
                    if (HB.mfma_mt == 2) { if (sd_nb == 3) MMW_SDM_LAUNCH(2, 3); else MMW_SDM_LAUNCH(2, 2); }
                    else { if (sd_nb == 3) MMW_SDM_LAUNCH(1, 3); else MMW_SDM_LAUNCH(1, 2); }
#undef MMW_SDM_LAUNCH
                    if (want_st) {
                        std::vector<unsigned long long> h(n_st);
                        MMW_HIP(hipMemcpyAsync(h.data(), sdm_stamps.p, n_st * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                        MMW_HIP(hipStreamSynchronize(st));
                        double sum[8] = {0}, life_max = 0;
                        int nw = 0;
                        for (size_t w = 0; w < n_st / 8; ++w) {
                            const unsigned long long* q = &h[w * 8];
                            if (!q[4]) continue;
                            ++nw;
                            for (int k = 0; k < 8; ++k) sum[k] += (double)q[k];
                            life_max = std::max(life_max, (double)q[4]);
                        }
                        {   // the slowest twentieth of the waves: where their time went
                            std::vector<std::pair<unsigned long long, size_t>> byl;
                            for (size_t w = 0; w < n_st / 8; ++w) if (h[w * 8 + 4]) byl.push_back({h[w * 8 + 4], w});
                            std::sort(byl.rbegin(), byl.rend());
                            const size_t top = std::max<size_t>(1, byl.size() / 20);
                            double ts[8] = {0};
                            for (size_t i = 0; i < top && i < byl.size(); ++i) for (int k = 0; k < 8; ++k) ts[k] += (double)h[byl[i].second * 8 + k];
                            if (!byl.empty())
                                fprintf(stderr, "[sddmm stamps] slowest %zu waves: prologue %.0f, wait+barrier %.0f, issue %.0f, reads+products %.0f, sums %.0f, stores %.0f, lifetime %.0f; by (wg.y): ", top,
                                        ts[0] / top, ts[1] / top, ts[2] / top, ts[3] / top, ts[5] / top, ts[7] / top, ts[4] / top);
                            int cnt[8] = {0};
                            for (size_t i = 0; i < top && i < byl.size(); ++i) { const size_t wg = byl[i].second / (size_t)(4 * HB.mfma_mt); const unsigned y = (unsigned)(wg / grid.x); if (y < 8) ++cnt[y]; }
                            for (unsigned y = 0; y < grid.y && y < 8; ++y) fprintf(stderr, "%d ", cnt[y]);
                            fprintf(stderr, "\n");
                        }
                        if (nw)
                            fprintf(stderr, "[sddmm stamps] grid %u x %u, %d working waves; shader clocks per wave: prologue %.0f, wait+barrier %.0f, issue %.0f, reads+products %.0f, "
                                            "row/column sums %.0f, tile+stores+atomics %.0f, lifetime %.0f (max %.0f)\n",
                                    grid.x, grid.y, nw, sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[5] / nw, sum[7] / nw, sum[4] / nw, life_max);
                    }
                    sd_done = true;
                    // (MMW_FV_IN_SDDMM=0: certified by spare workgroups of the next iteration's DUAL phase, or by a launch of its own after the chunk's last)
                    if (first_it && !fv_rides) fv_pending = fv_now;
                    rs_ok = rs_out != nullptr;
                }
            }
            if (!sd_done && eng.use_blk) MMW_TRY(ensure_sd());
            if (sd_done) {
            } else if (sddmm_blk2 && eng.use_blk) {
                unsigned long long* sd_stamps = nullptr;  // MMW_SD_STAMPS=1: phase stamps of the last iteration's SDDMM
                DevBuf<unsigned long long> stamp_buf;
                if (it + 1 == n && getenv("MMW_SD_STAMPS")) {
                    MMW_TRY(stamp_buf.alloc((size_t)16 * 8192));
                    MMW_HIP(hipMemsetAsync(stamp_buf.p, 0, (size_t)16 * 8192 * sizeof(unsigned long long), st));
                    sd_stamps = stamp_buf.p;
                }
                Sd2Dev S;
                S.ptr = b_sd2ptr.p; S.ab = b_sd2ab.p; S.epos = b_sd2epos.p; S.items = b_sd2items.p; S.nitems = sd2_nitems;
                constexpr int CT2 = B2_ROW_BYTES / (int)sizeof(T);
                const int per = (sd2_nitems + 7) / 8;
                SketchArgs<T> sk{};
                const size_t sd_lds = std::max((size_t)HB.un8_max * B2_ROW_BYTES, std::min((size_t)(SD2_THREADS / WAVE) * Dpad * sizeof(double), (size_t)65536));
                const bool lzm = eng.method == MMW_EXPM_LANCZOS;
                if (fuse_sketch && !randv && it + 1 < n && (size_t)(SD2_THREADS / WAVE) * Dpad * sizeof(double) <= sd_lds) {
                    // the start block and its norm slabs are free once the combination has run: draw the next iteration's sketch here
                    constexpr int VBW = SD2_THREADS / BLOCK;  // a workgroup here stands for this many of the stand-alone kernel's
                    sk.nblocks = (sketch_slabs() + VBW - 1) / VBW;
                    sk.K = K; sk.D = D; sk.seed = seed; sk.iter = (uint32_t)(iter + 1);
                    sk.R = eng.start_block();
                    sk.colsq_part = lzm ? eng.partial_sq.p : nullptr;
                    sk.planes = eng.start_planes();
                    sk.planes_f16 = 0;
                    eng.planes_ready[0] = sk.planes != nullptr;
                    eng.planes0_f16 = false;
                    sketch_done_for = (int64_t)iter + 1;
                    sketch_done_seed = seed;
                    sketch_done_slabs = sk.nblocks * VBW;
                }
                hipLaunchKernelGGL((k_sddmm_blk2<T>), dim3(per * 8 + sk.nblocks), dim3(SD2_THREADS), sd_lds, st, blkdev(), S, P, Dpad,
                                   (Dpad + CT2 - 1) / CT2, Xh.p, drow.p, tr_part.p, gr, xval.p, xavg.p, acc, sk, sd_stamps);
                if (sd_stamps) MMW_TRY(dump_stamps(sd_stamps));
            } else if (sddmm_blk && eng.use_blk) {
                SdDev S;
                S.ptr = b_sdptr.p; S.la = b_sdla.p; S.lb = b_sdlb.p; S.epos = b_sdepos.p;
                constexpr int CT = BLK_TILE_BYTES / (int)sizeof(T);
                const int ntiles = (Dpad + CT - 1) / CT;
                const int per = (HB.nb() + 7) / 8;
                hipLaunchKernelGGL((k_sddmm_blk<T>), dim3(per * 8), dim3(BLK_THREADS), (size_t)BLK_UNION_ROWS * BLK_TILE_BYTES, st, blkdev(), S, P, Dpad,
                                   ntiles, Xh.p, drow.p, tr_part.p, gr, xval.p, xavg.p, acc);
            } else
            switch (eng.lay.NCH) {
                case 1: hipLaunchKernelGGL((k_sddmm<T, 1>), dim3(gr), dim3(BLOCK), 0, st, P, Dpad, eng.lay.LPR, eng.lay.G, Xh.p, drow.p, tr_part.p, gr, xval.p, xavg.p, acc); break;
                case 2: hipLaunchKernelGGL((k_sddmm<T, 2>), dim3(gr), dim3(BLOCK), 0, st, P, Dpad, eng.lay.LPR, eng.lay.G, Xh.p, drow.p, tr_part.p, gr, xval.p, xavg.p, acc); break;
                case 3: hipLaunchKernelGGL((k_sddmm<T, 3>), dim3(gr), dim3(BLOCK), 0, st, P, Dpad, eng.lay.LPR, eng.lay.G, Xh.p, drow.p, tr_part.p, gr, xval.p, xavg.p, acc); break;
                default: hipLaunchKernelGGL((k_sddmm<T, 4>), dim3(gr), dim3(BLOCK), 0, st, P, Dpad, eng.lay.LPR, eng.lay.G, Xh.p, drow.p, tr_part.p, gr, xval.p, xavg.p, acc); break;
            }
            // the running sum of X (mmw.py:77-78): one coalesced pass; none of the SDDMM kernels read-modify-writes xavg
            if (sd_done) {  // the matrix-core SDDMM added X to its running sum itself (tile order)
            } else if (acc && it + 1 < n && !kt_exact()) xavg_deferred = true;  // the next iteration's LOSS pass adds it
            else if (acc) hipLaunchKernelGGL((k_accumulate<T>), dim3((unsigned)std::min<size_t>(((size_t)H.nnzL() + BLOCK - 1) / BLOCK, 4096)), dim3(BLOCK), 0, st, (size_t)H.nnzL(), xval.p, xavg.p);
            MMW_TRY(kt.end());
            MMW_HIP(hipGetLastError());
            MMW_TRY(record(3));
            ++iter;
        }
        if (fv_pending.plan) {  // the chunk's last first-order exponential
            hipLaunchKernelGGL(k_first_verify, dim3(fv_pending.nwg), dim3(BLOCK), 0, st, fv_pending);
            MMW_HIP(hipGetLastError());
        }
        if (optimistic && n > 1 && lag_chunk && eng.method == MMW_EXPM_LANCZOS) {
            // the chunk's last plan was extrapolated and no later plan of the chunk sees its matrix: check it here
            hipLaunchKernelGGL((k_rowsums<T>), dim3(eng.nwide), dim3(BLOCK), 0, st, K, d_indptr.p, d_col.p, lval.p, 0.5, eng.row_part.p);
            hipLaunchKernelGGL(k_plan_verify, dim3(1), dim3(PLAN_THREADS), 0, st, K, eng.row_part.p, eng.nwide, eng.plan_d.p, eng.viol_d.p, age() - 1);
            MMW_HIP(hipGetLastError());
        }
        // until settle() finds a violation or something touches the iterate.  A handle that has had to replay a chunk keeps restarting
        // its chunks exactly (measured on er-1pct, whose order rises during the run: 5 100 it/s so, 4 500 chained)
        chain_ok = optimistic && n > 1 && replays == 0 && !getenv("MMW_NO_CHUNK_CHAIN");
        rs_last = rs_ok;
        return MMW_OK;
    }
    int sync() override {
        if (host_only) return fail(MMW_ERR_STATE, "this handle was created with device -1 (host pattern only)");
        MMW_HIP(hipSetDevice(device));
        MMW_TRY(settle());
        const bool want_emax = emax_enq_iter == iter && emax_iter != iter && emax_d.p != nullptr;
        if (want_emax) MMW_HIP(hipMemcpyAsync(&emax_h, emax_d.p, sizeof(double), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        if (want_emax) emax_iter = iter;
        MMW_TRY(kt.flush());
        return flush_events();
    }

    // The Philox sketch of (seed, iteration) exactly as the loop draws it -- the generator is counter-based, so this is the block
    // iteration `iteration` of a device-RNG run with that seed multiplied, whatever chunk it ran in (parity tests give it to the oracle).
    int sketch(uint64_t seed, int32_t iteration, double* out, int64_t n) override {
        if (host_only) return fail(MMW_ERR_STATE, "this handle was created with device -1 (host pattern only)");
        if (iteration < 0) return fail(MMW_ERR_ARG, "mmw_sketch: iteration must be >= 0");
        MMW_HIP(hipSetDevice(device));
        MMW_TRY(sync());
        hipLaunchKernelGGL((k_sketch_rng<T>), dim3(grid_rows(K)), dim3(BLOCK), 0, st, K, D, eng.lay.Dpad, seed, (uint32_t)iteration, eng.Tm.p, (double*)nullptr);
        MMW_HIP(hipGetLastError());
        return export_block(eng.Tm.p, out, n);
    }

    int export_T(const T* src, size_t n, double* out, int64_t have) {
        if ((int64_t)n != have) return fail(MMW_ERR_ARG, "mmw_read_f64: wrong length " + std::to_string(have) + ", expected " + std::to_string(n));
        hipLaunchKernelGGL((k_to_f64<T>), dim3(grid_elems(n)), dim3(BLOCK), 0, st, n, src, out64.p);
        MMW_HIP(hipGetLastError());
        MMW_TRY(copy_d2h(out, out64.p, (size_t)n * sizeof(double), st));
        return MMW_OK;
    }
    int export_host(const std::vector<double>& v, double* out, int64_t have) {
        if ((int64_t)v.size() != have) return fail(MMW_ERR_ARG, "mmw_read_f64: wrong length");
        if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(double));
        return MMW_OK;
    }
    int export_block(const T* src, double* out, int64_t have) {
        const size_t n = (size_t)K * D;
        if ((int64_t)n != have) return fail(MMW_ERR_ARG, "mmw_read_f64: wrong length for a K x D block");
        hipLaunchKernelGGL((k_export_block<T>), dim3(grid_elems(n)), dim3(BLOCK), 0, st, K, D, eng.lay.Dpad, src, out64.p);
        MMW_HIP(hipGetLastError());
        MMW_TRY(copy_d2h(out, out64.p, (size_t)n * sizeof(double), st));
        return MMW_OK;
    }
    int read_f64(int which, double* out, int64_t n) override {
        if (host_only) {
            switch (which) {
                case MMW_F_S_SUM: return export_host(H.S_sum, out, n);
                case MMW_F_NORM_H: return export_host(H.norm_H, out, n);
                case MMW_F_ST_DATA: return export_host(H.st_data, out, n);
                default: return fail(MMW_ERR_STATE, "this handle was created with device -1 (host pattern only)");
            }
        }
        MMW_HIP(hipSetDevice(device));
        MMW_TRY(sync());
        if (which == MMW_F_ST_DATA) MMW_TRY(ensure_host_lists());
        const size_t nnz = (size_t)H.nnzL(), C = (size_t)H.C();
        switch (which) {
            case MMW_F_Y: return export_T(Y.p, C, out, n);
            case MMW_F_E_ACCU: return export_T(e_accu.p, C, out, n);
            case MMW_F_E_THIS: return export_T(e_this.p, C, out, n);
            case MMW_F_E_MAX: {
                if (n != 1) return fail(MMW_ERR_ARG, "the maximum violation is one number");
                if (emax_iter == iter && emax_iter >= 0) {  // reduced behind the last mmw_iterate's work and fetched by mmw_sync
                    out[0] = emax_h;
                    return MMW_OK;
                }
                hipLaunchKernelGGL((k_max_of<T>), dim3(1), dim3(1024), 0, st, C, e_this.p, out64.p);
                MMW_HIP(hipGetLastError());
                return copy_d2h(out, out64.p, sizeof(double), st);
            }
            case MMW_F_LVAL: return export_T(lval.p, nnz, out, n);
            case MMW_F_XVAL: MMW_TRY(x_csr_view()); return export_T(xval.p, nnz, out, n);
            case MMW_F_XAVG: MMW_TRY(x_csr_view()); return export_T(xavg.p, nnz, out, n);
            case MMW_F_YAVG: return export_T(yavg.p, C, out, n);
            case MMW_F_XHALF: return export_block(Xh.p, out, n);
            case MMW_F_SKETCH: {
                if (!last_was_rng || iter == 0) return fail(MMW_ERR_STATE, "the sketch can be read back only after a device-generated iteration");
                hipLaunchKernelGGL((k_sketch_rng<T>), dim3(grid_rows(K)), dim3(BLOCK), 0, st, K, D, eng.lay.Dpad, last_seed, (uint32_t)(iter - 1), eng.Tm.p, (double*)nullptr);
                return export_block(eng.Tm.p, out, n);
            }
            case MMW_F_S_SUM: return export_host(H.S_sum, out, n);
            case MMW_F_NORM_H: return export_host(H.norm_H, out, n);
            case MMW_F_ST_DATA: return export_host(H.st_data, out, n);
            case MMW_F_PHASE_US: return export_host(phase_us, out, n);
            case MMW_F_EXPM_INFO: {
                if (n != 4) return fail(MMW_ERR_ARG, "expm info has 4 entries");
                out[0] = eng.last.rho; out[1] = eng.last.m_eff > 0 ? eng.last.m_eff : eng.last.m; out[2] = eng.last.nsub; out[3] = eng.last.mu;
                return MMW_OK;
            }
            case MMW_F_BLOCKING: {
                if (n != 4) return fail(MMW_ERR_ARG, "blocking info has 4 entries");
                out[0] = eng.use_blk ? 1.0 : 0.0; out[1] = HB.usable ? HB.nb() : 0; out[2] = HB.reuse; out[3] = (double)replays;
                return MMW_OK;
            }
            case MMW_F_SPMM_KIND: {
                if (n != 2) return fail(MMW_ERR_ARG, "spmm kind has 2 entries");
                out[0] = !eng.use_blk ? 0.0 : (eng.use_mfma ? 3.0 : (eng.blk.half_tile ? 2.0 : 1.0));
                out[1] = eng.use_mfma && eng.last_mfma_ok ? 1.0 : 0.0;
                return MMW_OK;
            }
            case MMW_F_DUAL_INFO: {
                if (n != 4) return fail(MMW_ERR_ARG, "dual info has 4 entries");
                out[0] = (double)n_rs_iters; out[1] = (double)n_fused_iters; out[2] = (double)n_first_iters; out[3] = (double)n_first16_iters;
                return MMW_OK;
            }
            case MMW_F_FACTOR: return extras.read_factor(out, n);
            case MMW_F_KERNEL_US: {
                if (n != 2 * KT_NSLOT) return fail(MMW_ERR_ARG, "kernel timers have 2*9 entries");
                for (int i = 0; i < KT_NSLOT; ++i) { out[2 * i] = kt.total_us[i]; out[2 * i + 1] = kt.count[i]; }
                return MMW_OK;
            }
            default: return fail(MMW_ERR_ARG, "mmw_read_f64: unknown field");
        }
    }
    int export_i32(const std::vector<int32_t>& v, int32_t* out, int64_t have) {
        if ((int64_t)v.size() != have) return fail(MMW_ERR_ARG, "mmw_read_i32: wrong length");
        if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int32_t));
        return MMW_OK;
    }
    int read_i32(int which, int32_t* out, int64_t n) override {
        MMW_TRY(ensure_host_lists());
        switch (which) {
            case MMW_I_L_INDPTR: return export_i32(H.l_indptr, out, n);
            case MMW_I_L_INDICES: return export_i32(H.l_indices, out, n);
            case MMW_I_ST_INDPTR: return export_i32(H.st_indptr, out, n);
            case MMW_I_ST_INDICES: return export_i32(H.st_indices, out, n);
            case MMW_I_GAIN_X: return export_i32(H.gain_x, out, n);
            case MMW_I_GAIN_Y: return export_i32(H.gain_y, out, n);
            case MMW_I_ASSO_X: return export_i32(H.asso_x, out, n);
            case MMW_I_ASSO_Y: return export_i32(H.asso_y, out, n);
            case MMW_I_DIAG_POS: return export_i32(H.diag_pos, out, n);
            case MMW_I_ASSO_POS: return export_i32(H.asso_pos, out, n);
            default: return fail(MMW_ERR_ARG, "mmw_read_i32: unknown field");
        }
    }
    int gap(double out[3]) override {
        if (host_only) return fail(MMW_ERR_STATE, "host-only handle");
        MMW_HIP(hipSetDevice(device));
        MMW_TRY(settle());
        if (iter >= nit) return fail(MMW_ERR_STATE, "mmw_gap: call it before an iteration (the running sums then hold iter+1 terms)");
        MMW_TRY(x_csr_view());
        PatternDev<T> Pc = pat();  // the gap's kernels walk the CSR copy
        Pc.e2w = nullptr; Pc.xasso = nullptr; Pc.xdiag_base = -1;
        return extras.gap(Pc, d_lrow.p, xavg.p, yavg.p, iter + 1, out);
    }
    int factor(int32_t rank, double* out, uint64_t seed) override {
        if (host_only) return fail(MMW_ERR_STATE, "host-only handle");
        MMW_HIP(hipSetDevice(device));
        MMW_TRY(settle());
        if (iter < nit) return fail(MMW_ERR_STATE, "mmw_factor: run all announced iterations first (the average divides by nit)");
        MMW_TRY(x_csr_view());
        return extras.factor(d_indptr.p, d_col.p, xavg.p, nit, rank, out, seed);
    }
    int round(int32_t Zr, int32_t Dp, const double* gX, int32_t nbatch, const double* randv, int32_t* z_out, int32_t* rem_out) override {
        if (host_only) return fail(MMW_ERR_STATE, "host-only handle");
        MMW_HIP(hipSetDevice(device));
        return extras.round(Zr, Dp, gX, nbatch, randv, z_out, rem_out);
    }
};

template <typename T>
int expm_apply_impl(int device, int method, int max_order, double tol, int32_t K, int32_t D, const int32_t* indptr,
                    const int32_t* indices, const double* data, const double* B, double* out, double info[4], int32_t reps,
                    double* kernel_us) {
    MMW_HIP(hipSetDevice(device));
    hipStream_t st;
    MMW_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Guard {
        hipStream_t s;
        ~Guard() { (void)hipStreamDestroy(s); }
    } guard{st};
    const int64_t nnz = indptr[K];
    std::vector<int32_t> ip(indptr, indptr + K + 1), ci(indices, indices + nnz);
    std::vector<double> vv(data, data + nnz);
    DevBuf<int> d_ip, d_ci;
    DevBuf<T> d_v, d_out;
    DevBuf<double> d_b64, d_o64;
    MMW_TRY(d_ip.upload(ip, st));
    MMW_TRY(d_ci.upload(ci, st));
    MMW_TRY(d_v.upload_cast(vv, st));
    ExpmEngine<T> eng;
    MMW_TRY(eng.init(st, K, D, d_ip.p, d_ci.p, d_v.p));
    eng.method = method; eng.max_order = max_order; eng.tol = tol;
    MMW_TRY(d_out.alloc(eng.bs));
    MMW_TRY(d_b64.alloc((size_t)K * D));
    MMW_TRY(d_o64.alloc((size_t)K * D));
    MMW_TRY(copy_h2d(d_b64.p, B, (size_t)K * D * sizeof(double), st));
    hipEvent_t e0, e1;
    MMW_HIP(hipEventCreate(&e0));
    MMW_HIP(hipEventCreate(&e1));
    double us = 0.0;
    if (reps < 1) reps = 1;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL((k_import_block<T>), dim3(grid_elems(eng.bs)), dim3(BLOCK), 0, st, K, D, eng.lay.Dpad, d_b64.p, eng.start_block());
        MMW_HIP(hipEventRecord(e0, st));
        int rc = eng.apply(d_out.p, 1.0);
        if (rc != MMW_OK) return rc;
        MMW_HIP(hipEventRecord(e1, st));
        MMW_HIP(hipStreamSynchronize(st));
        float ms = 0;
        MMW_HIP(hipEventElapsedTime(&ms, e0, e1));
        us += ms * 1e3;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    hipLaunchKernelGGL((k_export_block<T>), dim3(grid_elems((size_t)K * D)), dim3(BLOCK), 0, st, K, D, eng.lay.Dpad, d_out.p, d_o64.p);
    MMW_HIP(hipGetLastError());
    MMW_TRY(copy_d2h(out, d_o64.p, (size_t)K * D * sizeof(double), st));
    if (info) {
        info[0] = eng.last.rho; info[1] = eng.last.m_eff > 0 ? eng.last.m_eff : eng.last.m; info[2] = eng.last.nsub; info[3] = eng.last.mu;
    }
    if (kernel_us) *kernel_us = us / reps;
    return MMW_OK;
}

}  // namespace

extern "C" {

const char* mmw_last_error(void) { return last_error_ref().c_str(); }
int mmw_version(void) { return 300; }
int mmw_device_count(int* n) {
    if (!n) return fail(MMW_ERR_ARG, "null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(MMW_ERR_HIP, std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
    }
    *n = c;
    return MMW_OK;
}

int mmw_create(mmw_solver** out, int device, int dtype, int32_t K, int32_t Z, int32_t rank_radio, double eta, int32_t nit,
               const int32_t* S_indptr, const int32_t* S_indices, const double* S_data, const int32_t* Q_indptr,
               const int32_t* Q_indices, const double* Q_data, const double* h_max) {
    if (!out || !S_indptr || !S_indices || !S_data || !Q_indptr || !Q_indices || !Q_data || !h_max)
        return fail(MMW_ERR_ARG, "mmw_create: null pointer");
    *out = nullptr;
    if (rank_radio < 1) return fail(MMW_ERR_ARG, "rank_radio must be >= 1");
    if (nit < 1) return fail(MMW_ERR_ARG, "nit must be >= 1");
    const bool host_only = device == -1;
    if (!host_only) {
        int ndev = 0;
        MMW_TRY(mmw_device_count(&ndev));
        if (device < 0 || device >= ndev) return fail(MMW_ERR_HIP, "mmw_create: no such HIP device " + std::to_string(device) + " (" + std::to_string(ndev) + " visible)");
    }
    int rc;
    if (dtype == MMW_F32) {
        auto s = std::make_unique<Solver<float>>();
        s->host_only = host_only;
        rc = s->init(device, K, Z, rank_radio, eta, nit, S_indptr, S_indices, S_data, Q_indptr, Q_indices, Q_data, h_max);
        if (rc == MMW_OK) *out = s.release();
    } else if (dtype == MMW_F64) {
        auto s = std::make_unique<Solver<double>>();
        s->host_only = host_only;
        rc = s->init(device, K, Z, rank_radio, eta, nit, S_indptr, S_indices, S_data, Q_indptr, Q_indices, Q_data, h_max);
        if (rc == MMW_OK) *out = s.release();
    } else {
        rc = fail(MMW_ERR_ARG, "dtype must be MMW_F32 or MMW_F64");
    }
    return rc;
}
int mmw_create_from_env(mmw_solver** out, mmw_env* env, int dtype, int32_t Z, int32_t rank_radio, double eta, int32_t nit) {
    if (!out || !env) return fail(MMW_ERR_ARG, "mmw_create_from_env: null pointer");
    *out = nullptr;
    if (rank_radio < 1) return fail(MMW_ERR_ARG, "rank_radio must be >= 1");
    if (nit < 1) return fail(MMW_ERR_ARG, "nit must be >= 1");
    int rc;
    if (dtype == MMW_F32) {
        auto s = std::make_unique<Solver<float>>();
        rc = s->init_env(env->e.device, env->e, Z, rank_radio, eta, nit);
        if (rc == MMW_OK) *out = s.release();
    } else if (dtype == MMW_F64) {
        auto s = std::make_unique<Solver<double>>();
        rc = s->init_env(env->e.device, env->e, Z, rank_radio, eta, nit);
        if (rc == MMW_OK) *out = s.release();
    } else {
        rc = fail(MMW_ERR_ARG, "dtype must be MMW_F32 or MMW_F64");
    }
    return rc;
}
int mmw_env_bounds(mmw_env* e, int32_t out[2]) {
    if (!e || !out) return fail(MMW_ERR_ARG, "null pointer");
    return e->e.bounds(out);
}
int mmw_destroy(mmw_solver* s) {
    delete s;
    return MMW_OK;
}
#define MMW_NEED(s) \
    if (!(s)) return fail(MMW_ERR_ARG, "null solver handle")
int mmw_sizes(mmw_solver* s, int64_t out[10]) { MMW_NEED(s); return s->sizes(out); }
int mmw_set_expm(mmw_solver* s, int method, int max_order, double tol) { MMW_NEED(s); return s->set_expm(method, max_order, tol); }
int mmw_set_timing(mmw_solver* s, int enabled) { MMW_NEED(s); return s->set_timing(enabled); }
int mmw_set_profile(mmw_solver* s, int enabled) { MMW_NEED(s); return s->set_profile(enabled); }
int mmw_bench_spmm(mmw_solver* s, int blocked, int reps, double* avg_us) { MMW_NEED(s); return s->bench_spmm(blocked, reps, avg_us); }
int mmw_reset(mmw_solver* s, int32_t nit) { MMW_NEED(s); return s->reset(nit); }
int mmw_set_slots(mmw_solver* s, int32_t Z, int32_t nit) { MMW_NEED(s); return s->set_slots(Z, nit, 0); }
int mmw_set_slots_warm(mmw_solver* s, int32_t Z, int32_t nit) { MMW_NEED(s); return s->set_slots(Z, nit, 1); }
int mmw_set_eta(mmw_solver* s, double eta) { MMW_NEED(s); return s->set_eta(eta); }
int mmw_iterate(mmw_solver* s, int32_t n, const double* randv, uint64_t seed) { MMW_NEED(s); return s->iterate(n, randv, seed); }
int mmw_sync(mmw_solver* s) { MMW_NEED(s); return s->sync(); }
int mmw_sketch(mmw_solver* s, uint64_t seed, int32_t iteration, double* out, int64_t n) { MMW_NEED(s); if (!out && n) return fail(MMW_ERR_ARG, "null output"); return s->sketch(seed, iteration, out, n); }
int mmw_read_f64(mmw_solver* s, int which, double* out, int64_t n) { MMW_NEED(s); if (!out && n) return fail(MMW_ERR_ARG, "null output"); return s->read_f64(which, out, n); }
int mmw_read_i32(mmw_solver* s, int which, int32_t* out, int64_t n) { MMW_NEED(s); if (!out && n) return fail(MMW_ERR_ARG, "null output"); return s->read_i32(which, out, n); }
int mmw_gap(mmw_solver* s, double out[3]) { MMW_NEED(s); return s->gap(out); }
int mmw_factor(mmw_solver* s, int32_t rank, double* out, uint64_t seed) { MMW_NEED(s); return s->factor(rank, out, seed); }
int mmw_round(mmw_solver* s, int32_t Zr, int32_t Dp, const double* gX, int32_t nbatch, const double* randv, int32_t* z_out, int32_t* rem_out) {
    MMW_NEED(s);
    return s->round(Zr, Dp, gX, nbatch, randv, z_out, rem_out);
}

int mmw_expm_apply(int device, int dtype, int method, int max_order, double tol, int32_t K, int32_t D, const int32_t* indptr,
                   const int32_t* indices, const double* data, const double* B, double* out, double info[4], int32_t reps,
                   double* kernel_us) {
    if (!indptr || !indices || !data || !B || !out) return fail(MMW_ERR_ARG, "mmw_expm_apply: null pointer");
    if (K < 1 || D < 1) return fail(MMW_ERR_ARG, "mmw_expm_apply: K and D must be positive");
    if (max_order < 1 || max_order > MAX_ORDER) return fail(MMW_ERR_ARG, "max_order must be in [1,16]");
    if (method != MMW_EXPM_LANCZOS && method != MMW_EXPM_TAYLOR) return fail(MMW_ERR_ARG, "unknown expm method");
    int ndev = 0;
    MMW_TRY(mmw_device_count(&ndev));
    if (device < 0 || device >= ndev) return fail(MMW_ERR_HIP, "mmw_expm_apply: no such HIP device");
    if (dtype == MMW_F32) return expm_apply_impl<float>(device, method, max_order, tol, K, D, indptr, indices, data, B, out, info, reps, kernel_us);
    if (dtype == MMW_F64) return expm_apply_impl<double>(device, method, max_order, tol, K, D, indptr, indices, data, B, out, info, reps, kernel_us);
    return fail(MMW_ERR_ARG, "dtype must be MMW_F32 or MMW_F64");
}


int mmw_sym_eig(int device, int32_t b, const double* G, double rel_tol, int32_t max_sweeps, double* theta, double* Q, int32_t* sweeps) {
    if (!G || !theta || !Q) return fail(MMW_ERR_ARG, "mmw_sym_eig: null pointer");
    if (b < 1 || b > 2048) return fail(MMW_ERR_ARG, "mmw_sym_eig: b must be in [1, 2048]");
    if (!(rel_tol > 0.0) || max_sweeps < 1) return fail(MMW_ERR_ARG, "mmw_sym_eig: rel_tol and max_sweeps must be positive");
    int ndev = 0;
    MMW_TRY(mmw_device_count(&ndev));
    if (device < 0 || device >= ndev) return fail(MMW_ERR_HIP, "mmw_sym_eig: no such HIP device");
    MMW_HIP(hipSetDevice(device));
    struct Stream {
        hipStream_t s = nullptr;
        ~Stream() { if (s) (void)hipStreamDestroy(s); }
    } stream;
    MMW_HIP(hipStreamCreate(&stream.s));
    mmw::DenseWork<double> dw;
    dw.st = stream.s;
    MMW_TRY(dw.ensure(b, 1));
    MMW_TRY(copy_h2d(dw.G.p, G, (size_t)b * b * sizeof(double), stream.s));
    int sw = 0;
    MMW_TRY(dw.jacobi(b, rel_tol, max_sweeps, &sw));
    MMW_TRY(copy_d2h(theta, dw.diag.p, (size_t)b * sizeof(double), stream.s));
    MMW_TRY(copy_d2h(Q, dw.Q.p, (size_t)b * b * sizeof(double), stream.s));
    if (sweeps) *sweeps = sw;
    return MMW_OK;
}

// ---- problem generator and scorer on the device (include/mmw_hip.h) ----------------------------------------
int mmw_env_create(mmw_env** out, int device, int32_t K, int32_t A, const double* sta_xy, const double* ap_xy, double fre_Hz, double txp_offset,
                   double min_s_n_ratio, double min_sinr, double noise_floor_dbm) {
    if (!out || !sta_xy || !ap_xy) return fail(MMW_ERR_ARG, "mmw_env_create: null pointer");
    *out = nullptr;
    if (K < 1 || A < 1) return fail(MMW_ERR_ARG, "mmw_env_create: K and A must be positive");
    if (!(min_sinr > 0.0) || !(txp_offset > 0.0) || !(fre_Hz > 0.0)) return fail(MMW_ERR_ARG, "mmw_env_create: min_sinr, txp_offset and fre_Hz must be positive");
    int ndev = 0;
    MMW_TRY(mmw_device_count(&ndev));
    if (device < 0 || device >= ndev) return fail(MMW_ERR_HIP, "mmw_env_create: no such HIP device " + std::to_string(device) + " (" + std::to_string(ndev) + " visible)");
    auto h = std::make_unique<mmw_env>();
    const int rc = h->e.init(device, K, A, sta_xy, ap_xy, fre_Hz, txp_offset, min_s_n_ratio, min_sinr, noise_floor_dbm);
    if (rc == MMW_OK) *out = h.release();
    return rc;
}
int mmw_env_destroy(mmw_env* e) {
    delete e;
    return MMW_OK;
}
int mmw_env_sizes(mmw_env* e, int64_t out[4]) {
    if (!e || !out) return fail(MMW_ERR_ARG, "null pointer");
    out[0] = e->e.K; out[1] = e->e.A; out[2] = e->e.nnzS; out[3] = e->e.nnzQ;
    return MMW_OK;
}
int mmw_env_state(mmw_env* e, int32_t* S_indptr, int32_t* S_indices, double* S_data, int32_t* Q_indptr, int32_t* Q_indices, double* Q_data,
                  double* h_max) {
    if (!e || !S_indptr || !S_indices || !S_data || !Q_indptr || !Q_indices || !Q_data || !h_max) return fail(MMW_ERR_ARG, "mmw_env_state: null pointer");
    return e->e.state(S_indptr, S_indices, S_data, Q_indptr, Q_indices, Q_data, h_max);
}
int mmw_env_evaluate(mmw_env* e, const double* z_vec, int32_t Z, double packet_bit, double bandwidth, double slot_time, double* sinr_out,
                     double* bler_out) {
    if (!e || !z_vec || !sinr_out) return fail(MMW_ERR_ARG, "mmw_env_evaluate: null pointer");
    return e->e.evaluate(z_vec, Z, packet_bit, bandwidth, slot_time, sinr_out, bler_out);
}

}  // extern "C"
