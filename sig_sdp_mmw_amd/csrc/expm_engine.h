// Host orchestration of exp(ascale * A) B on one stream: planning (one-norm bound -> order, substeps),
// per-column Lanczos or shifted Taylor on top of the CSR SpMM.  Used by the MMW loop (A = L_accu, ascale = 1/2)
// and by the stand-alone seam mmw_expm_apply.
#pragma once
#include <cmath>

#include <functional>
#include <type_traits>

#include "kernels_expm.h"
#include "kernels_mfma.h"
#include "runtime.h"

namespace mmw {

inline int make_layout(int D, int vec, BlockLayout& lay, std::string& err) {
    int lpr = (D + vec - 1) / vec;
    if (lpr <= 32) {
        int p = 1;
        while (p < lpr) p <<= 1;
        lpr = p;
        lay.G = WAVE / lpr;
        lay.NCH = 1;
    } else {
        lpr = (lpr + 7) / 8 * 8;  // rows in whole 128-byte cache lines: a gathered 128-byte piece then is ONE line, not two halves
        lay.G = 1;
        lay.NCH = (lpr + WAVE - 1) / WAVE;
        if (lay.NCH > 4) {
            err = "sketch width D too large for this build (max 1024 f32 / 512 f64 columns)";
            return MMW_ERR_ARG;
        }
    }
    lay.D = D;
    lay.LPR = lpr;
    lay.Dpad = lpr * vec;
    return MMW_OK;
}

// The generic SpMM's small-K form (k_spmm_slice): bytes of a staged row slice (0: not applicable) and the row ranges of its grid --
// the partial slabs a Lanczos launch really fills
template <typename T> inline int slice_bytes(int K, const BlockLayout& lay) {
    static const bool off = getenv("MMW_NO_SLICE_SPMM") != nullptr;
    const int row_bytes = lay.Dpad * (int)sizeof(T);
    if (off || K < 256) return 0;
    if ((size_t)K * 64 <= (size_t)SLICE_LDS_MAX && row_bytes % 64 == 0) return 64;
    if ((size_t)K * 32 <= (size_t)SLICE_LDS_MAX && row_bytes % 32 == 0) return 32;
    return 0;
}
template <typename T> inline int slice_ranges(int K, const BlockLayout& lay, int sb) {
    const int nslices = lay.Dpad * (int)sizeof(T) / sb;
    // a slice takes most of a CU's LDS: one workgroup per CU, ONE round of them (a second round stages everything again)
    return std::max(1, std::min((K + 15) / 16, device_cus() / std::max(1, nslices)));
}
// slabs of partial sums a Lanczos launch of the generic path fills (the rest of the caller's `nblk` slabs it clears)
template <typename T> inline int generic_slabs(int K, const BlockLayout& lay, int nblk) {
    const int sb = slice_bytes<T>(K, lay);
    return sb ? std::min(nblk, slice_ranges<T>(K, lay, sb)) : nblk;
}
// one launch of the CSR SpMM (any fused epilogue) for a block with layout `lay`
template <typename T, int MODE>
inline int spmm_launch(hipStream_t st, int K, const BlockLayout& lay, int nblk, const int* indptr, const int* col, const T* val,
                       const T* in, T* out, T* F, const T* X2, double c1, double c2, double c3, double* partial,
                       const ExpmPlan* plan = nullptr, int step = 0, double* partial_o2 = nullptr, int nslabs_fold = -1) {
    if (const int sb = slice_bytes<T>(K, lay)) {  // small K without locality: the block's column slices staged in LDS (k_spmm_slice)
        const int nslices = lay.Dpad * (int)sizeof(T) / sb;
        int nranges = slice_ranges<T>(K, lay, sb);
        if (MODE == SPMM_LANCZOS) nranges = std::min(nranges, nblk);
        const int fold = nslabs_fold > 0 ? nslabs_fold : nblk;  // slabs the caller folds: the kernel clears the ones past its own
        const size_t lds = (size_t)K * sb;
        if (sb == 64) {
            MMW_TRY(set_max_lds(reinterpret_cast<const void*>(&k_spmm_slice<T, MODE, 64>), SLICE_LDS_MAX));
            hipLaunchKernelGGL((k_spmm_slice<T, MODE, 64>), dim3(nslices, nranges), dim3(SLICE_THREADS), lds, st, K, lay.Dpad, indptr, col, val, in, out, F, X2, c1, c2, c3,
                               partial, plan, step, partial_o2, fold);
        } else {
            MMW_TRY(set_max_lds(reinterpret_cast<const void*>(&k_spmm_slice<T, MODE, 32>), SLICE_LDS_MAX));
            hipLaunchKernelGGL((k_spmm_slice<T, MODE, 32>), dim3(nslices, nranges), dim3(SLICE_THREADS), lds, st, K, lay.Dpad, indptr, col, val, in, out, F, X2, c1, c2, c3,
                               partial, plan, step, partial_o2, fold);
        }
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    const size_t sh = MODE == SPMM_LANCZOS ? (size_t)WAVES_PER_BLOCK * lay.Dpad * sizeof(double) : 0;
    switch (lay.NCH) {
        case 1: hipLaunchKernelGGL((k_spmm<T, 1, MODE>), dim3(nblk), dim3(BLOCK), sh, st, K, lay, indptr, col, val, in, out, F, X2, c1, c2, c3, partial, plan, step, partial_o2); break;
        case 2: hipLaunchKernelGGL((k_spmm<T, 2, MODE>), dim3(nblk), dim3(BLOCK), sh, st, K, lay, indptr, col, val, in, out, F, X2, c1, c2, c3, partial, plan, step, partial_o2); break;
        case 3: hipLaunchKernelGGL((k_spmm<T, 3, MODE>), dim3(nblk), dim3(BLOCK), sh, st, K, lay, indptr, col, val, in, out, F, X2, c1, c2, c3, partial, plan, step, partial_o2); break;
        default: hipLaunchKernelGGL((k_spmm<T, 4, MODE>), dim3(nblk), dim3(BLOCK), sh, st, K, lay, indptr, col, val, in, out, F, X2, c1, c2, c3, partial, plan, step, partial_o2); break;
    }
    MMW_HIP(hipGetLastError());
    return MMW_OK;
}

// Static schedule of the half-tile SpMM.  Two workgroups are resident per CU and a workgroup is latency-bound (it takes
// about the same time alone or paired), so the launch is cut into ~2.5 rounds of short workgroups: tile groups of
// nb * ntiles / (2.5 * slots) tiles.  Short workgroups re-stage the block's entries once per tile group (fabric traffic
// 79 MB vs 41 MB algorithmic at the bench config, all of it Infinity-Cache hits) but two co-resident workgroups in
// different phases interleave better than two long ones in lockstep: whole blocks first and only the last partial
// round cut into pieces (MMW_SCHED=2) moves 40 % fewer bytes and measures 5 % slower.
inline Blk2Sched blk2_schedule(int nb, int ntiles) {
    const int slots = 2 * device_cus();
    Blk2Sched s;
    static const bool two_phase = getenv("MMW_SCHED") && atoi(getenv("MMW_SCHED")) == 2;
    if (!two_phase) {
        int tpw = (int)((double)nb * ntiles / (2.5 * slots) + 0.5);  // 2.5-4 rounds measure the same; fewer groups re-stage less
        if (const char* e = getenv("MMW_TPW")) tpw = atoi(e);
        tpw = tpw < 1 ? 1 : (tpw > ntiles ? ntiles : tpw);
        s.nfull = 0; s.grid1 = 0; s.tpw_tail = tpw; s.groups_tail = (ntiles + tpw - 1) / tpw;
        return s;
    }
    s.nfull = nb / slots * slots;
    s.grid1 = (s.nfull + 7) / 8 * 8;
    const int rem = nb - s.nfull;
    s.tpw_tail = ntiles; s.groups_tail = 1;
    double best = 1e300;
    for (int g = 1; g <= ntiles && rem > 0; ++g) {  // rounds x (prologue + tiles), in units of one tile's time
        const int tpw = (ntiles + g - 1) / g, groups = (ntiles + tpw - 1) / tpw;
        const double cost = (double)((rem * groups + slots - 1) / slots) * (1.2 + tpw);
        if (cost < best - 1e-9) { best = cost; s.tpw_tail = tpw; s.groups_tail = groups; }
    }
    return s;
}

inline unsigned long long* g_blk_stamps = nullptr;  // diagnostic builds only: per-workgroup phase stamps
// one launch of the LDS-staged blocked SpMM (values in blocked order)
template <typename T, int MODE>
inline int spmm_blk_launch(hipStream_t st, const BlkDev& B, int Dpad, const T* val_blk, const T* in, T* out, T* F, const T* X2,
                           double c1, double c2, double c3, double* partial, const ExpmPlan* plan = nullptr, int step = 0,
                           double* partial_o2 = nullptr) {
    if (B.half_tile) {
        constexpr int CT2 = B2_ROW_BYTES / (int)sizeof(T);
        const int ntiles = (Dpad + CT2 - 1) / CT2;
        const Blk2Sched sched = blk2_schedule(B.nb, ntiles);
        const int rem = B.nb - sched.nfull;
        const int grid = sched.grid1 + ((rem * sched.groups_tail + 7) / 8) * 8;
        MMW_TRY(set_max_lds(reinterpret_cast<const void*>(&k_spmm_blk2<T, MODE>), B2_LDS_BYTES));
        hipLaunchKernelGGL((k_spmm_blk2<T, MODE>), dim3(grid), dim3(B2_THREADS), B2_LDS_BYTES, st, B, Dpad, ntiles, sched, val_blk, in, out, F, X2, c1, c2, c3, partial, partial_o2, plan, step, g_blk_stamps);
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    constexpr int CT = BLK_TILE_BYTES / (int)sizeof(T);
    const int ntiles = (Dpad + CT - 1) / CT;
    int tpw = (int)((double)B.nb * ntiles / (3.0 * 256.0) + 0.5);  // ~3 workgroups per CU over the launch
    if (getenv("MMW_TPW")) tpw = atoi(getenv("MMW_TPW"));
    tpw = tpw < 1 ? 1 : (tpw > ntiles ? ntiles : tpw);
    const int total = B.nb * ((ntiles + tpw - 1) / tpw);
    const int per = (total + 7) / 8;
    const size_t sh = (size_t)BLK_UNION_ROWS * BLK_TILE_BYTES + BLK_META_LDS + BLK_ROWINFO_LDS + BLK_UNOFF_LDS + (size_t)BLK_WAVES * CT * sizeof(double);
    MMW_TRY(set_max_lds(reinterpret_cast<const void*>(&k_spmm_blk<T, MODE>), (int)sh));
    hipLaunchKernelGGL((k_spmm_blk<T, MODE>), dim3(per * 8), dim3(BLK_THREADS), sh, st, B, Dpad, ntiles, tpw, val_blk, in, out, F, X2, c1, c2, c3, partial, plan, step, g_blk_stamps);
    MMW_HIP(hipGetLastError());
    return MMW_OK;
}

inline unsigned long long* g_mf_stamps = nullptr;  // diagnostic runs only: per-wave phase clocks of the matrix-core SpMM
// one launch of the matrix-core SpMM (kernels_mfma.h): fp32 blocks, plain or Lanczos epilogue.  A workgroup takes a row block
// and a group of GT column tiles; the group is as wide as still gives the chip ~1.5 workgroups per CU (narrower groups re-read
// the block's A fragments once per group).
template <int MODE>
inline int spmm_mfma_launch(hipStream_t st, const MfmaDev& M, int mt, int Dpad, size_t plane_bytes, const char* planes, const float* in,
                            float* out, double ascale, double shift, double* partial, double* partial_o2, const ExpmPlan* plan, int step, int* viol,
                            MfEpi epi = MfEpi{}, int* grid_out = nullptr /* workgroups launched (the first-order epilogue's trace slab) */,
                            hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr /* recorded by the launch itself (KernelTimers::begin_attached) */) {
    const int ntiles = Dpad / 32;
    const int grid_x = (M.nb + 7) / 8 * 8;
    const int cus = device_cus();
#define MMW_MF_LAUNCH(MT, NT, NW, MS, KC, NB)                                                                                          \
    do {                                                                                                                               \
        constexpr int gtw = (NW / MS) * NT;                                                                                            \
        constexpr int lds_bytes = mf_lds_bytes<MT, gtw, KC, NB, mf_planes(MODE), mf_apieces(MODE)>();                                  \
        MMW_TRY(set_max_lds(reinterpret_cast<const void*>(&k_spmm_mfma<MODE, MT, NT, NW, MS, KC, NB>), lds_bytes));                    \
        if (ev_a)                                                                                                                      \
            hipExtLaunchKernelGGL((k_spmm_mfma<MODE, MT, NT, NW, MS, KC, NB>), dim3(grid_x, (ntiles + gtw - 1) / gtw), dim3(NW * 64),  \
                                  lds_bytes, st, ev_a, ev_b, 0, M, Dpad, plane_bytes, planes, in, out, ascale, shift, partial,         \
                                  partial_o2, plan, step, viol, g_mf_stamps, epi);                                                     \
        else                                                                                                                           \
        hipLaunchKernelGGL((k_spmm_mfma<MODE, MT, NT, NW, MS, KC, NB>), dim3(grid_x, (ntiles + gtw - 1) / gtw), dim3(NW * 64),         \
                           lds_bytes, st, M, Dpad, plane_bytes, planes, in, out, ascale, shift, partial, partial_o2,                   \
                           plan, step, viol, g_mf_stamps, epi);                                                                        \
        if (grid_out) *grid_out = grid_x * ((ntiles + gtw - 1) / gtw);                                                                 \
    } while (0)
    int gt = 12;  // column tiles per workgroup: 12, 8 or 4
    static const int gt_env = getenv("MMW_MF_GT") ? atoi(getenv("MMW_MF_GT")) : 0;
    if (gt_env) gt = gt_env;
    else {
        auto wgs = [&](int g) { return (long)M.nb * ((ntiles + g - 1) / g); };
        if (wgs(12) * 2 < 3L * cus) gt = 8;
        if (gt == 8 && wgs(8) * 2 < 3L * cus) gt = 4;
    }
    if (ntiles <= 4) gt = 4;
    else if (ntiles <= 8 && gt > 8) gt = 8;
    static const int cfg = getenv("MMW_MF_CFG") ? atoi(getenv("MMW_MF_CFG")) : 0;  // experiments
    // The fragment image pads a block's k-steps to a multiple of MF_KPAD = 4 (3.6 % more k-steps than a padding to 2) so that the first-order
    // product can take four k-steps per barrier: half the barriers, 32 KB of DMA in flight per workgroup.
    // (row tiles, column tiles per wave, waves, row-tile groups of waves, k-steps per chunk, chunks resident).  Measured at the
    // benchmark (159 blocks of 63 rows, 12 column tiles): 8 waves with the two row tiles on different waves, 2 k-steps per barrier,
    // 2 chunks resident (three workgroups per CU) 22.6 us; 4 waves 26.0; 3 chunks resident 25.9; 1 k-step per barrier 33.4;
    // 8 / 12 column tiles per workgroup 36.7 / 48.0 (too few workgroups for 256 CUs).
    if (mt == 1) {
        if (gt == 4) { if (cfg == 1) MMW_MF_LAUNCH(1, 1, 4, 1, 2, 3); else MMW_MF_LAUNCH(1, 1, 4, 1, 2, 2); }
        else if (gt == 8) MMW_MF_LAUNCH(1, 2, 4, 1, 2, 2);
        else MMW_MF_LAUNCH(1, 3, 4, 1, 1, 3);
    } else {
        if (gt == 4) {
            if (cfg == 1) MMW_MF_LAUNCH(2, 1, 4, 1, 2, 2);
            else if (cfg == 4) MMW_MF_LAUNCH(2, 1, 8, 2, 2, 3);
            else if (mf_first(MODE) && cfg != 6) MMW_MF_LAUNCH(2, 1, 8, 2, 4, 2);  // fp16 operands: four k-steps per barrier fit (68 KB): 21.3 -> 20.2 us
            else MMW_MF_LAUNCH(2, 1, 8, 2, 2, 2);
        }
        else if (gt == 8) MMW_MF_LAUNCH(2, 2, 8, 2, 2, 2);
        else MMW_MF_LAUNCH(2, 3, 4, 1, 1, 3);
    }
#undef MMW_MF_LAUNCH
    MMW_HIP(hipGetLastError());
    return MMW_OK;
}

template <typename T> struct ExpmEngine {
    hipStream_t st = nullptr;
    int K = 0;
    BlockLayout lay{};
    int method = MMW_EXPM_LANCZOS, max_order = 8;
    double tol = 1e-9;
    int nblk = 1, nwide = 1;
    const int* indptr = nullptr;
    const int* col = nullptr;
    const T* val = nullptr;
    DevBuf<T> U;        // (max_order + 1) blocks of K*Dpad; block 0 is the start block
    DevBuf<T> Tm;       // A * U_j
    DevBuf<double> partial_du;  // column sums of (u - fp16(u))^2 of the start block's fp16 plane, per sketch slab (kernels_mfma.h, first_verify)
    DevBuf<double> partial, partial_sq, partial_o2, colsum, scal, row_part;  // partial_o2: column sums of squares of the product (a-posteriori stop)  // partial: alpha numerators; partial_sq: column sums of squares
    DevBuf<ExpmPlan> plan_d;
    DevBuf<int> viol_d;
    ExpmPlan* plan_h = nullptr;  // pinned
    ExpmPlan last{};
    size_t bs = 0;      // elements per block
    KernelTimers* kt = nullptr;  // optional
    bool use_blk = false;        // LDS-staged blocked SpMM (blocking.h) instead of the generic gather kernel
    BlkDev blk{};
    const T* val_blk = nullptr;
    int npart = 1;               // partial slabs one SpMM launch writes
    // matrix-core SpMM (kernels_mfma.h): fp32 handles on a blocking with <= 32 rows per block
    bool use_mfma = false;
    MfmaDev mf{};
    int mf_mt = 1;               // row tiles of 32 per block
    DevBuf<unsigned short> planes;  // bf16 hi / lo planes of the basis blocks (same bytes as the fp32 blocks)
    bool mfma_now() const { return use_mfma && use_blk && std::is_same<T, float>::value && (lay.Dpad % 32) == 0 && last_mfma_ok; }
    int spmm_slabs() const { return mfma_now() ? mf.nb : npart; }  // partial slabs the next SpMM launch writes
    // every launch that writes `slabs` slabs of Dpad doubles into partial / partial_o2 checks this first (an undersized buffer would be
    // an out-of-bounds device write)
    int check_slabs(int slabs) const {
        if ((size_t)slabs * lay.Dpad > partial.cap || (size_t)slabs * lay.Dpad > partial_o2.cap)
            return fail(MMW_ERR_STATE, "internal: the per-block slabs of the SpMM outgrew their buffers");
        return MMW_OK;
    }
    bool last_mfma_ok = true;    // what the last plan the host has seen said (the matrix starts at zero)
    bool* blk_stale = nullptr;   // owner's flag: val_blk lags the CSR values
    std::function<int()> blk_refresh;  // rebuilds val_blk from the CSR values
    T* rownorm_d = nullptr;      // optional: the combination also emits ||y_row||^2 and its per-block sums (nblk slabs)
    double* rownorm_part = nullptr;
    unsigned short* out_planes = nullptr;  // optional (fp32): the combination also writes the result as bf16 hi / lo halves, interleaved per 32 columns (k_sddmm_mfma)
    bool kt_exact = false;                 // the kernel timers are on in their synchronous mode (mmw_set_profile(s, 1))
    bool planes_only = false;              // ... and nothing else: no one reads this application's fp32 result
    bool start_colsq_ready = false;  // the producer of the start block already filled `partial` with its column sums of squares (npart_start slabs)
    int npart_start = 0;
    // the a-posteriori stop rides on the shifted Lanczos epilogue of the half-tile and of the generic SpMM (not the full-tile one)
    bool apost_off = getenv("MMW_NO_APOST") != nullptr;  // read when the handle is created
    bool apost() const { return (!use_blk || blk.half_tile) && method == MMW_EXPM_LANCZOS && !apost_off; }
    int kbegin(int slot) { return kt ? kt->begin(slot) : MMW_OK; }
    int kend() { return kt ? kt->end() : MMW_OK; }

    ~ExpmEngine() {
        if (plan_h) (void)hipHostFree(plan_h);
    }
    int init(hipStream_t s, int K_, int D, const int* ip, const int* ci, const T* v) {
        st = s;
        K = K_;
        indptr = ip;
        col = ci;
        val = v;
        return resize(D);
    }
    // (re)allocate every width-dependent buffer for a block of D columns
    int resize(int D) {
        std::string err;
        if (make_layout(D, V16<T>::N, lay, err) != MMW_OK) return fail(MMW_ERR_ARG, err);
        bs = (size_t)K * lay.Dpad;
        nblk = grid_slabs(K);   // generic SpMM (alpha slabs)
        nwide = grid_rows(K);   // row kernels with scalar partials (1-norm bound, combination)
        ublocks = 0;
        for (auto& r : planes_ready) r = false;
        MMW_TRY(ensure_blocks(4));  // the basis grows on demand: the MMW loop rarely needs more than 3 vectors
        MMW_TRY(Tm.alloc(bs));
        // slabs one SpMM launch may write: the generic kernel's, the LDS-staged blocking's, the matrix-core blocking's (mf.nb is set once,
        // by the owner's setup_blocking, and survives every later resize)
        const size_t slabs = std::max((size_t)MAX_PART, std::max(use_blk ? (size_t)blk.nb : (size_t)0, use_mfma ? (size_t)mf.nb : (size_t)0));
        MMW_TRY(partial.alloc(slabs * lay.Dpad));
        MMW_TRY(partial_sq.alloc((size_t)MAX_PART * lay.Dpad));
        MMW_TRY(partial_du.alloc((size_t)MAX_PART * lay.Dpad));
        MMW_TRY(partial_o2.alloc(slabs * lay.Dpad));
        npart = generic_slabs<T>(K, lay, nblk);
        MMW_TRY(colsum.alloc(lay.Dpad));
        MMW_TRY(scal.alloc((size_t)4 * (MAX_ORDER + 2) * lay.Dpad));
        MMW_TRY(row_part.alloc((size_t)3 * ROW_GRID_MAX));
        MMW_TRY(plan_d.alloc(1));
        MMW_HIP(hipMemsetAsync(plan_d.p, 0, sizeof(ExpmPlan), st));
        MMW_TRY(reset_plan_history(false));
        MMW_TRY(viol_d.alloc(1));
        MMW_HIP(hipMemsetAsync(viol_d.p, 0, sizeof(int), st));
        if (!plan_h) MMW_HIP(hipHostMalloc((void**)&plan_h, sizeof(ExpmPlan)));
        if (use_blk) MMW_TRY(enable_blocking(blk, val_blk));
        return MMW_OK;
    }
    int ublocks = 0;  // blocks currently allocated in U
    // at least `n` basis blocks; contents need not survive (every application rebuilds the basis from block 0,
    // which the caller fills AFTER asking for the capacity it may need)
    int ensure_blocks(int n) {
        if (n <= ublocks) return MMW_OK;
        if (n > MAX_ORDER + 1) n = MAX_ORDER + 1;
        if (U.p && U.cap >= bs * (size_t)n) {  // room left from a wider run of this handle: no allocation, the start block stays where it is
            if ((size_t)n > (size_t)std::max(ublocks, 1))
                MMW_HIP(hipMemsetAsync(U.p + bs * (size_t)std::max(ublocks, 1), 0, bs * (size_t)(n - std::max(ublocks, 1)) * sizeof(T), st));
            if (ublocks == 0) MMW_HIP(hipMemsetAsync(U.p, 0, bs * sizeof(T), st));
            U.n = bs * (size_t)n;
            ublocks = n;
            return MMW_OK;
        }
        DevBuf<T> bigger;
        MMW_TRY(bigger.alloc(bs * (size_t)n));
        MMW_HIP(hipMemsetAsync(bigger.p, 0, bs * (size_t)n * sizeof(T), st));
        if (U.p && ublocks > 0) MMW_HIP(hipMemcpyAsync(bigger.p, U.p, bs * sizeof(T), hipMemcpyDeviceToDevice, st));  // keep the start block
        MMW_HIP(hipStreamSynchronize(st));
        std::swap(U.p, bigger.p);
        std::swap(U.n, bigger.n);
        std::swap(U.cap, bigger.cap);
        ublocks = n;
        return MMW_OK;
    }
    T* start_block() { return U.p; }
    T* block(int j) { return U.p + (size_t)j * bs; }  // U_j lives in block j-1
    LanczosScalars scalars() {
        LanczosScalars S;
        const size_t n = (size_t)(MAX_ORDER + 2) * lay.Dpad;
        S.alpha = scal.p;
        S.beta = scal.p + n;
        S.sinv = scal.p + 2 * n;
        S.coef = scal.p + 3 * n;
        return S;
    }

    // bf16 hi / lo planes of basis block `idx` (block 0 = start block), made by a pass of their own unless the block's producer
    // wrote them (planes_ready)
    unsigned short* planes_of(int idx) { return planes.p + (size_t)idx * 2 * bs; }
    bool planes_ready[MAX_ORDER + 2] = {false};
    bool planes0_f16 = false;  // the start block's planes hold ONE fp16 plane (the first-order product's operand) instead of bf16 hi / lo
    int reserve_planes() {  // at creation: the planes of the Krylov blocks the matrix-core products read
        if (planes.n < (size_t)ublocks * 2 * bs) MMW_TRY(planes.alloc((size_t)ublocks * 2 * bs));
        for (auto& r : planes_ready) r = false;
        return MMW_OK;
    }
    int ensure_planes() {
        if (planes.n < (size_t)ublocks * 2 * bs) {
            MMW_HIP(hipStreamSynchronize(st));
            MMW_TRY(planes.alloc((size_t)ublocks * 2 * bs));
            for (auto& r : planes_ready) r = false;
        }
        return MMW_OK;
    }
    // where the producer of the start block should write its planes (nullptr: the next product does not run on the matrix cores)
    unsigned short* start_planes() {
        if (!mfma_now() || ensure_planes() != MMW_OK) return nullptr;
        return planes_of(0);
    }
    int make_planes(int idx) {
        if constexpr (std::is_same<T, float>::value) {
            MMW_TRY(ensure_planes());
            if (idx == 0 && planes0_f16) planes_ready[0] = false;  // written for the first-order product: not the two bf16 halves
            if (planes_ready[idx]) return MMW_OK;
            if (idx == 0) planes0_f16 = false;
            MMW_TRY(kbegin(KT_KRYLOV_VEC));
            hipLaunchKernelGGL(k_split_planes, dim3(grid_elems(bs / 4)), dim3(BLOCK), 0, st, bs / 4, reinterpret_cast<const float4*>(block(idx)),
                               reinterpret_cast<uint2*>(planes_of(idx)), reinterpret_cast<uint2*>(planes_of(idx) + bs));
            MMW_HIP(hipGetLastError());
            return kend();
        }
        return MMW_OK;
    }
    template <int MODE> int launch_spmm(const T* in, T* out, T* F, double ascale, double shift, double inv_k,
                                        const ExpmPlan* plan = nullptr, int step = 0, const unsigned short* planes_in = nullptr) {
        if constexpr (std::is_same<T, float>::value && (MODE == SPMM_PLAIN || MODE == SPMM_LANCZOS)) {
            if (planes_in && mfma_now()) {
                MMW_TRY(check_slabs(mf.nb));
                hipEvent_t ea = nullptr, eb = nullptr;
                if (kt) MMW_TRY(kt->begin_attached(KT_SPMM, &ea, &eb));
                MMW_TRY((spmm_mfma_launch<MODE>(st, mf, mf_mt, lay.Dpad, bs * sizeof(unsigned short), reinterpret_cast<const char*>(planes_in), in, out, ascale,
                                                shift, partial.p, apost() ? partial_o2.p : nullptr, plan, step, viol_d.p, MfEpi{}, nullptr, ea, eb)));
                return kend();
            }
        }
        if (use_blk && blk_stale && *blk_stale) {  // the owner left the blocked copy of the values behind while the matrix cores ran
            MMW_TRY(blk_refresh());
            *blk_stale = false;
        }
        MMW_TRY(kbegin(KT_SPMM));
        if (use_blk)
            MMW_TRY((spmm_blk_launch<T, MODE>(st, blk, lay.Dpad, val_blk, in, out, F, nullptr, ascale, shift, inv_k, partial.p, plan, step,
                                              apost() ? partial_o2.p : nullptr)));
        else
            MMW_TRY((spmm_launch<T, MODE>(st, K, lay, nblk, indptr, col, val, in, out, F, nullptr, ascale, shift, inv_k, partial.p, plan, step,
                                          apost() ? partial_o2.p : nullptr, npart)));
        return kend();
    }
    int enable_blocking(const BlkDev& b, const T* values_blocked) {
        blk = b;
        val_blk = values_blocked;
        use_blk = true;
        npart = b.nb;
        if ((size_t)b.nb > (size_t)MAX_PART) {
            MMW_TRY(partial.alloc((size_t)b.nb * lay.Dpad));
            MMW_TRY(partial_o2.alloc((size_t)b.nb * lay.Dpad));
        }
        return MMW_OK;
    }
    template <int OP> int colreduce(int nb, int j, const ExpmPlan* plan) {
        const double eps = sizeof(T) == 4 ? 1e-6 : 1e-14;
        hipLaunchKernelGGL((k_colreduce<OP>), dim3((lay.Dpad + 15) / 16), dim3(1024), 0, st, nb, lay.Dpad, partial.p, colsum.p, j, eps, scalars(), plan);
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }

    // plan from the current values.
    // m_launch == 0: read the plan back (one stream sync) and launch exactly what it asks for;
    // m_launch  > 0: no readback -- the caller launches m_launch single-substep stages and the kernels
    //                themselves skip the stages beyond the device-side order (viol is raised if it needs more).
    int plan_iter = 0;  // iteration index of the matrix `val` currently holds (the owner keeps it; stand-alone uses stay at 0)
    int reset_plan_history(bool keep_sums) {
        hipLaunchKernelGGL(k_plan_reset, dim3(1), dim3(64), 0, st, plan_d.p, keep_sums ? 1 : 0);
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    int make_plan(double ascale, int m_launch) {
        hipLaunchKernelGGL((k_rowsums<T>), dim3(nwide), dim3(BLOCK), 0, st, K, indptr, col, val, ascale, row_part.p);
        hipLaunchKernelGGL(k_plan, dim3(1), dim3(PLAN_THREADS), 0, st, K, method, max_order, tol, row_part.p, nwide, plan_d.p, m_launch, viol_d.p, apost() ? 1 : 0,
                           plan_iter);
        MMW_HIP(hipGetLastError());
        if (m_launch > 0) return MMW_OK;
        MMW_HIP(hipMemcpyAsync(plan_h, plan_d.p, sizeof(ExpmPlan), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        last = *plan_h;
        last_mfma_ok = last.mfma_ok != 0;
        if (last.overflow) return fail(MMW_ERR_STATE, "expm: the one-norm of the matrix is too large for max_order (raise max_order)");
        return MMW_OK;
    }
    // read the last plan and the sticky violation flag (synchronises the stream)
    int fetch_plan(int* violated) {
        int v = 0;
        MMW_HIP(hipMemcpyAsync(plan_h, plan_d.p, sizeof(ExpmPlan), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipMemcpyAsync(&v, viol_d.p, sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        last = *plan_h;
        last_mfma_ok = last.mfma_ok != 0;
        if (getenv("MMW_VERBOSE")) {
            union { unsigned u; float f; } c1, c2;
            c1.u = last.conv[1]; c2.u = last.conv[2];
            fprintf(stderr, "[plan] rho %.3e absn %.3e mfma_ok %d tol %.1e m %d (a-priori %d) apost %d m_eff %d est[1] %.3e est[2] %.3e viol %d\n", last.rho,
                    last.absn, last.mfma_ok, last.tol, last.m, last.m_apriori, last.apost, last.m_eff, c1.f, c2.f, v);
        }
        if (violated) *violated = v;
        return MMW_OK;
    }
    int clear_violation() {
        MMW_HIP(hipMemsetAsync(viol_d.p, 0, sizeof(int), st));
        return MMW_OK;
    }

    // The whole exponential in ONE product (fp32 handles on the matrix cores; the loop's optimistic chunks while one Lanczos step is
    // accepted with room): y = u + (ascale A - mu I) u, certified afterwards by first_order_bound from the column sums this launch
    // leaves in partial_o2 (k_sddmm_mfma's verification workgroup).  No scalar launch, no combination: y leaves the product's
    // epilogue as the SDDMM's planes (and as fp32 in `out` when given), its row norms as fixed-point totals in `dfx` (zero at launch)
    // and the trace as one share per workgroup in `tr_part` (*ntr entries).  A uniform factor e^mu is dropped: X = y y^T / tr.
    // afrag16 != nullptr: the matrix as ONE fp16 half in an image of its own (SPMM_FIRST16; the caller's LOSS pass wrote it)
    int apply_first(T* out, double ascale, int m_launch, bool plan_made, unsigned short* y_planes, long long* dfx, double* tr_part, int* ntr,
                    const unsigned short* afrag16 = nullptr) {
        if constexpr (std::is_same<T, float>::value) {
            if (!plan_made) MMW_TRY(make_plan(ascale, m_launch));
            MMW_TRY(ensure_planes());
            if (!(planes_ready[0] && planes0_f16)) {  // the sketch's producer did not leave the fp16 plane
                MMW_TRY(kbegin(KT_KRYLOV_VEC));
                hipLaunchKernelGGL(k_plane_f16, dim3(grid_elems(bs / 4)), dim3(BLOCK), 0, st, bs / 4, reinterpret_cast<const float4*>(U.p),
                                   reinterpret_cast<uint2*>(planes_of(0)));
                MMW_HIP(hipGetLastError());
                MMW_TRY(kend());
            }
            MfEpi E;
            E.y_planes = y_planes; E.dfx = dfx; E.tr_part = tr_part;
            MMW_TRY(check_slabs(mf.nb));
            hipEvent_t ea = nullptr, eb = nullptr;
            if (kt) MMW_TRY(kt->begin_attached(KT_SPMM, &ea, &eb));
            if (afrag16) {
                MfmaDev m16 = mf;
                m16.afrag = reinterpret_cast<const unsigned*>(afrag16);
                MMW_TRY((spmm_mfma_launch<SPMM_FIRST16>(st, m16, mf_mt, lay.Dpad, bs * sizeof(unsigned short), reinterpret_cast<const char*>(planes_of(0)), U.p, out,
                                                        ascale / (double)MF_F16_SCALE, 0.0, partial.p, partial_o2.p, plan_d.p, 1, viol_d.p, E, ntr, ea, eb)));
            } else
                MMW_TRY((spmm_mfma_launch<SPMM_FIRST>(st, mf, mf_mt, lay.Dpad, bs * sizeof(unsigned short), reinterpret_cast<const char*>(planes_of(0)), U.p, out,
                                                      ascale / (double)MF_F16_SCALE, 0.0, partial.p, partial_o2.p, plan_d.p, 1, viol_d.p, E, ntr, ea, eb)));
            MMW_TRY(kend());
            planes_ready[0] = false;
            planes0_f16 = false;
            start_colsq_ready = false;
            return MMW_OK;
        }
        return fail(MMW_ERR_STATE, "the first-order product runs on fp32 matrix-core handles only");
    }
    int first_grid_max() const { return (mf.nb + 7) / 8 * 8 * (lay.Dpad / 32); }  // upper bound of apply_first's *ntr

    // out = exp(ascale*A) * start_block().  `out` must not alias the engine's blocks.
    // plan_made: the caller's own kernels already left this application's (lagged) plan in plan_d
    int apply(T* out, double ascale, int m_launch = 0, bool plan_made = false) {
        if (!plan_made) MMW_TRY(make_plan(ascale, m_launch));
        const int m = m_launch > 0 ? m_launch : last.m;
        const int nsub = m_launch > 0 ? 1 : last.nsub;
        MMW_TRY(ensure_blocks(std::max(3, m)));
        ExpmPlan* pd = plan_d.p;
        const int Dpad = lay.Dpad;
        const int gcol = (Dpad + 63) / 64;
        const int gel = grid_elems(bs);
        const size_t shcol = (size_t)BLOCK * sizeof(double);
        for (int sub = 0; sub < nsub; ++sub) {
            if (sub > 0) {
                hipLaunchKernelGGL((k_copy<T>), dim3(gel), dim3(BLOCK), 0, st, bs, out, U.p);
            }
            if (method == MMW_EXPM_LANCZOS) {
                LanczosScalars S = scalars();
                const int gr = grid_slabs(K * 4);  // k_colsq / k_lz_update stride rows by workgroup
                const double eps = sizeof(T) == 4 ? 1e-6 : 1e-14;
                int nsq = npart_start;  // slabs of column sums of squares waiting in partial_sq
                if (!(sub == 0 && start_colsq_ready)) {
                    MMW_TRY(kbegin(KT_KRYLOV_VEC));
                    hipLaunchKernelGGL((k_colsq<T>), dim3(gr), dim3(BLOCK), shcol, st, K, Dpad, U.p, partial_sq.p);
                    MMW_TRY(kend());
                    nsq = gr;
                }
                start_colsq_ready = false;
                for (int j = 1; j <= m; ++j) {
                    const bool mf_step = mfma_now();
                    if (mf_step) MMW_TRY(make_planes(j - 1));
                    MMW_TRY((launch_spmm<SPMM_LANCZOS>(block(j - 1), Tm.p, nullptr, ascale, 0.0, 1.0, pd, j, mf_step ? planes_of(j - 1) : nullptr)));
                    planes_ready[j - 1] = false;  // consumed; the block is rewritten by the next application
                    MMW_TRY(kbegin(KT_KRYLOV_VEC));
                    // one launch: the norms of U_j, alpha_j, and after the last product the small exponentials
#define MMW_LZS(NMAX)                                                                                                               \
    hipLaunchKernelGGL((k_lz_scalars<NMAX>), dim3((Dpad + LZS_COLS - 1) / LZS_COLS), dim3(1024), 0, st, mf_step ? mf.nb : npart, partial.p, \
                       apost() ? partial_o2.p : (double*)nullptr, nsq, partial_sq.p, Dpad, j, m, 1.0 / nsub, eps, S, pd)
                    if (j + 2 <= 4) MMW_LZS(4);
                    else if (j + 2 <= 8) MMW_LZS(8);
                    else MMW_LZS(MAX_ORDER + 2);
#undef MMW_LZS
                    if (kt && kt->on && kt_exact && apost() && j < m) {  // profiling mode 1 counts exact launches: look at the estimate before going on
                        MMW_TRY(kend());
                        MMW_HIP(hipMemcpyAsync(plan_h, plan_d.p, sizeof(ExpmPlan), hipMemcpyDeviceToHost, st));
                        MMW_HIP(hipStreamSynchronize(st));
                        union { float f; unsigned u; } tb;
                        tb.f = (float)plan_h->tol;
                        if (plan_h->apost && plan_h->conv[j] <= tb.u) break;
                        MMW_TRY(kbegin(KT_KRYLOV_VEC));
                    }
                    if (j < m) {  // the last product A U_m goes straight into the combination (corrected scheme)
                        unsigned short* pl_next = nullptr;  // U_{j+1} lives in block j
                        if (mf_step && ensure_planes() == MMW_OK && j < ublocks) pl_next = planes_of(j);
                        hipLaunchKernelGGL((k_lz_update<T>), dim3(gr), dim3(BLOCK), shcol, st, K, Dpad, j, Tm.p, block(j - 1),
                                           j > 1 ? block(j - 2) : block(j - 1), block(j), S, partial_sq.p, pd, pl_next);
                        planes_ready[j] = pl_next != nullptr;
                        nsq = gr;
                    }
                    MMW_TRY(kend());
                }
                MMW_TRY(kbegin(KT_KRYLOV_VEC));
                const bool last = sub + 1 == nsub;
                hipLaunchKernelGGL((k_lz_combine<T>), dim3(nwide), dim3(BLOCK), 0, st, K, Dpad, m, U.p, bs, Tm.p, S.coef, out, pd,
                                   last ? rownorm_d : (T*)nullptr, last ? rownorm_part : (double*)nullptr, viol_d.p, last ? out_planes : (unsigned short*)nullptr,
                                   last && out_planes && planes_only ? 1 : 0);
                MMW_TRY(kend());
            } else {
                hipLaunchKernelGGL((k_copy<T>), dim3(gel), dim3(BLOCK), 0, st, bs, U.p, out);
                for (int k = 1; k <= m; ++k) {
                    T* in = k == 1 ? U.p : block((k - 1) % 2 + 1);
                    T* o = block(k % 2 + 1);
                    MMW_TRY((launch_spmm<SPMM_TAYLOR>(in, o, out, ascale / nsub, last.mu / nsub, 1.0 / k, pd, k)));
                }
                hipLaunchKernelGGL((k_scale<T>), dim3(gel), dim3(BLOCK), 0, st, bs, out, 1.0, pd);
            }
            MMW_HIP(hipGetLastError());
        }
        return MMW_OK;
    }
};

}  // namespace mmw
