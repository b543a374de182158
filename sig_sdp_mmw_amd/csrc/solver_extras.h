// Rounding, epilogue factor and duality-gap routines that hang off a solver handle.
#pragma once
#include "factor.h"
#include "kernels_round.h"
#include "pattern.h"
#include "runtime.h"

namespace mmw {

template <typename T> struct Extras {
    hipStream_t st = nullptr;
    const HostPattern* H = nullptr;
    KernelTimers* kt = nullptr;
    int K = 0;
    // rounding-side state (float64 whatever T is)
    DevBuf<int> so_indptr, so_indices, q_indptr, q_indices;
    DevBuf<double> so_data, h_max, so_hmax;  // so_hmax[e] = h_max[so_indices[e]]: the greedy reads it like so_data
    DevBuf<GreedyHdr> ghdr;
    DevBuf<double> gX, randv, P, gain, nrm;
    DevBuf<int> pref, slot, order, rem, glag, rank_part, gsched, gnsteps;
    DevBuf<unsigned> gmask;
    DevBuf<GreedyHdr> ghdr_s;  // [steps][GB_WAVES]: the users' headers in schedule order
    Factorizer<T> fac;
    // gap work
    DevBuf<T> g_x, g_l, g_y, g_e1, g_e2, g_r, g_vec, g_tm;
    DevBuf<double> g_part, g_scal, g_lz, g_colsum;

    int maxdeg_h = 1, maxq_h = 1;  // longest row of S_gain without its diagonal / of Q (the greedy kernels' LDS budget)
    // the rounding's state straight from the device generator (mmw_create_from_env): so_* / q_* / h_max are filled by the caller's kernels
    int init_device(hipStream_t s, int K_, KernelTimers* k, int maxdeg, int maxq) {
        st = s; H = nullptr; K = K_; kt = k;
        maxdeg_h = std::max(1, maxdeg); maxq_h = std::max(1, maxq);
        MMW_TRY(ghdr.alloc((size_t)K));
        return fac.init(st, K, kt);
    }
    int init(hipStream_t s, const HostPattern* h, int K_, KernelTimers* k) {
        st = s; H = h; K = K_; kt = k;
        for (int q = 0; q < K; ++q) {
            maxdeg_h = std::max(maxdeg_h, H->so_indptr[q + 1] - H->so_indptr[q]);
            maxq_h = std::max(maxq_h, H->q_indptr[q + 1] - H->q_indptr[q]);
        }
        MMW_TRY(so_indptr.upload(H->so_indptr, st));
        MMW_TRY(so_indices.upload(H->so_indices, st));
        MMW_TRY(so_data.upload(H->so_data, st));
        {
            std::vector<double> hm(H->so_indices.size());
            for (size_t e = 0; e < hm.size(); ++e) hm[e] = H->h_max[H->so_indices[e]];
            MMW_TRY(so_hmax.upload(hm, st));
        }
        MMW_TRY(ghdr.alloc((size_t)K));
        MMW_TRY(q_indptr.upload(H->q_indptr, st));
        MMW_TRY(q_indices.upload(H->q_indices, st));
        MMW_TRY(h_max.upload(H->h_max, st));
        MMW_HIP(hipStreamSynchronize(st));
        return fac.init(st, K, kt);
    }
    template <typename B> static int ensure(DevBuf<B>& b, size_t n) {
        if (b.n >= n && b.p) return MMW_OK;
        return b.alloc(n);
    }

    // ---- LOG_GAP (mmw.py:79-117): nterms = number of X / Y terms in the running sums
    int gap(const PatternDev<T>& P, const int* lrow, const T* xavg, const T* yavg, int nterms, double out[3]) {
        const size_t nnz = (size_t)P.nnzL, C = (size_t)P.C;
        const int gr = grid_rows(K);
        MMW_TRY(ensure(g_x, nnz)); MMW_TRY(ensure(g_l, nnz)); MMW_TRY(ensure(g_y, C)); MMW_TRY(ensure(g_e1, C)); MMW_TRY(ensure(g_e2, C));
        MMW_TRY(ensure(g_r, K)); MMW_TRY(ensure(g_part, ROW_GRID_MAX)); MMW_TRY(ensure(g_scal, 8));
        const double inv = 1.0 / (double)nterms;
        hipLaunchKernelGGL((k_scaled_copy<T>), dim3(grid_elems(nnz)), dim3(BLOCK), 0, st, nnz, xavg, inv, g_x.p);
        hipLaunchKernelGGL((k_scaled_copy<T>), dim3(grid_elems(C)), dim3(BLOCK), 0, st, C, yavg, inv, g_y.p);
        // violations at Xbar: reuse the DUAL kernels with a zero accumulator and eta = 1 -> block maxima of e
        MMW_HIP(hipMemsetAsync(g_e2.p, 0, C * sizeof(T), st));
        hipLaunchKernelGGL((k_dual_rows<T>), dim3(gr), dim3(BLOCK), 0, st, P, g_x.p, g_r.p, g_e1.p);
        hipLaunchKernelGGL((k_dual_h<T>), dim3(gr), dim3(BLOCK), 0, st, P, g_r.p, g_e1.p, g_e2.p, 1.0, g_part.p);
        hipLaunchKernelGGL(k_max_reduce, dim3(1), dim3(BLOCK), 0, st, gr, g_part.p, g_scal.p + 4);
        // L(Ybar) on the pattern
        hipLaunchKernelGGL((k_ysums<T>), dim3(1), dim3(BLOCK), 0, st, K, P.E_asso, g_y.p, P.cH, P.inv_norm_H, g_scal.p);
        MMW_HIP(hipMemsetAsync(g_l.p, 0, nnz * sizeof(T), st));
        hipLaunchKernelGGL((k_hweights<T>), dim3(grid_elems((size_t)K)), dim3(BLOCK), 0, st, K, g_y.p + (K + P.E_asso), P.inv_norm_H, g_r.p);  // g_r is free again
        hipLaunchKernelGGL((k_loss<T>), dim3(gr), dim3(BLOCK), 0, st, P, lrow, g_y.p, g_r.p, g_scal.p, g_l.p, -1.0, (const int*)nullptr, (T*)nullptr);
        MMW_HIP(hipGetLastError());
        double emax = 0.0;
        MMW_HIP(hipMemcpyAsync(&emax, g_scal.p + 4, sizeof(double), hipMemcpyDeviceToHost, st));
        double lam = 0.0;
        MMW_TRY(lambda_min(P.indptr, P.col, g_l.p, &lam));
        MMW_HIP(hipStreamSynchronize(st));
        out[0] = emax;
        out[1] = lam * (double)K;
        out[2] = out[0] - out[1];
        return MMW_OK;
    }

    // smallest eigenvalue of the symmetric matrix (pattern, val): plain Lanczos (the extreme Ritz value of the
    // three-term recurrence converges to lambda_min with or without reorthogonalisation), tridiagonal on the host
    int lambda_min(const int* indptr, const int* col, const T* val, double* lam) {
        BlockLayout lay;
        std::string err;
        MMW_TRY(make_layout(1, V16<T>::N, lay, err));
        const int Dp = lay.Dpad;
        const size_t bs = (size_t)K * Dp;
        const int m_max = std::min(K, 600), chunk = 20;
        MMW_TRY(ensure(g_vec, 3 * bs)); MMW_TRY(ensure(g_tm, bs));
        MMW_TRY(ensure(g_lz, (size_t)4 * (m_max + 3) * Dp));
        MMW_TRY(ensure(g_colsum, Dp));
        DevBuf<double>& part = fac.partial;
        if (part.n < (size_t)MAX_PART * Dp) MMW_TRY(part.alloc((size_t)MAX_PART * std::max(Dp, 4)));
        LanczosScalars S;
        const size_t n = (size_t)(m_max + 3) * Dp;
        S.alpha = g_lz.p; S.beta = g_lz.p + n; S.sinv = g_lz.p + 2 * n; S.coef = g_lz.p + 3 * n;
        const int nblk = grid_slabs(K), gr = grid_slabs(K * 4);
        const size_t shcol = (size_t)BLOCK * sizeof(double);
        auto blk = [&](int j) { return g_vec.p + (size_t)((j - 1) % 3) * bs; };
        const double eps = sizeof(T) == 4 ? 1e-6 : 1e-14;
        // deterministic start vector: Philox normals, one column
        hipLaunchKernelGGL((k_sketch_rng<T>), dim3(nblk), dim3(BLOCK), 0, st, K, 1, Dp, 0x6a09e667f3bcc908ull, 7u, blk(1), (double*)nullptr);
        // rows are normalised to +-1 by the sketch kernel; that is a fine start vector
        hipLaunchKernelGGL((k_colsq<T>), dim3(gr), dim3(BLOCK), shcol, st, K, Dp, blk(1), part.p);
        hipLaunchKernelGGL((k_colreduce<LZ_INIT>), dim3((Dp + 15) / 16), dim3(1024), 0, st, gr, Dp, part.p, g_colsum.p, 0, eps, S, (const ExpmPlan*)nullptr);
        std::vector<double> a, b, ha((size_t)(m_max + 3) * Dp), hb((size_t)(m_max + 3) * Dp);
        const double tol = sizeof(T) == 4 ? 1e-5 : 1e-11;
        double theta = 0.0;
        int j = 1;
        while (j <= m_max) {
            const int jend = std::min(m_max, j + chunk - 1);
            for (; j <= jend; ++j) {
                MMW_TRY((spmm_launch<T, SPMM_LANCZOS>(st, K, lay, nblk, indptr, col, val, blk(j), g_tm.p, nullptr, nullptr, 1.0, 0.0, 1.0, part.p)));
                hipLaunchKernelGGL((k_colreduce<LZ_ALPHA>), dim3((Dp + 15) / 16), dim3(1024), 0, st, nblk, Dp, part.p, g_colsum.p, j, eps, S, (const ExpmPlan*)nullptr);
                hipLaunchKernelGGL((k_lz_update<T>), dim3(gr), dim3(BLOCK), shcol, st, K, Dp, j, g_tm.p, blk(j), j > 1 ? blk(j - 1) : blk(j), blk(j + 1), S, part.p, (const ExpmPlan*)nullptr);
                hipLaunchKernelGGL((k_colreduce<LZ_BETA>), dim3((Dp + 15) / 16), dim3(1024), 0, st, gr, Dp, part.p, g_colsum.p, j, eps, S, (const ExpmPlan*)nullptr);
            }
            MMW_HIP(hipGetLastError());
            const int m = j - 1;
            MMW_HIP(hipMemcpyAsync(ha.data(), S.alpha, (size_t)(m + 1) * Dp * sizeof(double), hipMemcpyDeviceToHost, st));
            MMW_HIP(hipMemcpyAsync(hb.data(), S.beta, (size_t)(m + 1) * Dp * sizeof(double), hipMemcpyDeviceToHost, st));
            MMW_HIP(hipStreamSynchronize(st));
            a.assign(m, 0.0);
            b.assign(m, 0.0);
            int mm = m;
            for (int i = 1; i <= m; ++i) {
                a[i - 1] = ha[(size_t)i * Dp];
                b[i - 1] = hb[(size_t)i * Dp];  // beta_i couples i and i+1
                if (i < m && b[i - 1] == 0.0) {  // invariant subspace found
                    mm = i;
                    break;
                }
            }
            double lastc = 0.0;
            theta = tridiag_min_eig(a, b, mm, &lastc);
            double scale = 0.0;
            for (int i = 0; i < mm; ++i) scale = std::max(scale, std::fabs(a[i]) + (i < mm - 1 ? std::fabs(b[i]) : 0.0));
            const double resid = std::fabs(b[mm - 1] * lastc);
            if (mm < m || resid <= tol * std::max(scale, 1e-300) || m >= K) break;
        }
        *lam = theta;
        return MMW_OK;
    }

    int factor(const int* indptr, const int* col, const T* xavg, int nit, int32_t rank, double* out, uint64_t seed) {
        return fac.run(indptr, col, xavg, 1.0 / (double)nit, rank, seed, out);
    }
    int read_factor(double* out, int64_t n) {
        if ((int64_t)fac.last_n != n || n == 0) return fail(MMW_ERR_STATE, "mmw_read_f64(FACTOR): no factor of that size has been computed");
        MMW_TRY(fac.fetch_last(fac.last_n));
        memcpy(out, fac.last_host.p, fac.last_n * sizeof(double));
        return MMW_OK;
    }

    // handle creation: the buffers the factor and one batched rounding call of (Z, D', nbatch) would size on first use
    int fac_reserve(hipStream_t s, int K_, int rank, bool mf_possible) {
        fac.st = s; fac.K = K_; fac.dw.st = s;
        return fac.reserve(rank, mf_possible);
    }
    PinnedBuf stage;  // host side of mmw_round's transfers
    int round_reserve(int K_, int32_t Z, int32_t Dp, int32_t nb) {
        MMW_TRY(stage.ensure((size_t)K_ * Dp * sizeof(double) + (size_t)nb * Z * Dp * sizeof(double) + (size_t)nb * K_ * sizeof(int) + 8 + (size_t)nb * sizeof(int)));
        const size_t nP = (size_t)nb * K_ * Z;
        MMW_TRY(ensure(gX, (size_t)K_ * Dp)); MMW_TRY(ensure(randv, (size_t)nb * Z * Dp));
        MMW_TRY(ensure(P, nP)); MMW_TRY(ensure(pref, nP)); MMW_TRY(ensure(gain, nP));
        MMW_TRY(ensure(slot, (size_t)nb * K_)); MMW_TRY(ensure(nrm, K_)); MMW_TRY(ensure(order, K_)); MMW_TRY(ensure(rem, nb));
        MMW_TRY(ensure(glag, (size_t)K_));
        MMW_TRY(ensure(rank_part, (size_t)RANK_SPLIT * K_));
        return MMW_OK;
    }

    int round(int32_t Z, int32_t Dp, const double* gX_h, int32_t nb, const double* randv_h, int32_t* z_out, int32_t* rem_out) {
        if (Z < 1 || Dp < 1 || nb < 1) return fail(MMW_ERR_ARG, "mmw_round: Z, D' and nbatch must be positive");
        if (!randv_h || !z_out || !rem_out) return fail(MMW_ERR_ARG, "mmw_round: null pointer");
        // gX == nullptr: the factor this handle computed last, where mmw_factor left it on the device (no copy out and in again)
        if (!gX_h && ((size_t)K * Dp != fac.last_n || Dp != fac.last_rank || fac.last_n == 0))
            return fail(MMW_ERR_STATE, "mmw_round: gX is null and the handle holds no factor of K x D'");
        if ((size_t)Z * sizeof(int) > 60000) return fail(MMW_ERR_ARG, "mmw_round: Z too large");
        const size_t nP = (size_t)nb * K * Z;
        if (gX_h) MMW_TRY(ensure(gX, (size_t)K * Dp));
        const double* gX_d = gX_h ? gX.p : fac.out64.p;
        MMW_TRY(ensure(randv, (size_t)nb * Z * Dp));
        MMW_TRY(ensure(P, nP));
        MMW_TRY(ensure(pref, nP));
        MMW_TRY(ensure(gain, nP));
        MMW_TRY(ensure(slot, (size_t)nb * K));
        MMW_TRY(ensure(nrm, K));
        MMW_TRY(ensure(order, K));
        MMW_TRY(ensure(rank_part, (size_t)RANK_SPLIT * K));
        MMW_TRY(ensure(rem, nb));
        // inputs and outputs cross through the handle's page-locked staging buffer (runtime.h, PinnedBuf)
        const size_t b_gx = gX_h ? (size_t)K * Dp * sizeof(double) : 0, b_rv = (size_t)nb * Z * Dp * sizeof(double);
        const size_t b_z = (((size_t)nb * K * sizeof(int)) + 7) & ~(size_t)7, b_rem = (size_t)nb * sizeof(int);
        MMW_TRY(stage.ensure(b_gx + b_rv + b_z + b_rem));
        char* sp = static_cast<char*>(stage.p);
        if (gX_h) memcpy(sp, gX_h, b_gx);
        memcpy(sp + b_gx, randv_h, b_rv);
        if (gX_h) MMW_HIP(hipMemcpyAsync(gX.p, sp, b_gx, hipMemcpyHostToDevice, st));
        MMW_HIP(hipMemcpyAsync(randv.p, sp + b_gx, b_rv, hipMemcpyHostToDevice, st));
        MMW_HIP(hipMemsetAsync(gain.p, 0, nP * sizeof(double), st));
        MMW_HIP(hipMemsetAsync(slot.p, 0xFF, (size_t)nb * K * sizeof(int), st));
        if (kt) MMW_TRY(kt->begin(KT_PROJECT));
        hipLaunchKernelGGL(k_row_norms_f64, dim3(grid_rows(K)), dim3(BLOCK), 0, st, K, Dp, gX_d, nrm.p);
        hipLaunchKernelGGL(k_rank_count, dim3((K + BLOCK - 1) / BLOCK, RANK_SPLIT), dim3(BLOCK), 0, st, K, nrm.p, rank_part.p);
        hipLaunchKernelGGL(k_rank_scatter, dim3((K + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, K, (const int*)rank_part.p, order.p);
        hipLaunchKernelGGL(k_project_mfma, dim3((K + 63) / 64, (Z + 15) / 16, nb), dim3(BLOCK), 0, st, K, Z, Dp, gX_d, randv.p, P.p);
        hipLaunchKernelGGL(k_slot_pref, dim3(K, nb), dim3(BLOCK), (size_t)Z * sizeof(double), st, K, Z, P.p, pref.p);
        if (kt) MMW_TRY(kt->end());
        MMW_HIP(hipGetLastError());
        if (kt) MMW_TRY(kt->begin(KT_GREEDY));
        {
            int maxdeg = maxdeg_h, maxq = maxq_h;
            maxdeg = (maxdeg + 1) / 2 * 2;  // keep the int arrays after the doubles 8-byte aligned
            if (getenv("MMW_GREEDY_SEQ") && (maxdeg > 4 * BLOCK || maxq > 4 * BLOCK || Z > 4 * BLOCK))
                return fail(MMW_ERR_ARG, "mmw_round: more than 1024 neighbours / slots per user is not supported by the sequential greedy kernel");
            hipLaunchKernelGGL(k_greedy_headers, dim3(grid_elems((size_t)K)), dim3(BLOCK), 0, st, K, order.p, so_indptr.p, q_indptr.p, h_max.p, ghdr.p);
            static const bool sequential = getenv("MMW_GREEDY_SEQ") != nullptr;  // the one-user-per-step kernel, kept for comparison
            static const bool runs_only = getenv("MMW_GREEDY_RUNS") != nullptr;   // contiguous runs instead of the out-of-order schedule
            if (!sequential && !runs_only) {
                // steps scheduled out of order (kernels_round.h, k_greedy_schedule): interaction masks of the visiting order, the schedule,
                // the headers step by step; then one workgroup per attempt follows it
                MMW_TRY(ensure(gmask, (size_t)K));
                MMW_TRY(ensure(gsched, (size_t)K * GB_WAVES));
                MMW_TRY(ensure(gnsteps, 1));
                MMW_TRY(ensure(ghdr_s, (size_t)K * GB_WAVES));  // (at most K steps)
                MMW_HIP(hipMemsetAsync(gmask.p, 0, (size_t)K * sizeof(unsigned), st));
                const size_t pairs = (size_t)K * GS_W;
                hipLaunchKernelGGL(k_greedy_cmask, dim3((unsigned)((pairs + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, K, (const GreedyHdr*)ghdr.p,
                                   so_indices.p, q_indices.p, gmask.p);
                hipLaunchKernelGGL(k_greedy_schedule, dim3(1), dim3(WAVE), 0, st, K, (const unsigned*)gmask.p, gsched.p, gnsteps.p);
                hipLaunchKernelGGL(k_greedy_sched_headers, dim3(grid_elems((size_t)K * 2)), dim3(BLOCK), 0, st, K, (const int*)gnsteps.p, (const int*)gsched.p,
                                   (const GreedyHdr*)ghdr.p, ghdr_s.p);
                if (getenv("MMW_VERBOSE")) {
                    int ns = 0;
                    MMW_HIP(hipMemcpyAsync(&ns, gnsteps.p, sizeof(int), hipMemcpyDeviceToHost, st));
                    MMW_HIP(hipStreamSynchronize(st));
                    fprintf(stderr, "[round] %d users in %d steps (%.1f per step)\n", K, ns, (double)K / std::max(ns, 1));
                }
                const size_t baseb = (size_t)GB_WAVES * Z * 4;
                if (baseb > 150 * 1024) return fail(MMW_ERR_ARG, "mmw_round: 64 Z bytes exceed the greedy kernel's LDS");
                const bool slot_lds = baseb + (size_t)K * 4 <= 150 * 1024;
                const size_t sh = baseb + (slot_lds ? (size_t)K * 4 : 0);
                if (slot_lds) {
                    MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_greedy_b<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                    hipLaunchKernelGGL((k_greedy_b<true, true>), dim3(nb), dim3(GB_WAVES * 64), sh, st, K, Z, (const GreedyHdr*)ghdr_s.p, (const int*)nullptr, pref.p,
                                       so_indices.p, so_data.p, so_hmax.p, q_indices.p, gain.p, slot.p, rem.p, (const int*)gnsteps.p);
                } else {
                    MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_greedy_b<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                    hipLaunchKernelGGL((k_greedy_b<false, true>), dim3(nb), dim3(GB_WAVES * 64), sh, st, K, Z, (const GreedyHdr*)ghdr_s.p, (const int*)nullptr, pref.p,
                                       so_indices.p, so_data.p, so_hmax.p, q_indices.p, gain.p, slot.p, rem.p, (const int*)gnsteps.p);
                }
            } else if (!sequential) {
                // several mutually non-interacting users per step (k_greedy_b): interaction lags of the visiting order first
                MMW_TRY(ensure(glag, (size_t)K));
                MMW_HIP(hipMemsetAsync(glag.p, 0x7F, (size_t)K * sizeof(int), st));
                const size_t pairs = (size_t)K * (GB_WAVES - 1);
                hipLaunchKernelGGL(k_greedy_conflicts, dim3((unsigned)((pairs + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, K, (const GreedyHdr*)ghdr.p,
                                   so_indices.p, q_indices.p, glag.p);
                const size_t baseb = (size_t)GB_WAVES * Z * 4 + (((size_t)K + 3) & ~(size_t)3);
                if (baseb > 150 * 1024) return fail(MMW_ERR_ARG, "mmw_round: K + 32 Z bytes exceed the greedy kernel's LDS");
                const bool slot_lds = baseb + (size_t)K * 4 <= 150 * 1024;
                const size_t sh = baseb + (slot_lds ? (size_t)K * 4 : 0);
                if (slot_lds) {
                    MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_greedy_b<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                    hipLaunchKernelGGL((k_greedy_b<true>), dim3(nb), dim3(GB_WAVES * 64), sh, st, K, Z, (const GreedyHdr*)ghdr.p, (const int*)glag.p, pref.p,
                                       so_indices.p, so_data.p, so_hmax.p, q_indices.p, gain.p, slot.p, rem.p);
                } else {
                    MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_greedy_b<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                    hipLaunchKernelGGL((k_greedy_b<false>), dim3(nb), dim3(GB_WAVES * 64), sh, st, K, Z, (const GreedyHdr*)ghdr.p, (const int*)glag.p, pref.p,
                                       so_indices.p, so_data.p, so_hmax.p, q_indices.p, gain.p, slot.p, rem.p);
                }
            } else {
            const size_t base = (size_t)2 * ((size_t)maxdeg * 20 + (size_t)maxq * 4 + (size_t)Z * 4) + (size_t)Z * 4;
            const bool slot_lds = base + (size_t)K * 4 <= 150 * 1024;
            const size_t sh = base + (slot_lds ? (size_t)K * 4 : 0);
            if (sh > 160 * 1024) return fail(MMW_ERR_ARG, "mmw_round: a user's neighbour list does not fit the greedy kernel's LDS record");
            if (slot_lds) {
                MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_greedy<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                hipLaunchKernelGGL((k_greedy<true>), dim3(nb), dim3(BLOCK), sh, st, K, Z, maxdeg, maxq, (const GreedyHdr*)ghdr.p, pref.p, so_indices.p,
                                   so_data.p, so_hmax.p, q_indices.p, gain.p, slot.p, rem.p);
            } else {
                MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_greedy<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                hipLaunchKernelGGL((k_greedy<false>), dim3(nb), dim3(BLOCK), sh, st, K, Z, maxdeg, maxq, (const GreedyHdr*)ghdr.p, pref.p, so_indices.p,
                                   so_data.p, so_hmax.p, q_indices.p, gain.p, slot.p, rem.p);
            }
            }
        }
        if (kt) MMW_TRY(kt->end());
        MMW_HIP(hipGetLastError());
        MMW_HIP(hipMemcpyAsync(sp + b_gx + b_rv, slot.p, (size_t)nb * K * sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipMemcpyAsync(sp + b_gx + b_rv + b_z, rem.p, b_rem, hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        memcpy(z_out, sp + b_gx + b_rv, (size_t)nb * K * sizeof(int));
        memcpy(rem_out, sp + b_gx + b_rv + b_z, b_rem);
        return MMW_OK;
    }
};

}  // namespace mmw
