// Host orchestration of
//   * the epilogue factor (sim_src/alg/mmw.py:202-216): top-`rank` (largest |eigenvalue| = singular value)
//     invariant subspace of the averaged X by Chebyshev-filtered block subspace iteration on A^2 with
//     Rayleigh-Ritz, all linear algebra on the device (CSR SpMM + fp64-MFMA Gram / tall GEMM + Jacobi);
//   * the LOG_GAP branch (mmw.py:79-117): max violation at Xbar and K * lambda_min(L(Ybar)) by a device
//     Lanczos recurrence whose small tridiagonal is examined on the host (Sturm bisection).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>

#include "expm_engine.h"
#include "kernels_dense.h"
#include "kernels_loop.h"
#include "runtime.h"

namespace mmw {

// smallest eigenvalue of the symmetric tridiagonal (a[0..n), b[0..n-1)) by bisection, and the residual
// bound |b_last * s_n| of its Ritz vector (s = eigenvector of T, by inverse iteration).
inline double tridiag_min_eig(const std::vector<double>& a, const std::vector<double>& b, int n, double* last_comp) {
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < n; ++i) {
        const double r = (i > 0 ? std::fabs(b[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(b[i]) : 0.0);
        lo = std::min(lo, a[i] - r);
        hi = std::max(hi, a[i] + r);
    }
    auto count_below = [&](double x) {  // number of eigenvalues < x (Sturm sequence)
        int cnt = 0;
        double q = a[0] - x;
        if (q < 0) ++cnt;
        for (int i = 1; i < n; ++i) {
            const double den = std::fabs(q) < 1e-300 ? (q < 0 ? -1e-300 : 1e-300) : q;
            q = a[i] - x - b[i - 1] * b[i - 1] / den;
            if (q < 0) ++cnt;
        }
        return cnt;
    };
    double l = lo, h = hi;
    for (int it = 0; it < 200 && h - l > 4e-16 * std::max(std::fabs(l), std::fabs(h)); ++it) {
        const double m = 0.5 * (l + h);
        if (count_below(m) >= 1) h = m;
        else l = m;
    }
    const double theta = 0.5 * (l + h);
    if (last_comp) {  // inverse iteration (T - theta' I) s = s_prev, Thomas algorithm with a tiny shift
        std::vector<double> s(n, 1.0 / std::sqrt((double)n)), d(n), c(n), rhs(n);
        const double sh = theta - 1e-10 * std::max(1.0, std::fabs(hi - lo));
        for (int rep = 0; rep < 3; ++rep) {
            rhs = s;
            double piv = a[0] - sh;
            if (std::fabs(piv) < 1e-300) piv = 1e-300;
            d[0] = piv;
            for (int i = 1; i < n; ++i) {
                c[i - 1] = b[i - 1] / d[i - 1];
                d[i] = a[i] - sh - c[i - 1] * b[i - 1];
                if (std::fabs(d[i]) < 1e-300) d[i] = 1e-300;
                rhs[i] -= c[i - 1] * rhs[i - 1];
            }
            s[n - 1] = rhs[n - 1] / d[n - 1];
            for (int i = n - 2; i >= 0; --i) s[i] = (rhs[i] - b[i] * s[i + 1]) / d[i];
            double nn = 0;
            for (double v : s) nn += v * v;
            nn = std::sqrt(nn);
            for (double& v : s) v /= nn;
        }
        *last_comp = s[n - 1];
    }
    return theta;
}

template <typename T> struct DenseWork {
    hipStream_t st = nullptr;
    DevBuf<double> G, Q, Q2, G2, Q3, Gpart, cs, diag, dscale, off, scale;
    DevBuf<int> perm, flag;
    DevBuf<double> dfac;  // the current panel's factored diagonal block (k_chol_panel -> k_chol_update)
    DevBuf<double> dinv;  // the inverses of all diagonal blocks of the last factor (k_chol_panel -> k_trsm_blocked)
    const bool chol_lds = getenv("MMW_CHOL_LDS") != nullptr;  // the panel's diagonal block factored through LDS (for comparison)
    int bcap = 0;
    int sweeps_total = 0, calls_total = 0;
    int ensure(int b, int nslice) {
        if (b <= bcap && Gpart.n >= (size_t)nslice * b * b) return MMW_OK;
        bcap = std::max(b, bcap);
        const size_t bb = (size_t)bcap * bcap;
        MMW_TRY(G.alloc(bb)); MMW_TRY(Q.alloc(bb)); MMW_TRY(Q2.alloc(bb)); MMW_TRY(Gpart.alloc((size_t)nslice * bb));
        MMW_TRY(cs.alloc(bcap + 2)); MMW_TRY(diag.alloc(bcap)); MMW_TRY(dscale.alloc(bcap)); MMW_TRY(off.alloc(2));
        MMW_TRY(scale.alloc(bcap)); MMW_TRY(perm.alloc(bcap));
        return MMW_OK;
    }
    static void gram_geometry(int K, int b, int& nslice, int& rps) {
        const int tiles = ((b + 63) / 64) * ((b + 15) / 16);
        nslice = std::max(1, std::min(16, (1024 + tiles - 1) / tiles));
        rps = (K + nslice - 1) / nslice;
        rps = (rps + 3) / 4 * 4;
        nslice = (K + rps - 1) / rps;
    }
    // G = V^T W  (b x b)
    int gram(int K, int b, int ld, const T* V, const T* W, bool sym) {
        int nslice, rps;
        MMW_TRY(ensure(b, 16));
        static const bool old_gram = getenv("MMW_GRAM_WAVES") != nullptr;  // (the one-wave-per-tile kernel, for comparison)
        if (old_gram) {
            gram_geometry(K, b, nslice, rps);
            hipLaunchKernelGGL((k_gram<T>), dim3((b + 63) / 64, (b + 15) / 16, nslice), dim3(WAVE), 0, st, K, b, ld, V, W, rps, Gpart.p);
        } else {
            const int tiles = ((b + 63) / 64) * ((b + 63) / 64);
            nslice = std::max(1, std::min(16, (768 + tiles - 1) / tiles));
            rps = ((K + nslice - 1) / nslice + GR_ROWS - 1) / GR_ROWS * GR_ROWS;
            nslice = (K + rps - 1) / rps;
            hipLaunchKernelGGL((k_gram_tiles<T>), dim3((b + 63) / 64, (b + 63) / 64, nslice), dim3(BLOCK), 0, st, K, b, ld, V, W, rps, Gpart.p);
        }
        hipLaunchKernelGGL(k_gram_reduce, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, nslice, Gpart.p, G.p, sym ? 1 : 0);
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    // Jacobi of G (b x b) -> eigenvalues in `diag`, eigenvectors in Q (one launch per round, ping-pong buffers)
    DevBuf<double> bjH[2], bjQ[2], bjR;
    int reserve_jacobi(int b) {
        int M = (b + BJ_NB - 1) / BJ_NB;
        M = std::max(2, M + (M & 1));
        const size_t pp = (size_t)BJ_NB * M * BJ_NB * M;
        for (int q = 0; q < 2; ++q) {
            if (bjH[q].n < pp) MMW_TRY(bjH[q].alloc(pp));
            if (bjQ[q].n < pp) MMW_TRY(bjQ[q].alloc(pp));
        }
        if (bjR.n < (size_t)(M / 2) * BJ_N2 * BJ_N2) MMW_TRY(bjR.alloc((size_t)(M / 2) * BJ_N2 * BJ_N2));
        if (flag.n < 1) MMW_TRY(flag.alloc(1));
        if (dfac.n < (size_t)CH_NB * CH_NB) MMW_TRY(dfac.alloc((size_t)CH_NB * CH_NB));
        return MMW_OK;
    }
    // block Jacobi (kernels_dense.h, k_bj_solve / k_bj_apply): two launches per block round, M - 1 block rounds per sweep
    int jacobi_blocked(int b, double rel_tol, int max_sweeps, int* sweeps_done) {
        int M = (b + BJ_NB - 1) / BJ_NB;
        M = std::max(2, M + (M & 1));
        const int P = BJ_NB * M;
        const size_t pp = (size_t)P * P;
        for (int q = 0; q < 2; ++q) {
            if (bjH[q].n < pp) MMW_TRY(bjH[q].alloc(pp));
            if (bjQ[q].n < pp) MMW_TRY(bjQ[q].alloc(pp));
        }
        if (bjR.n < (size_t)(M / 2) * BJ_N2 * BJ_N2) MMW_TRY(bjR.alloc((size_t)(M / 2) * BJ_N2 * BJ_N2));
        MMW_TRY(set_max_lds(reinterpret_cast<const void*>(&k_bj_apply), BJ_APPLY_LDS));
        static const bool all_full = getenv("MMW_BJ_FULL") != nullptr;  // every meeting solves the whole 64 x 64 problem
        hipLaunchKernelGGL(k_bj_pad, dim3(grid_elems(pp)), dim3(BLOCK), 0, st, b, P, G.p, bjH[0].p, bjQ[0].p);
        int cur = 0, sw = 0;
        double h_off[2] = {0, 0};
        const int half = M / 2;
        for (; sw < max_sweeps; ++sw) {
            hipLaunchKernelGGL(k_offdiag, dim3(1), dim3(1024), 0, st, P, bjH[cur].p, off.p);
            MMW_HIP(hipMemcpyAsync(h_off, off.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            MMW_HIP(hipStreamSynchronize(st));
            if (!(std::sqrt(h_off[0]) > rel_tol * h_off[1] * std::sqrt((double)b)) || b < 2) break;
            for (int r = 0; r < M - 1; ++r) {
                hipLaunchKernelGGL(k_bj_solve, dim3(half), dim3(1024), 0, st, M, P, r, (r == 0 || all_full) ? 1 : 0, bjH[cur].p, bjR.p, rel_tol);
                hipLaunchKernelGGL(k_bj_apply, dim3(half * half + (P / BJ_N2) * half), dim3(BLOCK), BJ_APPLY_LDS, st, M, P, r, bjH[cur].p,
                                   bjH[cur ^ 1].p, bjQ[cur].p, bjQ[cur ^ 1].p, bjR.p);
                cur ^= 1;
            }
            MMW_HIP(hipGetLastError());
        }
        if (sweeps_done) *sweeps_done = sw;
        sweeps_total += sw;
        calls_total += 1;
        hipLaunchKernelGGL(k_bj_unpad, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, P, bjH[cur].p, bjQ[cur].p, diag.p, Q.p);
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    int jacobi(int b, double rel_tol, int max_sweeps, int* sweeps_done = nullptr) {
        static const bool blocked = getenv("MMW_JACOBI_ROUNDS") == nullptr;
        if (blocked) return jacobi_blocked(b, rel_tol, max_sweeps, sweeps_done);
        const int n = (b % 2 == 0) ? b : b + 1;
        if (b <= JAC_LDS_MAX) {  // small: the whole eigensolve in one launch, matrices in LDS
            const size_t sh = ((size_t)2 * b * b + n + 2) * sizeof(double);
            MMW_TRY(set_max_lds(reinterpret_cast<const void*>(&k_jacobi_lds), (int)(((size_t)2 * JAC_LDS_MAX * JAC_LDS_MAX + JAC_LDS_MAX + 4) * sizeof(double))));
            hipLaunchKernelGGL(k_jacobi_lds, dim3(1), dim3(1024), sh, st, b, G.p, diag.p, Q.p, rel_tol, max_sweeps, (int*)nullptr);
            MMW_HIP(hipGetLastError());
            calls_total += 1;
            if (sweeps_done) *sweeps_done = 0;
            return MMW_OK;
        }
        if (G2.n < (size_t)b * b) { MMW_TRY(G2.alloc((size_t)bcap * bcap)); MMW_TRY(Q3.alloc((size_t)bcap * bcap)); }
        hipLaunchKernelGGL(k_set_eye, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, Q.p);
        const int half = n / 2;
        const int ga = grid_elems((size_t)std::max(half * half, b * half));
        double h_off[2] = {0, 0};
        double* Hc = G.p;
        double* Hn = G2.p;
        double* Qc = Q.p;
        double* Qn = Q3.p;
        int sw = 0;
        for (; sw < max_sweeps; ++sw) {
            hipLaunchKernelGGL(k_offdiag, dim3(1), dim3(1024), 0, st, b, Hc, off.p);
            MMW_HIP(hipMemcpyAsync(h_off, off.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            MMW_HIP(hipStreamSynchronize(st));
            if (!(std::sqrt(h_off[0]) > rel_tol * h_off[1] * std::sqrt((double)b)) || b < 2) break;
            for (int r = 0; r < n - 1; ++r) {
                hipLaunchKernelGGL(k_jacobi_round, dim3(ga), dim3(BLOCK), 0, st, b, n, r, Hc, Hn, Qc, Qn);
                std::swap(Hc, Hn);
                std::swap(Qc, Qn);
            }
            MMW_HIP(hipGetLastError());
        }
        if (sweeps_done) *sweeps_done = sw;
        sweeps_total += sw;
        calls_total += 1;
        hipLaunchKernelGGL(k_get_diag, dim3(grid_elems(b)), dim3(BLOCK), 0, st, b, Hc, diag.p);
        if (Qc != Q.p) MMW_HIP(hipMemcpyAsync(Q.p, Qc, (size_t)b * b * sizeof(double), hipMemcpyDeviceToDevice, st));
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    // Cholesky of the unit-diagonal Gram matrix D G D in place (G <- L); *ok = false when it is not numerically SPD
    int chol_factor(int b, bool* ok) {
        if (flag.n < 1) MMW_TRY(flag.alloc(1));
        if (dfac.n < (size_t)CH_NB * CH_NB) MMW_TRY(dfac.alloc((size_t)CH_NB * CH_NB));
        const size_t n_inv = (size_t)((b + CH_NB - 1) / CH_NB) * CH_NB * CH_NB;
        if (dinv.n < n_inv) MMW_TRY(dinv.alloc(n_inv));
        hipLaunchKernelGGL(k_scale_sym, dim3(grid_elems(b)), dim3(BLOCK), 0, st, b, G.p, dscale.p);
        hipLaunchKernelGGL(k_apply_scale_sym, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, G.p, dscale.p);
        MMW_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
        for (int j0 = 0; j0 < b; j0 += CH_NB) {
            const int below = b - std::min(b, j0 + CH_NB);
            hipLaunchKernelGGL(k_chol_panel, dim3(std::max(1, (below + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, b, j0, G.p, flag.p, dfac.p, chol_lds ? 1 : 0, dinv.p);
            if (below > 0) {
                const int mt = (below + CH_NB - 1) / CH_NB;
                hipLaunchKernelGGL(k_chol_update, dim3(mt, mt), dim3(BLOCK), 0, st, b, j0, G.p, (const int*)flag.p, (const double*)dfac.p);
            }
        }
        int bad = 0;
        MMW_HIP(hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        *ok = bad == 0;
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
    // Vout = V * Qmat[b x n]
    int gemm(int K, int b, int n, int ld, const T* V, const double* Qmat, int ldq, T* Vout) {
        hipLaunchKernelGGL((k_gemm_tall<T>), dim3((K + 63) / 64, (n + 63) / 64), dim3(BLOCK), 0, st, K, b, n, ld, V, ldq, Qmat, ld, Vout);
        MMW_HIP(hipGetLastError());
        return MMW_OK;
    }
};

template <typename T> struct Factorizer {
    hipStream_t st = nullptr;
    KernelTimers* kt = nullptr;
    int K = 0;
    DenseWork<T> dw;
    DevBuf<T> V, W, Y1, Y2;      // K x ld blocks
    DevBuf<double> partial, colsum, rho_part, out64;
    PinnedBuf last_host;         // [K*rank] float64: the last factor, page-locked (read back by MMW_F_FACTOR too)
    size_t last_n = 0;
    int last_rank = 0;
    bool last_on_host = false;   // last_host holds the factor that sits in out64

    int outer_done = 0;
    double last_resid = 0.0;

    int init(hipStream_t s, int K_, KernelTimers* k) {
        st = s; K = K_; kt = k; dw.st = s;
        return MMW_OK;
    }
    // optional locality blocking of the pattern (same structure the MMW loop uses): the factor's SpMMs then run on
    // the LDS-staged kernel with the matrix values gathered once into the blocked traversal order
    bool have_blk = false;
    BlkDev blk{};
    const int* bepos = nullptr;
    int64_t nent = 0;
    DevBuf<T> val_blk;
    void set_blocking(const BlkDev& b, const int* blocked_to_csr, int64_t entries) {
        blk = b; bepos = blocked_to_csr; nent = entries; have_blk = true;
    }
    // optional matrix-core blocking (fp32): the Chebyshev filter's products run on k_spmm_mfma while the block is still far
    // from converged (see run()); the matrix goes once per call into a fragment image of its own
    bool have_mf = false;
    MfmaDev mf{};
    int mf_mt = 1;
    const int* mf_fpos = nullptr;
    size_t mf_image = 0;
    int64_t mf_nnz = 0;
    DevBuf<unsigned> afrag;
    void set_mfma(const MfmaDev& m, int mt, const int* fpos, size_t image_words, int64_t nnz) {
        mf = m; mf_mt = mt; mf_fpos = fpos; mf_image = image_words; mf_nnz = nnz; have_mf = true;
    }
    template <int MODE>
    int spmm(const BlockLayout& lay, int nblk, const int* indptr, const int* col, const T* val, const T* in, T* outp, T* Fp, const T* X2p,
             double c1, double c2, double c3) {
        if (have_blk) return spmm_blk_launch<T, MODE>(st, blk, lay.Dpad, val_blk.p, in, outp, Fp, X2p, c1, c2, c3, nullptr);
        return spmm_launch<T, MODE>(st, K, lay, nblk, indptr, col, val, in, outp, Fp, X2p, c1, c2, c3, nullptr);
    }

    // the matrix-core form of the two products above (fp32 only): `in` is read through its planes, the result is written with its own
    size_t mf_bs = 0;  // K * ld of the running call
    unsigned short* planes_of(T* block) const { return reinterpret_cast<unsigned short*>(block + mf_bs); }
    int split(T* block) {
        if constexpr (sizeof(T) == 4) {
            hipLaunchKernelGGL(k_split_planes, dim3(grid_elems(mf_bs / 4)), dim3(BLOCK), 0, st, mf_bs / 4, reinterpret_cast<const float4*>(block),
                               reinterpret_cast<uint2*>(planes_of(block)), reinterpret_cast<uint2*>(planes_of(block) + mf_bs));
            MMW_HIP(hipGetLastError());
        }
        return MMW_OK;
    }
    template <int MODE>
    int spmm_mf(int ld, T* in, T* outp, const T* Fp, const T* X2p, double c1, double c2, double c3) {
        if constexpr (sizeof(T) == 4) {
            MfEpi e;
            e.F = Fp; e.X2 = X2p; e.c3 = (float)c3; e.out_planes = planes_of(outp);
            return spmm_mfma_launch<MODE>(st, mf, mf_mt, ld, mf_bs * sizeof(unsigned short), reinterpret_cast<const char*>(planes_of(in)), in, outp, c1, c2,
                                          nullptr, nullptr, nullptr, 0, nullptr, e);
        }
        return fail(MMW_ERR_STATE, "matrix-core products are fp32 only");
    }

    // V <- V * F twice with F from the Gram matrix: Cholesky-QR (cheap) and, when the block is too
    // ill-conditioned for it, the eigen-based factor D Q Lambda^{-1/2}; columns orthonormal to rounding
    int orthonormalise(int b, int ld, DevBuf<T>& A, DevBuf<T>& B, double eps) {
        // tolerated deviation from the identity for the one-term form of the second pass: its result is off by 3/8 ||E||^2
        const double ns_max = sizeof(T) == 4 ? 1.6e-3 : 5e-7;
        static const bool ns_on = getenv("MMW_FACTOR_NO_NS") == nullptr;
        for (int pass = 0; pass < 2; ++pass) {
            MMW_TRY(dw.gram(K, b, ld, A.p, A.p, true));
            if (pass == 1 && ns_on) {
                // After the first pass G = I + E with a small E: V (I - E/2) is orthonormal to 3/8 ||E||^2 -- one small kernel and a
                // tall GEMM instead of a Cholesky factorisation (b / 32 dependent panel steps) and a row substitution.
                double dev2 = 0.0;
                hipLaunchKernelGGL(k_dev_from_identity, dim3(1), dim3(1024), 0, st, b, dw.G.p, dw.off.p);
                MMW_HIP(hipMemcpyAsync(&dev2, dw.off.p, sizeof(double), hipMemcpyDeviceToHost, st));
                MMW_HIP(hipStreamSynchronize(st));
                if (std::sqrt(dev2) <= ns_max) {
                    hipLaunchKernelGGL(k_ns_first, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, dw.G.p, dw.Q2.p);
                    MMW_HIP(hipMemsetAsync(B.p, 0, (size_t)K * ld * sizeof(T), st));
                    MMW_TRY(dw.gemm(K, b, b, ld, A.p, dw.Q2.p, b, B.p));
                    std::swap(A.p, B.p);
                    continue;
                }
            }
            bool ok = false;
            MMW_TRY(dw.chol_factor(b, &ok));
            if (!ok) {
                MMW_TRY(dw.gram(K, b, ld, A.p, A.p, true));  // the failed factorisation overwrote G
                hipLaunchKernelGGL(k_scale_sym, dim3(grid_elems(b)), dim3(BLOCK), 0, st, b, dw.G.p, dw.dscale.p);
                hipLaunchKernelGGL(k_apply_scale_sym, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, dw.G.p, dw.dscale.p);
                MMW_TRY(dw.jacobi(b, 1e-14, 30));
                hipLaunchKernelGGL(k_orth_factor, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, dw.Q.p, dw.dscale.p, dw.diag.p, eps,
                                   (double)b, dw.Q2.p);
                MMW_HIP(hipMemsetAsync(B.p, 0, (size_t)K * ld * sizeof(T), st));
                MMW_TRY(dw.gemm(K, b, b, ld, A.p, dw.Q2.p, b, B.p));
            } else {  // V <- (V D) L^{-T} by forward substitution on the rows
                static const bool trsm_rows = getenv("MMW_TRSM_ROWS") != nullptr;  // one wavefront per row (for comparison)
                if (trsm_rows)
                    hipLaunchKernelGGL((k_trsm_rows<T>), dim3(grid_rows(K)), dim3(BLOCK), (size_t)WAVES_PER_BLOCK * b * sizeof(double), st, K, b, ld,
                                       A.p, dw.G.p, dw.dscale.p, B.p);
                else
                    hipLaunchKernelGGL((k_trsm_blocked<T>), dim3((K + 63) / 64), dim3(BLOCK), 0, st, K, b, ld, (const T*)A.p, (const double*)dw.G.p,
                                       (const double*)dw.dscale.p, (const double*)dw.dinv.p, B.p);
                MMW_HIP(hipGetLastError());
            }
            std::swap(A.p, B.p);
        }
        return MMW_OK;
    }

    // block width and padded width run() uses for a given rank
    static int block_width(int K, int rank) {
        int b = std::min(K, rank + std::max(12, rank / 5));
        if (K <= 384 || 4 * b >= 3 * K) b = K;
        return b;
    }
    // the buffers run() sizes on first use, reserved ahead (handle creation)
    int reserve(int rank, bool mf_possible) {
        const int b = block_width(K, rank);
        BlockLayout lay;
        std::string err;
        if (make_layout(b, V16<T>::N, lay, err) != MMW_OK) return MMW_OK;  // run() reports it
        const size_t bs = (size_t)K * lay.Dpad;
        const size_t want = (mf_possible && sizeof(T) == 4 && lay.Dpad % 32 == 0 && b < K) ? 2 * bs : bs;
        if (V.n < want) { MMW_TRY(V.alloc(want)); MMW_TRY(W.alloc(want)); MMW_TRY(Y1.alloc(want)); MMW_TRY(Y2.alloc(want)); }
        if (partial.n < (size_t)MAX_PART * lay.Dpad) MMW_TRY(partial.alloc((size_t)MAX_PART * lay.Dpad));
        if (colsum.n < (size_t)lay.Dpad) MMW_TRY(colsum.alloc(lay.Dpad));
        if (rho_part.n < (size_t)MAX_PART) MMW_TRY(rho_part.alloc(MAX_PART));
        if (out64.n < (size_t)K * rank) MMW_TRY(out64.alloc((size_t)K * rank));
        MMW_TRY(last_host.ensure((size_t)K * rank * sizeof(double)));
        MMW_TRY(dw.ensure(b, 16));
        MMW_TRY(dw.reserve_jacobi(b));
        return MMW_OK;
    }

    // the page-locked host copy of the factor in out64, made when somebody asks for it
    int fetch_last(size_t n) {
        if (last_on_host) return MMW_OK;
        MMW_TRY(last_host.ensure(n * sizeof(double)));
        MMW_HIP(hipMemcpyAsync(last_host.p, out64.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        last_on_host = true;
        return MMW_OK;
    }
    // factor of A = ascale * (values `val` on the pattern).  out: K*rank float64 (host), or nullptr
    int run(const int* indptr, const int* col, const T* val, double ascale, int rank, uint64_t seed, double* out) {
        if (rank < 1 || rank >= K + 1) return fail(MMW_ERR_ARG, "mmw_factor: rank must be in [1, K]");
        auto vnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double v_t0 = vnow();
        const bool f32 = sizeof(T) == 4;
        const double tol = f32 ? 2e-5 : 1e-9;
        const int b = block_width(K, rank);  // small or nearly full: one exact Rayleigh-Ritz on the whole space
        BlockLayout lay;
        std::string err;
        if (make_layout(b, V16<T>::N, lay, err) != MMW_OK) return fail(MMW_ERR_ARG, "mmw_factor: " + err);
        const int ld = lay.Dpad;
        const size_t bs = (size_t)K * ld;
        const int nblk = grid_slabs(K);
        // matrix-core filter passes: every block carries its bf16 hi / lo planes right behind its fp32 values
        bool mf_use = false;
        if constexpr (sizeof(T) == 4) mf_use = have_mf && have_blk && b < K && ld % 32 == 0 && !getenv("MMW_FACTOR_NO_MFMA");
        const size_t bs_alloc = mf_use ? 2 * bs : bs;
        mf_bs = bs;
        if (V.n < bs_alloc) { MMW_TRY(V.alloc(bs_alloc)); MMW_TRY(W.alloc(bs_alloc)); MMW_TRY(Y1.alloc(bs_alloc)); MMW_TRY(Y2.alloc(bs_alloc)); }
        if (partial.n < (size_t)MAX_PART * ld) MMW_TRY(partial.alloc((size_t)MAX_PART * ld));
        if (colsum.n < (size_t)ld) MMW_TRY(colsum.alloc(ld));
        if (rho_part.n < (size_t)MAX_PART) MMW_TRY(rho_part.alloc(MAX_PART));
        MMW_TRY(dw.ensure(b, 16));
        if (getenv("MMW_FACTOR_VERBOSE")) { (void)hipStreamSynchronize(st); fprintf(stderr, "[factor]   allocations done at %.1f ms\n", (vnow() - v_t0) * 1e3); }
        if (kt) MMW_TRY(kt->begin(KT_FACTOR));
        if (have_blk) {  // matrix values once into the blocked order (padding entries stay zero)
            if (val_blk.n < (size_t)nent) MMW_TRY(val_blk.alloc((size_t)nent));
            hipLaunchKernelGGL((k_gather_blocked<T>), dim3(grid_elems((size_t)nent)), dim3(BLOCK), 0, st, (size_t)nent, bepos, val, val_blk.p);
            if (getenv("MMW_FACTOR_VERBOSE")) { (void)hipStreamSynchronize(st); fprintf(stderr, "[factor]   gather blocked at %.1f ms\n", (vnow() - v_t0) * 1e3); }
        }
        if constexpr (sizeof(T) == 4) {
            if (mf_use) {
                if (afrag.n < mf_image) MMW_TRY(afrag.alloc(mf_image));
                MMW_HIP(hipMemsetAsync(afrag.p, 0, mf_image * sizeof(unsigned), st));
                if (getenv("MMW_FACTOR_VERBOSE")) { (void)hipStreamSynchronize(st); fprintf(stderr, "[factor]   memset image at %.1f ms\n", (vnow() - v_t0) * 1e3); }
                hipLaunchKernelGGL((k_refrag<T>), dim3(grid_elems((size_t)mf_nnz)), dim3(BLOCK), 0, st, (size_t)mf_nnz, val, mf_fpos, afrag.p);
                if (getenv("MMW_FACTOR_VERBOSE")) { (void)hipStreamSynchronize(st); fprintf(stderr, "[factor]   refrag at %.1f ms\n", (vnow() - v_t0) * 1e3); }
                mf.afrag = afrag.p;
            }
        }
        if (getenv("MMW_FACTOR_VERBOSE")) { (void)hipStreamSynchronize(st); fprintf(stderr, "[factor]   matrix copies done at %.1f ms\n", (vnow() - v_t0) * 1e3); }
        // spectral scale: ||A||_1 >= |lambda|_max
        hipLaunchKernelGGL((k_rowabs<T>), dim3(nblk), dim3(BLOCK), 0, st, K, indptr, col, val, ascale, (const double*)nullptr, 0, rho_part.p);
        std::vector<double> hp(nblk);
        MMW_HIP(hipMemcpyAsync(hp.data(), rho_part.p, nblk * sizeof(double), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        const double rho = *std::max_element(hp.begin(), hp.end());
        if (getenv("MMW_FACTOR_VERBOSE")) { (void)hipStreamSynchronize(st); fprintf(stderr, "[factor]   norm bound done at %.1f ms\n", (vnow() - v_t0) * 1e3); }
        // random start block (rows of unit norm; any full-rank start works)
        hipLaunchKernelGGL((k_sketch_rng<T>), dim3(nblk), dim3(BLOCK), 0, st, K, b, ld, seed ^ 0x9E3779B97F4A7C15ull, 0u, V.p, (double*)nullptr);
        MMW_TRY(orthonormalise(b, ld, V, W, 1e-14));
        if (getenv("MMW_FACTOR_VERBOSE")) { (void)hipStreamSynchronize(st); fprintf(stderr, "[factor]   start block orthonormal at %.1f ms\n", (vnow() - v_t0) * 1e3); }
        std::vector<double> theta(b), res(b), hres((size_t)64 * b);
        std::vector<int> perm(b);
        std::vector<double> ones(b, 1.0);
        MMW_HIP(hipMemcpyAsync(dw.scale.p, ones.data(), b * sizeof(double), hipMemcpyHostToDevice, st));
        const int max_outer = 40;
        int outer = 0;
        bool done = false;
        double prev_resid = -1.0;
        bool mf_stalled = false, mf_last_pass = false;  // the matrix-core split stopped paying / the pass before this residual ran on it
        int degree = 12;  // filter degree schedule 12, 30, 40, 40, ...: few Rayleigh-Ritz / orthonormalisation rounds
        int rr_skip = 0;  // filter passes still to run before the next Rayleigh-Ritz
        bool skip_rr = false;
        for (; outer < max_outer && !done; ++outer) {
            // Matrix-core products (two-half bf16 split, error <= 2.3e-5 || |A| || per product) serve the filter while the last
            // residual says that at least two more passes follow: the subspace error they leave is far below what those passes
            // start from, and every Rayleigh-Ritz product and the last passes run on the fp32 kernel.
            // (floor: 100 x tol until the end of round 3; with 3 x tol the passes end on the same residuals to three digits -- 8.1e-6 / 1.20e-5 /
            // 1.42e-5 / 5.07e-7 at ranks 370 / 208 / 128 / 88 of the benchmark's probes -- in the same number of passes, 2 ms sooner per call.
            // With 1 x tol -- every filter pass there is, since a pass runs only while the residual is above the tolerance -- the last
            // residuals end at 2.6e-6 instead of 5e-7: the split's own floor, an eighth of the tolerance, and the last pass of the small
            // ranks costs half.  A pass on the matrix cores that does not halve a residual below 100 x tol sends the rest to the fp32 kernel.)
            static const double mf_floor = getenv("MMW_FACTOR_MF_FLOOR") ? atof(getenv("MMW_FACTOR_MF_FLOOR")) : 1.0;
            const bool mf_stage = mf_use && !mf_stalled && (outer == 0 || last_resid > mf_floor * tol);
            const bool no_rr_next = rr_skip > 0 && b < K;
            // ---- Rayleigh-Ritz on span(V)
            if (mf_stage && no_rr_next) {  // this product only feeds the filter
                MMW_TRY(split(V.p));
                MMW_TRY((spmm_mf<SPMM_PLAIN>(ld, V.p, W.p, nullptr, nullptr, ascale, 0.0, 0.0)));
            } else
            MMW_TRY((spmm<SPMM_PLAIN>(lay, nblk, indptr, col, val, V.p, W.p, nullptr, nullptr, ascale, 0.0, 0.0)));
            // The Rayleigh-Ritz step (Gram matrix, dense eigensolve, two tall GEMMs, residuals) only rotates the basis and tells how
            // far it is: the filter works on the subspace whatever its basis.  While the last residual says that more than one
            // more filter pass is needed anyway, passes run back to back on the Ritz values of the last Rayleigh-Ritz.
            const bool no_rr = rr_skip > 0 && b < K;
            if (no_rr) --rr_skip;
            else {
            MMW_TRY(dw.gram(K, b, ld, V.p, W.p, true));
            // A random start block spans nothing of interest yet: its Rayleigh quotients (the diagonal of G) are all the first
            // filter needs, and the dense eigensolve of a random projection is the most expensive one of the run (b > 96: ~10
            // Jacobi sweeps of b - 1 launches each).
            // (Seeding the block with the previous probe's Ritz vectors was tried: the averaged X of neighbouring slot counts
            // do not share their leading subspace -- first residual 0.2 either way -- so every call starts from a random block.)
            skip_rr = outer == 0 && b > JAC_LDS_MAX && b < K && !getenv("MMW_FACTOR_FULL_RR");
            if (skip_rr) {
                hipLaunchKernelGGL(k_set_eye, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, dw.Q.p);
                hipLaunchKernelGGL(k_get_diag, dim3(grid_elems(b)), dim3(BLOCK), 0, st, b, dw.G.p, dw.diag.p);
            } else {
                // The eigensolve only has to be as sharp as the subspace is: what it leaves off the diagonal mixes Ritz pairs whose residuals are
                // still `last_resid`, so 1e-2 of that (never looser than 1e-4, never tighter than the final 1e-8 / 1e-13; the residuals that decide are true residuals of the rotated vectors) costs the residual
                // estimate nothing and the block Jacobi a sweep or two per call (b = 444: a sweep is 13 block rounds of two launches).
                static const bool jac_fixed = getenv("MMW_FACTOR_JACOBI_FIXED") != nullptr;
                const double jac_fin = f32 ? 1e-8 : 1e-13;
                static const double jac_rel = getenv("MMW_FACTOR_JACOBI_REL") ? atof(getenv("MMW_FACTOR_JACOBI_REL")) : 1e-2;
                static const double jac_cap = getenv("MMW_FACTOR_JACOBI_CAP") ? atof(getenv("MMW_FACTOR_JACOBI_CAP")) : 1e-4;
                const double jac_tol = (jac_fixed || outer == 0 || !(last_resid > 0.0)) ? jac_fin : std::max(jac_fin, std::min(jac_cap, jac_rel * last_resid));
                MMW_TRY(dw.jacobi(b, jac_tol, 30));
            }
            MMW_HIP(hipMemcpyAsync(theta.data(), dw.diag.p, b * sizeof(double), hipMemcpyDeviceToHost, st));
            MMW_HIP(hipStreamSynchronize(st));
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return std::fabs(theta[x]) > std::fabs(theta[y]); });
            MMW_HIP(hipMemcpyAsync(dw.perm.p, perm.data(), b * sizeof(int), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_select_cols, dim3(grid_elems((size_t)b * b)), dim3(BLOCK), 0, st, b, b, dw.Q.p, dw.perm.p, dw.scale.p, dw.Q2.p);
            MMW_HIP(hipMemsetAsync(Y1.p, 0, bs * sizeof(T), st));
            MMW_HIP(hipMemsetAsync(Y2.p, 0, bs * sizeof(T), st));
            MMW_TRY(dw.gemm(K, b, b, ld, V.p, dw.Q2.p, b, Y1.p));
            MMW_TRY(dw.gemm(K, b, b, ld, W.p, dw.Q2.p, b, Y2.p));
            std::swap(V.p, Y1.p);
            std::swap(W.p, Y2.p);
            std::vector<double> ths(b);
            for (int i = 0; i < b; ++i) ths[i] = theta[perm[i]];
            theta = ths;
            // residual norms ||A v - theta v|| per Ritz pair
            MMW_HIP(hipMemcpyAsync(dw.diag.p, theta.data(), b * sizeof(double), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((k_resid_colsq<T>), dim3(64), dim3(BLOCK), 0, st, K, b, ld, W.p, V.p, dw.diag.p, partial.p);
            MMW_HIP(hipMemcpyAsync(hres.data(), partial.p, (size_t)64 * b * sizeof(double), hipMemcpyDeviceToHost, st));
            MMW_HIP(hipStreamSynchronize(st));
            double worst = 0.0;
            for (int c = 0; c < b; ++c) {
                double s = 0.0;
                for (int q = 0; q < 64; ++q) s += hres[(size_t)q * b + c];
                res[c] = std::sqrt(s);
                if (c < rank) worst = std::max(worst, res[c]);
            }
            const double scale_top = std::max(std::fabs(theta[0]), 1e-300);
            last_resid = worst / scale_top;
            if (mf_last_pass && prev_resid > 0.0 && last_resid < 100.0 * tol && last_resid > 0.5 * prev_resid) mf_stalled = true;
            if (getenv("MMW_FACTOR_VERBOSE")) fprintf(stderr, "[factor]   outer %d degree %d resid %.2e at %.1f ms\n", outer, degree, last_resid, (vnow() - v_t0) * 1e3);
            if (!skip_rr && (b >= K || last_resid <= tol)) {
                done = true;
                break;
            }
            if (!skip_rr && !getenv("MMW_FACTOR_FULL_RR")) rr_skip = last_resid > 1e3 * tol ? 2 : (last_resid > 30.0 * tol ? 1 : 0);
            }  // Rayleigh-Ritz
            // ---- Chebyshev filter on B = A^2 damping [0, cut], cut = smallest Ritz value of B in the block
            const double mu_top = theta[0] * theta[0];
            double cut = theta[b - 1] * theta[b - 1];
            cut = std::max(cut, 1e-12 * mu_top);
            cut = std::min(cut, 0.999 * theta[rank - 1] * theta[rank - 1] + 1e-300);
            const double e = 0.5 * cut, cen = 0.5 * cut;
            const double a0 = skip_rr ? rho * rho : std::max(mu_top, rho * rho * 1e-30);  // no Ritz values yet: the 1-norm bounds the spectrum
            double sigma1 = e / (a0 - cen), sigma = sigma1;
            // Y = sigma1/e (B V - cen V): T1 = A V (already W); Ycur = c1 * A W + c2 * V
            const bool mf_pass = mf_stage && !mf_stalled && (no_rr || last_resid > mf_floor * tol || skip_rr);  // this pass's residual is known by now
            mf_last_pass = mf_pass;
            if (mf_pass) {
                if (!(mf_stage && no_rr)) MMW_TRY(split(W.p));  // W came from the fp32 kernel (or was rotated by the Rayleigh-Ritz step)
                MMW_TRY((spmm_mf<SPMM_AXPBY>(ld, W.p, Y1.p, V.p, V.p, ascale * sigma1 / e, -cen * sigma1 / e, 0.0)));
            } else
            MMW_TRY((spmm<SPMM_AXPBY>(lay, nblk, indptr, col, val, W.p, Y1.p, V.p, V.p, ascale * sigma1 / e, -cen * sigma1 / e, 0.0)));
            // V = previous, Y1 = current
            T* prev = V.p;
            T* cur = Y1.p;
            T* nxt = Y2.p;
            for (int i = 2; i <= degree; ++i) {
                const double sigma2 = 1.0 / (2.0 / sigma1 - sigma);
                // nxt = 2 sigma2/e (A W - cen cur) - sigma sigma2 prev
                if (mf_pass) {
                    MMW_TRY((spmm_mf<SPMM_PLAIN>(ld, cur, W.p, nullptr, nullptr, ascale, 0.0, 0.0)));
                    MMW_TRY((spmm_mf<SPMM_AXPBY>(ld, W.p, nxt, cur, prev, ascale * 2.0 * sigma2 / e, -cen * 2.0 * sigma2 / e, -sigma * sigma2)));
                } else {
                MMW_TRY((spmm<SPMM_PLAIN>(lay, nblk, indptr, col, val, cur, W.p, nullptr, nullptr, ascale, 0.0, 0.0)));
                MMW_TRY((spmm<SPMM_AXPBY>(lay, nblk, indptr, col, val, W.p, nxt, cur, prev, ascale * 2.0 * sigma2 / e, -cen * 2.0 * sigma2 / e,
                                          -sigma * sigma2)));
                }
                T* t = prev;
                prev = cur;
                cur = nxt;
                nxt = t;
                sigma = sigma2;
            }
            // move the filtered block into V, keep the three buffers distinct
            if (cur != V.p) {
                T* oldV = V.p;
                V.p = cur;
                if (cur == Y1.p) Y1.p = oldV; else Y2.p = oldV;
            }
            MMW_TRY(orthonormalise(b, ld, V, W, 1e-14));
            // adaptive degree: grow while the residual keeps falling, back off when rounding stalls the filter
            if (no_rr) degree = degree < 30 ? 30 : 40;  // no new residual: keep following the schedule
            else {
                if (prev_resid > 0.0 && last_resid > 0.5 * prev_resid) degree = std::max(4, degree / 2);
                else degree = degree < 30 ? 30 : 40;
                prev_resid = last_resid;
            }
        }
        outer_done = outer;
        if (kt) MMW_TRY(kt->end());
        if (getenv("MMW_FACTOR_VERBOSE")) fprintf(stderr, "[factor]   iteration done at %.1f ms\n", (vnow() - v_t0) * 1e3);
        if (getenv("MMW_FACTOR_VERBOSE"))
            fprintf(stderr, "[factor] K=%d rank=%d b=%d outer=%d degree_last=%d resid=%.2e jacobi_sweeps=%d jacobi_calls=%d\n", K, rank, b, outer, degree,
                    last_resid, dw.sweeps_total, dw.calls_total);
        if (!done && last_resid > 100 * tol)
            return fail(MMW_ERR_STATE, "mmw_factor: subspace iteration did not converge (relative residual " + std::to_string(last_resid) + ")");
        // ---- X_half = V[:, top rank] sqrt|theta|, columns in ascending |theta| (svds order, mmw.py:215-216)
        std::vector<int> sel(rank);
        std::vector<double> sc(rank);
        for (int c = 0; c < rank; ++c) {
            sel[c] = rank - 1 - c;
            sc[c] = std::sqrt(std::fabs(theta[rank - 1 - c]));
        }
        if (out64.n < (size_t)K * rank) MMW_TRY(out64.alloc((size_t)K * rank));
        MMW_HIP(hipMemcpyAsync(dw.perm.p, sel.data(), rank * sizeof(int), hipMemcpyHostToDevice, st));
        MMW_HIP(hipMemcpyAsync(dw.diag.p, sc.data(), rank * sizeof(double), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL((k_export_factor<T>), dim3(grid_elems((size_t)K * rank)), dim3(BLOCK), 0, st, K, rank, ld, V.p, dw.perm.p, dw.diag.p, out64.p);
        MMW_HIP(hipGetLastError());
        last_n = 0;
        last_on_host = false;
        if (out) MMW_TRY(fetch_last((size_t)K * rank));  // (out == nullptr: the factor stays on the device until somebody reads it)
        else MMW_HIP(hipStreamSynchronize(st));
        last_n = (size_t)K * rank;
        last_rank = rank;
        if (getenv("MMW_FACTOR_VERBOSE")) fprintf(stderr, "[factor]   export + copy-out done at %.1f ms\n", (vnow() - v_t0) * 1e3);

        // restore the unit column scales used by k_select_cols
        if (out) memcpy(out, last_host.p, last_n * sizeof(double));
        if (getenv("MMW_FACTOR_VERBOSE")) fprintf(stderr, "[factor]   host copy done at %.1f ms\n", (vnow() - v_t0) * 1e3);
        return MMW_OK;
    }
};

}  // namespace mmw
