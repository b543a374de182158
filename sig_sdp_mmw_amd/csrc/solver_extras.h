// Rounding, epilogue factor and duality-gap routines that hang off a solver handle.
#pragma once
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
    DevBuf<double> so_data, h_max;
    DevBuf<double> gX, randv, P, gain, nrm;
    DevBuf<int> pref, slot, order, rem;

    int init(hipStream_t s, const HostPattern* h, int K_, KernelTimers* k) {
        st = s; H = h; K = K_; kt = k;
        MMW_TRY(so_indptr.upload(H->so_indptr, st));
        MMW_TRY(so_indices.upload(H->so_indices, st));
        MMW_TRY(so_data.upload(H->so_data, st));
        MMW_TRY(q_indptr.upload(H->q_indptr, st));
        MMW_TRY(q_indices.upload(H->q_indices, st));
        MMW_TRY(h_max.upload(H->h_max, st));
        MMW_HIP(hipStreamSynchronize(st));
        return MMW_OK;
    }
    template <typename B> static int ensure(DevBuf<B>& b, size_t n) {
        if (b.n >= n && b.p) return MMW_OK;
        return b.alloc(n);
    }

    int gap(double*) { return fail(MMW_ERR_STATE, "mmw_gap: not built yet"); }
    int factor(int32_t, double*, uint64_t) { return fail(MMW_ERR_STATE, "mmw_factor: not built yet"); }
    int read_factor(double*, int64_t) { return fail(MMW_ERR_STATE, "mmw_factor: not built yet"); }

    int round(int32_t Z, int32_t Dp, const double* gX_h, int32_t nb, const double* randv_h, int32_t* z_out, int32_t* rem_out) {
        if (Z < 1 || Dp < 1 || nb < 1) return fail(MMW_ERR_ARG, "mmw_round: Z, D' and nbatch must be positive");
        if (!gX_h || !randv_h || !z_out || !rem_out) return fail(MMW_ERR_ARG, "mmw_round: null pointer");
        if ((size_t)Z * sizeof(int) > 60000) return fail(MMW_ERR_ARG, "mmw_round: Z too large");
        const size_t nP = (size_t)nb * K * Z;
        MMW_TRY(ensure(gX, (size_t)K * Dp));
        MMW_TRY(ensure(randv, (size_t)nb * Z * Dp));
        MMW_TRY(ensure(P, nP));
        MMW_TRY(ensure(pref, nP));
        MMW_TRY(ensure(gain, nP));
        MMW_TRY(ensure(slot, (size_t)nb * K));
        MMW_TRY(ensure(nrm, K));
        MMW_TRY(ensure(order, K));
        MMW_TRY(ensure(rem, nb));
        MMW_HIP(hipMemcpyAsync(gX.p, gX_h, (size_t)K * Dp * sizeof(double), hipMemcpyHostToDevice, st));
        MMW_HIP(hipMemcpyAsync(randv.p, randv_h, (size_t)nb * Z * Dp * sizeof(double), hipMemcpyHostToDevice, st));
        MMW_HIP(hipMemsetAsync(gain.p, 0, nP * sizeof(double), st));
        MMW_HIP(hipMemsetAsync(slot.p, 0xFF, (size_t)nb * K * sizeof(int), st));
        if (kt) MMW_TRY(kt->begin(KT_PROJECT));
        hipLaunchKernelGGL(k_row_norms_f64, dim3(grid_rows(K)), dim3(BLOCK), 0, st, K, Dp, gX.p, nrm.p);
        hipLaunchKernelGGL(k_rank_desc, dim3((K + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, K, nrm.p, order.p);
        hipLaunchKernelGGL(k_project_mfma, dim3((K + 63) / 64, (Z + 15) / 16, nb), dim3(BLOCK), 0, st, K, Z, Dp, gX.p, randv.p, P.p);
        hipLaunchKernelGGL(k_slot_pref, dim3(K, nb), dim3(BLOCK), (size_t)Z * sizeof(double), st, K, Z, P.p, pref.p);
        if (kt) MMW_TRY(kt->end());
        MMW_HIP(hipGetLastError());
        if (kt) MMW_TRY(kt->begin(KT_GREEDY));
        hipLaunchKernelGGL(k_greedy, dim3(nb), dim3(BLOCK), (size_t)Z * sizeof(int), st, K, Z, order.p, pref.p, so_indptr.p,
                           so_indices.p, so_data.p, q_indptr.p, q_indices.p, h_max.p, gain.p, slot.p, rem.p);
        if (kt) MMW_TRY(kt->end());
        MMW_HIP(hipGetLastError());
        MMW_HIP(hipMemcpyAsync(z_out, slot.p, (size_t)nb * K * sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipMemcpyAsync(rem_out, rem.p, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        return MMW_OK;
    }
};

}  // namespace mmw
