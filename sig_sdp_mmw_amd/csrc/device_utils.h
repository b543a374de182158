// Small device-side helpers shared by all kernels (gfx950: 64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mmw {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;               // 4 waves per workgroup
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int MAX_ORDER = 16;            // Krylov order cap per substep
constexpr int MAX_PART = 1024;           // upper bound on per-block partial slabs

template <typename T> struct V16;
template <> struct V16<float> {
    using type = float4;
    static constexpr int N = 4;
};
template <> struct V16<double> {
    using type = double2;
    static constexpr int N = 2;
};

template <typename T, int N> struct Pack {
    T v[N];
};

__device__ __forceinline__ void load16(const float* p, float (&x)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
}
__device__ __forceinline__ void load16(const double* p, double (&x)[2]) {
    const double2 t = *reinterpret_cast<const double2*>(p);
    x[0] = t.x; x[1] = t.y;
}
__device__ __forceinline__ void store16(float* p, const float (&x)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ void store16(double* p, const double (&x)[2]) {
    *reinterpret_cast<double2*>(p) = make_double2(x[0], x[1]);
}

// butterfly all-reduce over `width` consecutive lanes (width a power of two <= 64)
template <typename R> __device__ __forceinline__ R group_sum(R v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
template <typename R> __device__ __forceinline__ R wave_sum(R v) { return group_sum(v, WAVE); }
template <typename R> __device__ __forceinline__ R wave_max(R v) {
    for (int o = WAVE >> 1; o > 0; o >>= 1) {
        const R w = __shfl_xor(v, o, WAVE);
        v = w > v ? w : v;
    }
    return v;
}

// workgroup reductions through LDS; every thread gets the result. `sh` needs WAVES_PER_BLOCK entries.
template <typename R> __device__ __forceinline__ R block_sum(R v, R* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    R t = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}
template <typename R> __device__ __forceinline__ R block_max(R v, R* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    R t = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = sh[i] > t ? sh[i] : t;
    return t;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al., SC'11) for the on-device Gaussian sketch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// two standard normals from 4 random words (Box-Muller on a 53-bit / 32-bit uniform pair, in double)
__device__ __forceinline__ void box_muller(const uint32_t (&w)[4], double& n0, double& n1) {
    const double u1 = ((double)(((uint64_t)w[0] << 21) ^ (uint64_t)(w[1] >> 11)) + 1.0) * (1.0 / 9007199254740993.0);
    const double u2 = ((double)w[2] + (double)w[3] * (1.0 / 4294967296.0)) * (1.0 / 4294967296.0);
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    n0 = r * c;
    n1 = r * s;
}

}  // namespace mmw
