// Small device-side helpers shared by all kernels (gfx950: 64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mmw {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;               // 4 waves per workgroup
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int MAX_ORDER = 16;            // Krylov order cap per substep
constexpr int MAX_PART = 1024;           // upper bound on per-block partial slabs (Dpad-wide rows of doubles)
constexpr int ROW_GRID_MAX = 4096;       // upper bound on the grid of one-wave-per-row kernels (scalar partials per block)

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

// ---- cross-lane exchanges without LDS traffic ---------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32, which occupies the LDS pipe the blocked kernels are bound by.
// Within a 16-lane row DPP modifiers do the exchange in the VALU; across rows gfx950 has v_permlane16_swap /
// v_permlane32_swap, which with both operands equal leave (even rows' value, odd rows' value) in the two results.
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) { return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v))); }
template <int CTRL> __device__ __forceinline__ int dpp(int v) { return (int)dpp_u32<CTRL>((unsigned)v); }
template <int CTRL> __device__ __forceinline__ double dpp(double v) {
    const unsigned long long w = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = dpp_u32<CTRL>((unsigned)w), hi = dpp_u32<CTRL>((unsigned)(w >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
// (a, b) = (value held by the even 16-lane rows, by the odd rows) of each row pair, in every lane of the pair
__device__ __forceinline__ void rows16(float v, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void rows32(float v, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void rows16(int v, int& a, int& b) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    a = (int)r[0]; b = (int)r[1];
}
__device__ __forceinline__ void rows32(int v, int& a, int& b) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    a = (int)r[0]; b = (int)r[1];
}
__device__ __forceinline__ void rows16(double v, double& a, double& b) {
    const unsigned long long w = (unsigned long long)__double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)w, (unsigned)w, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(w >> 32), (unsigned)(w >> 32), false, false);
    a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
    b = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
__device__ __forceinline__ void rows32(double v, double& a, double& b) {
    const unsigned long long w = (unsigned long long)__double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)w, (unsigned)w, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(w >> 32), (unsigned)(w >> 32), false, false);
    a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
    b = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
// value held by lane (l ^ o), o a wave-uniform power of two
template <typename R> __device__ __forceinline__ R lane_xor(R v, int o) {
    const int lane = threadIdx.x & 63;
    switch (o) {
        case 32: { R a, b; rows32(v, a, b); return lane & 32 ? a : b; }
        case 16: { R a, b; rows16(v, a, b); return lane & 16 ? a : b; }
        case 8: return dpp<DPP_ROW_ROR8>(v);
        case 4: { const R up = dpp<0x104>(v), dn = dpp<0x114>(v); return lane & 4 ? dn : up; }  // row_shl:4 / row_shr:4
        case 2: return dpp<DPP_QUAD_XOR2>(v);
        default: return dpp<DPP_QUAD_XOR1>(v);
    }
}
// sum over the lanes l' == l (mod stride), stride a wave-uniform power of two: every lane gets its class total
template <typename R> __device__ __forceinline__ R stride_sum(R v, int stride) {
    if (stride <= 32) { R a, b; rows32(v, a, b); v = a + b; }
    if (stride <= 16) { R a, b; rows16(v, a, b); v = a + b; }
    if (stride <= 8) v += dpp<DPP_ROW_ROR8>(v);   // from here the values repeat every 8 / 4 / 2 lanes, so a
    if (stride <= 4) v += dpp<DPP_ROW_ROR4>(v);   // rotation fetches the same value the xor partner holds
    if (stride <= 2) v += dpp<DPP_ROW_ROR2>(v);
    if (stride <= 1) v += dpp<DPP_ROW_ROR1>(v);
    return v;
}
struct OpSum { template <typename R> __device__ __forceinline__ R operator()(R a, R b) const { return a + b; } };
struct OpMax { template <typename R> __device__ __forceinline__ R operator()(R a, R b) const { return b > a ? b : a; } };
// all-reduce over aligned groups of `width` consecutive lanes (a power of two <= 64); every lane gets the result,
// and every lane of a group combines the operands in the same order
template <typename R, typename Op> __device__ __forceinline__ R group_reduce(R v, int width, Op op) {
    if (width >= 64) { R a, b; rows32(v, a, b); v = op(a, b); }
    if (width >= 32) { R a, b; rows16(v, a, b); v = op(a, b); }
    if (width >= 16) {
        v = op(v, dpp<DPP_ROW_ROR8>(v));
        v = op(v, dpp<DPP_ROW_ROR4>(v));
        v = op(v, dpp<DPP_ROW_ROR2>(v));
        v = op(v, dpp<DPP_ROW_ROR1>(v));
        return v;
    }
    if (width >= 8) v = op(v, dpp<DPP_ROW_HALF_MIRROR>(v));
    if (width >= 4) v = op(v, dpp<DPP_QUAD_XOR2>(v));
    if (width >= 2) v = op(v, dpp<DPP_QUAD_XOR1>(v));
    return v;
}
template <typename R> __device__ __forceinline__ R group_sum(R v, int width) { return group_reduce(v, width, OpSum()); }
template <typename R> __device__ __forceinline__ R wave_sum(R v) { return group_reduce(v, WAVE, OpSum()); }
template <typename R> __device__ __forceinline__ R wave_max(R v) { return group_reduce(v, WAVE, OpMax()); }

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
