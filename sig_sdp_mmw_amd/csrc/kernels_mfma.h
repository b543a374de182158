// The CSR SpMM of exp(L/2)R on the matrix cores (fp32 handles, locality-blocked patterns).
//
// A row block of the blocking (blocking.h) is <= 32 matrix rows whose nonzeros all fall into <= 416 "union" columns, and the
// blocks of a geometric interference graph are 40-50 % dense in that union.  The block product
//     Out[rows, :] = A[rows, union] * U[union, :]
// is therefore run as a dense 32 x (16 ksteps) x D product on v_mfma_f32_32x32x16_bf16, with both operands split into two
// bf16 halves (x = hi + lo, 16 significant bits) and the three leading partial products accumulated in fp32:
//     A U ~= Ahi Uhi + Ahi Ulo + Alo Uhi          relative error <= ~2^-16 of sum |a||u|.
// The product enters exp(A)b multiplied by the step's norm (rho ~ 3e-3 at the benchmark), so the result keeps the fp32 path's
// accuracy as long as 2^-15 * max_i sum_j |a_ij| stays below the tolerance; k_plan checks exactly that (ExpmPlan::mfma_ok) and the
// host falls back to the fp32 LDS kernel (k_spmm_blk2) otherwise.  What this buys: the fp32 kernel spends its time issuing
// one v_pk_fma + one ds_read per nonzero and 16 bytes (VALU 55 % busy, LDS 45 %); here a k-step of 16 union rows costs a wave
// 4 transposed LDS reads and 3 MFMAs per 32 output columns whatever the fill, and what remains is the gather of the union's
// rows into LDS (the CU's L2 rate).
//
// Operands.
//   A: k_loss writes every stored entry of L as one 32-bit word (bf16 hi << 16 | bf16 lo) into a dense image of the block in
//      MFMA fragment order, [block k-step][lane][8]: lane l = (row r = l & 31, half h = l >> 5) holds A[r][16 s + 8 h + j],
//      j = 0..7 -- the A operand map of v_mfma_f32_32x32x16_bf16 -- so a wave fetches a k-step's fragment with two fully
//      coalesced 16-byte loads per lane and separates the halves with 8 v_perm_b32.  Holes stay zero (the pattern is fixed).
//   U: the producer of a Krylov block also writes its bf16 hi / lo planes ([K][Dpad] each, the same bytes as the fp32 block).
//      A k-step's 16 union rows of both planes are staged in LDS row-major, exactly as gathered, and read back with
//      ds_read_b64_tr_b16: the hardware transpose delivers, per lane, 4 consecutive k of one column -- the B operand map.
//      A row's 64-byte groups are rotated by (row & 3) where the pitch would otherwise put the four rows of a transposed read on
//      the same banks.
// One workgroup = (row block, group of 4 * NT column tiles of 32); wave w owns NT column tiles (48 accumulator registers at
// NT = 3); two LDS buffers, one barrier per k-step; the fused Lanczos epilogue (o = a A u - shift u, u.o and o.o column slabs)
// works straight on the accumulator layout: a lane holds 16 rows of one column, the other 16 sit in lane + 32.
#pragma once
#include "kernels_expm.h"

namespace mmw {

typedef short mf_s4 __attribute__((ext_vector_type(4)));
typedef short mf_s8 __attribute__((ext_vector_type(8)));
typedef __bf16 mf_bf8 __attribute__((ext_vector_type(8)));
typedef float mf_f16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short bf16_rn(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// x ~= hi + lo with hi = rn_bf16(x), lo = rn_bf16(x - hi); returned as hi << 16 | lo
__device__ __forceinline__ unsigned split_bf16(float x) {
    const unsigned short hi = bf16_rn(x);
    const unsigned short lo = bf16_rn(x - bf16_f32(hi));
    return ((unsigned)hi << 16) | (unsigned)lo;
}

// hi / lo planes of an fp32 block (n = K * Dpad elements, a multiple of 4)
__global__ __launch_bounds__(BLOCK) void k_split_planes(size_t n4, const float4* __restrict__ src, uint2* __restrict__ hi, uint2* __restrict__ lo) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n4; o += (size_t)gridDim.x * BLOCK) {
        const float4 x = src[o];
        const unsigned a = split_bf16(x.x), b = split_bf16(x.y), c = split_bf16(x.z), d = split_bf16(x.w);
        hi[o] = make_uint2((a >> 16) | (b & 0xFFFF0000u), (c >> 16) | (d & 0xFFFF0000u));
        lo[o] = make_uint2((a & 0xFFFFu) | (b << 16), (c & 0xFFFFu) | (d << 16));
    }
}

constexpr int MF_THREADS = 256;
constexpr int MF_WAVES = MF_THREADS / WAVE;
constexpr int MF_KROWS = 16;  // union rows per k-step
struct MfmaDev {
    const int* kbase;        // [nb+1] k-steps before each block
    const unsigned* afrag;   // fragment-ordered image of the matrix, hi << 16 | lo
};
template <int NT> constexpr int mf_lds_bytes() { return BLK_UNION_ROWS * 4 + 128 + 2 * (2 * MF_KROWS * MF_WAVES * NT * 64); }

template <int MODE, int NT>
__global__ __launch_bounds__(MF_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_spmm_mfma(BlkDev B, MfmaDev M, int Dpad, size_t plane_bytes, const char* __restrict__ Upl, const float* __restrict__ U,
                 float* __restrict__ Out, double ascale_d, double shift_d, double* __restrict__ partial, double* __restrict__ partial_o2,
                 const ExpmPlan* __restrict__ plan, int step, int* __restrict__ viol) {
    static_assert(MODE == SPMM_PLAIN || MODE == SPMM_LANCZOS, "the matrix-core SpMM has the plain and the Lanczos epilogue");
    bool shifted = false;
    if (plan) {
        if (MODE == SPMM_LANCZOS) {
            if (step > plan_steps(plan, step - 1)) return;
            shifted = plan->apost != 0 && partial_o2 != nullptr;
            if (shifted) shift_d = plan->mu;
        } else if (step > plan->m) return;
        // launched without a plan readback on a matrix whose norm has outgrown the two-half split: the chunk is replayed on the fp32 kernel
        if (!plan->mfma_ok && viol && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *viol = 1;
    }
    const float ascale = (float)ascale_d, shift = (float)shift_d;
    constexpr int GT = MF_WAVES * NT;  // column tiles (32 columns, 64 bytes per plane row) per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int* un_l = reinterpret_cast<int*>(smem_raw);                   // [BLK_UNION_ROWS] union column ids
    int* orow_l = reinterpret_cast<int*>(smem_raw + BLK_UNION_ROWS * 4);  // [32] output rows
    char* bufs = smem_raw + BLK_UNION_ROWS * 4 + 128;
    constexpr int BUF_BYTES = 2 * MF_KROWS * GT * 64;

    const int per = (B.nb + 7) >> 3;
    const int rb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);  // consecutive row blocks share an XCD's L2
    if (rb >= B.nb) return;
    const int col0 = blockIdx.y * (GT * 32);
    const int ng = min(GT, (Dpad - col0) >> 5);  // column tiles of this group (Dpad is a multiple of 32)
    const int spr = ng * 4;                      // 16-byte slots per plane row
    const int* dsc = B.desc + (size_t)rb * 8;
    const int q0 = dsc[0], nrows = dsc[1], nun = dsc[5];
    const int KS = (nun + MF_KROWS - 1) / MF_KROWS;
    const int kb = M.kbase[rb];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    for (int i = threadIdx.x; i < KS * MF_KROWS; i += MF_THREADS) un_l[i] = B.un_fixed[(size_t)rb * BLK_UNION_ROWS + i];
    if ((int)threadIdx.x < 32) orow_l[threadIdx.x] = (int)threadIdx.x < nrows ? B.order[q0 + threadIdx.x] : -1;

    // bank rotation of the 64-byte groups of a staged row (see the header): 4 rows of a transposed read on 4 bank quarters
    const int rmode = (ng & 3) == 0 ? 2 : ((ng & 3) == 2 ? 1 : 0);
    auto rot = [&](int q) { return rmode == 2 ? q : (rmode == 1 ? (q >> 1) : 0); };

    // staging slots of this thread: slot = tid + 256 j -> (plane, row, 16-byte slot of the row), fixed for all k-steps
    constexpr int NS = 2 * NT;
    unsigned soff[NS];  // byte offset of the source piece relative to (plane 0, row 0); ~0u: idle
    int srow[NS];
    const unsigned pitch = (unsigned)Dpad * 2u;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int slot = threadIdx.x + MF_THREADS * j;
        soff[j] = ~0u;
        srow[j] = 0;
        if (slot < 2 * MF_KROWS * spr) {
            const int p = slot / (MF_KROWS * spr), rem = slot - p * (MF_KROWS * spr);
            const int r = rem / spr, t = rem - r * spr;
            int G = (t >> 2) - rot(r & 3);
            if (G < 0) G += ng;
            soff[j] = (unsigned)p * (unsigned)plane_bytes + (unsigned)col0 * 2u + (unsigned)(G * 4 + (t & 3)) * 16u;
            srow[j] = r;
        }
    }
    uint4 st[NS];
    auto issue = [&](int s) {
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (soff[j] != ~0u) st[j] = *reinterpret_cast<const uint4*>(Upl + ((size_t)(unsigned)un_l[s * MF_KROWS + srow[j]] * pitch + soff[j]));
    };
    auto deposit = [&](int b) {
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (soff[j] != ~0u) *reinterpret_cast<uint4*>(bufs + b * BUF_BYTES + (threadIdx.x + MF_THREADS * j) * 16) = st[j];
    };

    // transposed-read addresses of this lane: 16-lane group g (h = g >> 1: k half, g & 1: column half), lane 4q + p of the group
    const int g16 = lane >> 4, l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3;
    unsigned rbase[NT];
    bool tile_on[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int G = wv * NT + i;
        tile_on[i] = G < ng;
        int Gp = G + rot(tq);
        if (Gp >= ng) Gp -= ng;
        rbase[i] = (unsigned)((8 * (g16 >> 1) + tq) * spr * 16 + Gp * 64 + (g16 & 1) * 32 + tp * 8);
    }
    const unsigned row4 = (unsigned)(4 * spr * 16), plane_l = (unsigned)(MF_KROWS * spr * 16);

    mf_f16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;

    __syncthreads();  // un_l
    issue(0);
    const uint4* ap = reinterpret_cast<const uint4*>(M.afrag) + ((size_t)kb * 64 + lane) * 2;
    uint4 a0 = ap[0], a1 = ap[1];
    deposit(0);
    if (KS > 1) issue(1);
    __syncthreads();
    for (int s = 0; s < KS; ++s) {
        const char* bb = bufs + (s & 1) * BUF_BYTES;
        // the two halves of the A fragment: word = hi << 16 | lo per k
        mf_s8 ahi, alo;
        {
            const unsigned w[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            unsigned h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x07060302u);
                l[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x05040100u);
            }
            const uint4 hv = make_uint4(h[0], h[1], h[2], h[3]), lv = make_uint4(l[0], l[1], l[2], l[3]);
            ahi = __builtin_bit_cast(mf_s8, hv);
            alo = __builtin_bit_cast(mf_s8, lv);
        }
        if (s + 1 < KS) {  // next fragment flies during the products
            a0 = ap[(size_t)(s + 1) * 128];
            a1 = ap[(size_t)(s + 1) * 128 + 1];
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (tile_on[i]) {  // wave-uniform
                typedef __attribute__((address_space(3))) mf_s4* lp;
                const char* p0 = bb + rbase[i];
                const mf_s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
                const mf_s4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0 + row4));
                const mf_s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0 + plane_l));
                const mf_s4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0 + plane_l + row4));
                const mf_s8 bhi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                const mf_s8 blo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, alo), __builtin_bit_cast(mf_bf8, bhi), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, ahi), __builtin_bit_cast(mf_bf8, blo), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, ahi), __builtin_bit_cast(mf_bf8, bhi), acc[i], 0, 0, 0);
            }
        }
        if (s + 1 < KS) {
            deposit((s + 1) & 1);  // everyone left that buffer at the barrier that ended step s - 1
            if (s + 2 < KS) issue(s + 2);
        }
        __syncthreads();
    }

    // ---- epilogue on the accumulator layout: lane = column (lane & 31), register v = row (v & 3) + 8 (v >> 2) + 4 (lane >> 5)
    const int h2 = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (!tile_on[i]) continue;
        const int col = col0 + (wv * NT + i) * 32 + cl;
        float dot = 0.f, dot2 = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int rl = (v & 3) + 8 * (v >> 2) + 4 * h2;
            const int row = orow_l[rl];
            if (row >= 0) {
                const size_t off = (size_t)row * Dpad + col;
                float o;
                if (MODE == SPMM_LANCZOS) {
                    const float u = U[off];
                    o = ascale * acc[i][v] - shift * u;  // shift is 0 unless the recurrence runs on A - mu I
                    dot += u * o;
                    dot2 += o * o;
                } else {
                    o = ascale * acc[i][v];
                }
                Out[off] = o;
            }
        }
        if (MODE == SPMM_LANCZOS) {
            float a, b;
            rows32(dot, a, b);
            const float d1 = a + b;
            rows32(dot2, a, b);
            const float d2 = a + b;
            if (h2 == 0) {
                partial[(size_t)rb * Dpad + col] = (double)d1;
                if (shifted) partial_o2[(size_t)rb * Dpad + col] = (double)d2;
            }
        }
    }
}

}  // namespace mmw
