// The CSR SpMM of exp(L/2)R on the matrix cores (fp32 handles, locality-blocked patterns).
//
// A row block of the blocking (blocking.h) is <= 32 * MT matrix rows whose nonzeros all fall into <= 416 "union" columns, and
// the blocks of a geometric interference graph are 40-50 % dense in that union.  The block product
//     Out[rows, :] = A[rows, union] * U[union, :]
// is therefore run as a dense (32 MT) x (16 ksteps) x D product on v_mfma_f32_32x32x16_bf16, with both operands split into two
// bf16 halves (x = hi + lo, 16 significant bits) and the three leading partial products accumulated in fp32:
//     A U ~= Ahi Uhi + Ahi Ulo + Alo Uhi          ||error||_F <= 3 * 2^-17 || |A| ||_2 ||U||_F.
// The product enters exp(A)b multiplied by the step's norm (rho ~ 3e-3 at the benchmark), so the result keeps the fp32 path's
// accuracy as long as 2.3e-5 * max_i sum_j |a_ij| stays below the tolerance; k_plan checks exactly that (ExpmPlan::mfma_ok) and
// the host falls back to the fp32 LDS kernel (k_spmm_blk2) otherwise.  What this buys: the fp32 kernel spends its time issuing
// one v_pk_fma + one ds_read per nonzero and 16 bytes (VALU 55 % busy, LDS 45 %); here a k-step of 16 union rows costs a wave
// 4 transposed LDS reads and 3 MFMAs per 32 output columns whatever the fill, and what remains is the gather of the union's
// rows into LDS (the CU's L2 rate).
//
// Operands.
//   A: k_loss writes every stored entry of L as one 32-bit word (bf16 hi << 16 | bf16 lo) into a dense image of the block in
//      MFMA fragment order, [block k-step][row tile][half][lane][4]: lane l = (row r = l & 31, k half h = l >> 5) holds
//      A[r][16 s + 8 h + j], j = 0..7 -- the A operand map of v_mfma_f32_32x32x16_bf16 -- words j = 0..3 in the first 1-KiB half,
//      j = 4..7 in the second.  Holes stay zero (the pattern is fixed).
//   U: the producer of a Krylov block also writes its bf16 hi / lo planes ([K][Dpad] each, the same bytes as the fp32 block).
// Pipeline.  One workgroup = (row block, group of NW * NT column tiles of 32 columns); wave w owns NT column tiles x MT row
// tiles.  A k-step's chunk -- the 16 union rows of both planes for the group's columns, row-major exactly as gathered, plus the
// A fragments of the step -- goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: per-lane source address, lane-linear 1-KiB
// destination, no registers), three chunks deep: while step s is multiplied, the pieces of steps s+1 and s+2 are in flight
// (48-56 KiB per workgroup, two or three workgroups per CU).  Per step: a counted s_waitcnt vmcnt for the wave's own pieces of
// chunk s, ONE s_barrier, the DMA issue for chunk s+2 into the buffer everyone left at that barrier, then
// ds_read_b64_tr_b16 x 4 (the hardware transpose delivers, per lane, 4 consecutive k of one column: the B operand map)
// and 3 MFMAs per (row tile, column tile).  A staged row's 64-byte groups are rotated by (row & 3) where the pitch would
// otherwise put the four rows of a transposed read on the same banks.
// The fused Lanczos epilogue (o = a A u - shift u, u.o and o.o column slabs) works straight on the accumulator layout:
// a lane holds 16 rows of one column, the other 16 sit in lane + 32.
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

constexpr double SDM_FX = 1099511627776.0;  // 2^40: fixed-point scale of the row totals added with integer atomics (row sums of X, row norms of y)
constexpr int MF_KROWS = 16;  // union rows per k-step
constexpr int MF_UNION_ROWS = 640;  // == MF_UNION (blocking.h)
constexpr int MF_KPAD = 4;    // a block's k-steps are padded to a multiple of this in the fragment image (zero fragments)
struct MfmaDev {             // the matrix-core kernel's own row blocks (blocking.h, build_mfma_blocking)
    int nb;
    const int* desc;         // [nb][8] {first position, rows, 0, 0, 0, union size, 0, 0}
    const int* un_fixed;     // [nb][MF_UNION_ROWS] union column ids, padded with the first
    const int* order;        // position -> matrix row
    const int* kbase;        // [nb+1] (padded) k-steps before each block
    const unsigned* afrag;   // fragment-ordered image of the matrix, hi << 16 | lo
};
// optional inputs / outputs of the epilogue: the AXPBY form (Out = ascale A U + shift F + c3 X2, the Chebyshev recurrences of
// factor.h) reads F and X2 at the output rows; out_planes, when set, also receives the result as bf16 hi / lo planes (the next
// product's B operand), hi plane first, the lo plane plane_bytes further
struct MfEpi {
    const float* F = nullptr;
    const float* X2 = nullptr;
    float c3 = 0.f;
    unsigned short* out_planes = nullptr;
    // SPMM_FIRST (the whole exponential in this launch, y = u + (ascale A - mu I) u; see first_order_bound): y leaves as the two bf16
    // halves interleaved per 32 columns that k_sddmm_mfma gathers (y_planes) and, when Out is given, as fp32; ||y_row||^2 is
    // added to dfx[row] as a 2^-40 fixed-point integer (zero at launch; the column groups of a row arrive in any order) and
    // every workgroup leaves its share of the trace in tr_part[blockIdx.y * gridDim.x + blockIdx.x] (zero where no workgroup works)
    unsigned short* y_planes = nullptr;
    long long* dfx = nullptr;
    double* tr_part = nullptr;
};
// a chunk = KC k-steps: per k-step the B image (2 planes x 16 rows x the group's columns), then the A fragments of all KC steps
// GT = column tiles per workgroup
// PL = planes of the B operand: 2 (bf16 hi / lo) or 1 (one fp16 plane: the first-order product, below)
// AP = 1-KiB pieces of the A operand per k-step and row tile: 2 (two 16-bit halves per entry) or 1 (one fp16 half: SPMM_FIRST16)
template <int MT, int GT, int KC, int PL = 2, int AP = 2> constexpr int mf_chunk_bytes() { return KC * (1024 * PL * GT + 1024 * AP * MT); }
// NB = chunks resident in LDS (one being multiplied, NB - 1 in flight)
template <int MT, int GT, int KC, int NB, int PL = 2, int AP = 2> constexpr int mf_lds_bytes() { return MF_UNION_ROWS * 4 + 64 * 4 + NB * mf_chunk_bytes<MT, GT, KC, PL, AP>(); }
constexpr bool mf_first(int mode) { return mode == SPMM_FIRST || mode == SPMM_FIRST16; }
constexpr int mf_planes(int mode) { return mf_first(mode) ? 1 : 2; }
constexpr int mf_apieces(int mode) { return mode == SPMM_FIRST16 ? 1 : 2; }
typedef _Float16 mf_h8 __attribute__((ext_vector_type(8)));
// The first-order product runs on fp16 operands: its result o = A'u enters y = u + o at q = ||A'u|| / ||u|| ~ 1e-4, so ONE fp16 plane of u
// (11 significant bits: ||d o|| <= 2^-12 || |A| ||_2 ||u||, ExpmPlan::f16_ok) replaces the two bf16 halves -- half the gather, two products
// per tile and k-step instead of three.  The matrix keeps two halves (fp16 hi + fp16 lo of the entry times 2^20: entries of L are ~1e-5,
// below fp16's normal range; the scale is taken out again with `ascale`).
constexpr float MF_F16_SCALE = 1048576.0f;
__device__ __forceinline__ unsigned short f16_rn(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float f16_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
// x * 2^20 ~= hi + lo in fp16, returned as hi << 16 | lo (the matrix image's word)
__device__ __forceinline__ unsigned split_f16_scaled(float x) {
    const float xs = x * MF_F16_SCALE;
    const unsigned short hi = f16_rn(xs);
    const unsigned short lo = f16_rn(xs - f16_f32(hi));
    return ((unsigned)hi << 16) | (unsigned)lo;
}
// position of an entry in the one-half image (SPMM_FIRST16) from its position `w` in the two-halves image (blocking.h, fpos): the same
// (k-step, row tile, lane), the lane's 8 entries contiguous
__device__ __forceinline__ size_t mf_pos16(int w) {
    const int j = (w & 3) + 4 * ((w >> 8) & 1), lane = (w >> 2) & 63;
    return ((size_t)(w >> 9) * 64 + lane) * 8 + j;
}
// one fp16 plane of an fp32 block (the first-order product's B operand, when the block's producer did not write it)
__global__ __launch_bounds__(BLOCK) void k_plane_f16(size_t n4, const float4* __restrict__ src, uint2* __restrict__ dst) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n4; o += (size_t)gridDim.x * BLOCK) {
        const float4 x = src[o];
        dst[o] = make_uint2((unsigned)f16_rn(x.x) | ((unsigned)f16_rn(x.y) << 16), (unsigned)f16_rn(x.z) | ((unsigned)f16_rn(x.w) << 16));
    }
}


// s_waitcnt vmcnt(n) for a wave-uniform n known only at run time (the instruction takes an immediate)
__device__ __forceinline__ void mf_wait_vmcnt(int n) {
    switch (n) {
#define MMW_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        MMW_W(0) MMW_W(1) MMW_W(2) MMW_W(3) MMW_W(4) MMW_W(5) MMW_W(6) MMW_W(7) MMW_W(8) MMW_W(9) MMW_W(10) MMW_W(11) MMW_W(12)
        MMW_W(13) MMW_W(14) MMW_W(15) MMW_W(16) MMW_W(17) MMW_W(18) MMW_W(19) MMW_W(20) MMW_W(21) MMW_W(22) MMW_W(23) MMW_W(24)
        MMW_W(25) MMW_W(26) MMW_W(27) MMW_W(28) MMW_W(29) MMW_W(30) MMW_W(31) MMW_W(32) MMW_W(33) MMW_W(34) MMW_W(35) MMW_W(36)
        MMW_W(37) MMW_W(38) MMW_W(39) MMW_W(40) MMW_W(41) MMW_W(42) MMW_W(43) MMW_W(44) MMW_W(45) MMW_W(46) MMW_W(47) MMW_W(48)
#undef MMW_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// n / d for the small non-negative integers of the piece decode (n < 4096, d <= 1024): exact, no integer division
__device__ __forceinline__ int mf_div(int n, int d, float rd) { (void)d; return (int)(((float)n + 0.5f) * rd); }
// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to the 1 KiB of LDS at byte address `lds` (wave-uniform).
// Written as inline assembly on purpose: the compiler orders every later LDS read behind a global_load_lds it knows about with
// s_waitcnt vmcnt(0) (it cannot tell the buffers apart), which drains the chunks that are supposed to stay in flight across the
// products of this step.  The waits for these pieces are the counted ones in the main loop.
__device__ __forceinline__ void mf_dma16(const char* src, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds) : "memory", "m0");
}

// MT row tiles per block, NT column tiles per wave, NW waves of which MS groups split the row tiles among them
// (NW / MS waves side by side along the columns), KC k-steps per chunk, NB chunks resident
template <int MODE, int MT, int NT, int NW, int MS, int KC, int NB>
__global__ __launch_bounds__(NW * 64)
void k_spmm_mfma(MfmaDev M, int Dpad, size_t plane_bytes, const char* __restrict__ Upl, const float* __restrict__ U,
                 float* __restrict__ Out, double ascale_d, double shift_d, double* __restrict__ partial, double* __restrict__ partial_o2,
                 const ExpmPlan* __restrict__ plan, int step, int* __restrict__ viol, unsigned long long* __restrict__ stamps, MfEpi E) {
    static_assert(MODE == SPMM_PLAIN || MODE == SPMM_LANCZOS || MODE == SPMM_AXPBY || mf_first(MODE), "the matrix-core SpMM has the plain, the Lanczos, the AXPBY and the first-order epilogue");
    static_assert(MF_KPAD % KC == 0, "the fragment image pads a block's k-steps to whole chunks");
    constexpr bool F16 = mf_first(MODE);      // one fp16 plane of u, fp16 hi / lo of the matrix (see MF_F16_SCALE)
    constexpr bool A16 = MODE == SPMM_FIRST16;  // ... the matrix in ONE fp16 half while 2 * 2^-12 absn <= tol (ExpmPlan::f16a_ok): an image of its own,
                                                // lane l of a (k-step, row tile) holds its 8 entries as 16 contiguous bytes -- half the A bytes, one product
    constexpr int AP = mf_apieces(MODE), AB = 1024 * AP;  // bytes of A per k-step and row tile
    // diagnostic runs only (stamps != nullptr): shader-clock sums per wave {prologue, wait + barrier, DMA issue, products, epilogue, steps}
    unsigned long long tk0 = 0, tk1 = 0, acc_wait = 0, acc_issue = 0, acc_comp = 0, t_pro = 0;
    if (stamps) tk0 = __builtin_amdgcn_s_memtime();
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    constexpr int THREADS = NW * 64;
    bool shifted = false;
    if (plan) {
        if (MODE == SPMM_LANCZOS) {
            if (step > plan_steps(plan, step - 1)) return;
            shifted = plan->apost != 0 && partial_o2 != nullptr;
            if (shifted) shift_d = plan->mu;
        } else if (F16) {
            shift_d = plan->mu;
        } else if (step > plan->m) return;
        // launched without a plan readback on a matrix whose norm has outgrown the two-half split: the chunk is replayed on the fp32 kernel
        if (!(A16 ? plan->f16a_ok : (F16 ? plan->f16_ok : plan->mfma_ok)) && viol && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicOr(viol, VIOL_OPERANDS);
    }
    const float ascale = (float)ascale_d, shift = (float)shift_d;
    static_assert(NW % MS == 0 && MT % MS == 0, "row tiles and waves split evenly");
    constexpr int NWN = NW / MS;   // waves side by side along the columns
    constexpr int MTW = MT / MS;   // row tiles per wave
    constexpr int GT = NWN * NT;   // column tiles (32 columns, 64 bytes per plane row) per workgroup
    constexpr int PL = mf_planes(MODE);
    constexpr int CHUNK = mf_chunk_bytes<MT, GT, KC, PL, AP>();
    constexpr int B1 = 1024 * PL * GT;  // B image of one k-step (at the group's full width)
    constexpr int A_OFF = KC * B1;     // the A fragments sit behind the B images of the chunk
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int* un_l = reinterpret_cast<int*>(smem_raw);                         // [MF_UNION_ROWS] union column ids
    int* orow_l = reinterpret_cast<int*>(smem_raw + MF_UNION_ROWS * 4);  // [64] output rows
    char* bufs = smem_raw + MF_UNION_ROWS * 4 + 64 * 4;
    const unsigned bufs_l = (unsigned)(size_t)(lds_vp)bufs;  // LDS byte address of the chunk buffers

    const int per = (M.nb + 7) >> 3;
    const int rb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);  // consecutive row blocks share an XCD's L2
    if (rb >= M.nb) return;
    const int col0 = blockIdx.y * (GT * 32);
    const int ng = min(GT, (Dpad - col0) >> 5);  // column tiles of this group (Dpad is a multiple of 32)
    const int spr = ng * 4;                      // 16-byte slots per plane row
    const int* dsc = M.desc + (size_t)rb * 8;
    const int q0 = dsc[0], nrows = dsc[1];
    const int kb = M.kbase[rb];
    const int KS = M.kbase[rb + 1] - kb;  // padded to a multiple of MF_KPAD: the padding fragments are zero
    const int NC = KS / KC;               // chunks
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv / NWN, wn = wv - wm * NWN;  // this wave's row-tile group and column position

    for (int i = threadIdx.x; i < KS * MF_KROWS; i += THREADS) un_l[i] = M.un_fixed[(size_t)rb * MF_UNION_ROWS + i];
    if ((int)threadIdx.x < 64) orow_l[threadIdx.x] = (int)threadIdx.x < nrows ? M.order[q0 + threadIdx.x] : -1;

    // bank rotation of the 64-byte groups of a staged row (see the header): 4 rows of a transposed read on 4 bank quarters
    const int rmode = (ng & 3) == 0 ? 2 : ((ng & 3) == 2 ? 1 : 0);
    auto rot = [&](int q) { return rmode == 2 ? q : (rmode == 1 ? (q >> 1) : 0); };

    // DMA pieces of this wave: piece i = wv + NW j of a chunk (j < cw).  Per k-step kk of the chunk, pieces [0, 2 ng) are 1-KiB
    // runs of the B image (64 slots of 16 bytes: slot -> plane, row, 16-byte piece of the row) and pieces [2 ng, 2 ng + 2 MT) the
    // A fragment halves.  Both kinds share one address form, src = base + un_l[16 KC c + row] * mul + c * step (B: mul = row
    // pitch, step = 0; A: mul = 0), so the issue loop has no branch but the piece count.
    constexpr int NJ = (KC * (PL * GT + AP * MT) + NW - 1) / NW;
    const int nB1 = PL * ng, nP1 = nB1 + AP * MT, nI = KC * nP1;
    const int cw = wv < nI ? (nI - wv + NW - 1) / NW : 0;  // pieces of this wave per chunk
    const char* pbase[NJ];
    unsigned pmul[NJ];
    int prow[NJ], pdst[NJ];
    const unsigned pitch = (unsigned)Dpad * 2u;
    const char* afr = reinterpret_cast<const char*>(M.afrag) + (size_t)kb * (AB * MT);
    const float rP1 = 1.0f / (float)nP1, rspr = 1.0f / (float)spr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = wv + NW * j;
        const int kk = mf_div(i, nP1, rP1), i1 = i - kk * nP1;
        pbase[j] = Upl;
        pmul[j] = 0;
        prow[j] = 0;
        pdst[j] = 0;
        if (i < nI && i1 < nB1) {
            const int slot = i1 * 64 + lane;
            const int p = (PL == 2 && slot >= MF_KROWS * spr) ? 1 : 0, rem = slot - p * (MF_KROWS * spr);
            const int r = mf_div(rem, spr, rspr), t = rem - r * spr;
            int G = (t >> 2) - rot(r & 3);
            if (G < 0) G += ng;
            pbase[j] = Upl + ((size_t)p * plane_bytes + (size_t)col0 * 2u + (size_t)(G * 4 + (t & 3)) * 16u);
            pmul[j] = pitch;
            prow[j] = kk * MF_KROWS + r;
            pdst[j] = kk * B1 + i1 * 1024;
        } else if (i < nI) {
            pbase[j] = afr + (kk * (AB * MT) + (i1 - nB1) * 1024 + lane * 16);
            pdst[j] = A_OFF + kk * (AB * MT) + (i1 - nB1) * 1024;
        }
        pdst[j] = __builtin_amdgcn_readfirstlane(pdst[j]);
    }
    auto issue = [&](int c) {  // chunk c -> buffer c % NB
        const unsigned dst_l = bufs_l + (unsigned)((c % NB) * CHUNK);
        const unsigned astep = (unsigned)c * (unsigned)(KC * AB * MT);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j < cw) {  // wave-uniform
                const char* src = pbase[j] + ((size_t)(unsigned)un_l[c * (KC * MF_KROWS) + prow[j]] * pmul[j] + (pmul[j] ? 0u : astep));
                mf_dma16(src, dst_l + (unsigned)pdst[j]);
            }
        }
    };

    // transposed-read addresses of this lane: 16-lane group g (h = g >> 1: k half, g & 1: column half), lane 4q + p of the group.
    // Column tiles past the group's last (a partial last group) read tile 0 again and are dropped in the epilogue: no branch
    // between a step's reads and products.
    const int g16 = lane >> 4, l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3;
    unsigned rbase[NT];
    bool tile_on[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int G = wn * NT + i;
        tile_on[i] = G < ng;
        int Gp = (tile_on[i] ? G : 0) + rot(tq);
        if (Gp >= ng) Gp -= ng;
        rbase[i] = (unsigned)((8 * (g16 >> 1) + tq) * spr * 16 + Gp * 64 + (g16 & 1) * 32 + tp * 8);
    }
    const unsigned row4 = (unsigned)(4 * spr * 16), plane_l = (unsigned)(MF_KROWS * spr * 16);

    mf_f16 acc[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[m][i][v] = 0.f;

    __syncthreads();  // un_l
#pragma unroll
    for (int c = 0; c < NB - 1; ++c)
        if (c < NC) issue(c);
    if (stamps) { tk1 = __builtin_amdgcn_s_memtime(); t_pro = tk1 - tk0; }
    for (int c = 0; c < NC; ++c) {
        // this wave's pieces of chunk c have landed when at most the pieces of the younger chunks are outstanding
        const int younger = min(NC - 1 - c, NB - 2);
        mf_wait_vmcnt(cw * younger);
        __builtin_amdgcn_s_barrier();  // everyone's pieces of chunk c are in LDS, and everyone has left the buffer of chunk c - 1
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_wait += t - tk1; tk1 = t; }
        if (c + NB - 1 < NC) issue(c + NB - 1);
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_issue += t - tk1; tk1 = t; }
        const char* cb = bufs + (c % NB) * CHUNK;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const char* bb = cb + kk * B1;
            mf_s8 ahi[MTW], alo[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {  // the two halves of the A fragment: word = hi << 16 | lo per k
                const char* af = cb + A_OFF + (kk * MT + wm * MTW + m) * AB + lane * 16;
                const uint4 a0 = *reinterpret_cast<const uint4*>(af);
                if constexpr (A16) {  // one half only: the lane's 8 entries as they stand
                    ahi[m] = __builtin_bit_cast(mf_s8, a0);
                    alo[m] = ahi[m];
                    continue;
                }
                const uint4 a1 = *reinterpret_cast<const uint4*>(af + 1024);
                const unsigned w[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                unsigned h[4], l[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x07060302u);
                    l[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x05040100u);
                }
                const uint4 hv = make_uint4(h[0], h[1], h[2], h[3]), lv = make_uint4(l[0], l[1], l[2], l[3]);
                ahi[m] = __builtin_bit_cast(mf_s8, hv);
                alo[m] = __builtin_bit_cast(mf_s8, lv);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                typedef __attribute__((address_space(3))) mf_s4* lp;
                const char* p0 = bb + rbase[i];
                const mf_s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
                const mf_s4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0 + row4));
                if constexpr (F16) {
                    const mf_s8 bh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
#pragma unroll
                    for (int m = 0; m < MTW; ++m) {
                        if constexpr (!A16) acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_h8, alo[m]), __builtin_bit_cast(mf_h8, bh), acc[m][i], 0, 0, 0);
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_h8, ahi[m]), __builtin_bit_cast(mf_h8, bh), acc[m][i], 0, 0, 0);
                    }
                } else {
                    const mf_s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0 + plane_l));
                    const mf_s4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0 + plane_l + row4));
                    const mf_s8 bhi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const mf_s8 blo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
                    for (int m = 0; m < MTW; ++m) {
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, alo[m]), __builtin_bit_cast(mf_bf8, bhi), acc[m][i], 0, 0, 0);
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, ahi[m]), __builtin_bit_cast(mf_bf8, blo), acc[m][i], 0, 0, 0);
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, ahi[m]), __builtin_bit_cast(mf_bf8, bhi), acc[m][i], 0, 0, 0);
                    }
                }
            }
        }
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_comp += t - tk1; tk1 = t; }
    }

    if constexpr (F16) {
        // ---- first-order epilogue: y = u + o, o = ascale A u - mu u.  Leaves y (planes, optionally fp32), the column sums of o^2 (the
        // bound's q), the row sums of y^2 and the workgroup's share of the trace.
        const int h2 = lane >> 5, cl = lane & 31;
        const float emu = Out ? (float)exp(shift_d) : 1.f;
        float* red = reinterpret_cast<float*>(bufs);   // [MS][GT * 32] column sums of o^2
        float* rowred = red + MS * GT * 32;            // [NWN][32 MT] row sums of y^2
        static_assert((MS * GT * 32 + NWN * 32 * MT) * 4 <= NB * CHUNK, "the epilogue's reductions fit into the chunk buffers");
        float rn[MTW][16], d2t[NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) rn[m][v] = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (!tile_on[i]) continue;
            const int col = col0 + (wn * NT + i) * 32 + cl;
            float dot2 = 0.f;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                int rows[16];
                float u[16], yy[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) rows[v] = orow_l[32 * (wm * MTW + m) + (v & 3) + 8 * (v >> 2) + 4 * h2];
#pragma unroll
                for (int v = 0; v < 16; ++v) u[v] = U[(size_t)max(rows[v], 0) * Dpad + col];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float o = ascale * acc[m][i][v] - shift * u[v];
                    const float y = u[v] + o;
                    yy[v] = 0.f;
                    if (rows[v] >= 0) {
                        dot2 += o * o;
                        yy[v] = y * y;
                        if (Out) Out[(size_t)rows[v] * Dpad + col] = y * emu;  // the copy the API hands out is exp(A) u: it carries the factor
                    }
                    // y's bf16 halves, two columns per store: the even lane of a pair writes the hi halves of both columns, the odd lane the lo halves
                    const unsigned w = split_bf16(y);
                    const unsigned wp = dpp_u32<DPP_QUAD_XOR1>(w);
                    if (rows[v] >= 0) {
                        const unsigned val = (cl & 1) ? ((wp & 0xFFFFu) | (w << 16)) : ((w >> 16) | (wp & 0xFFFF0000u));
                        unsigned short* g = E.y_planes + (size_t)rows[v] * Dpad * 2 + (size_t)(col >> 5) * 64 + ((cl & 1) ? 32 : 0) + (col & 30);
                        *reinterpret_cast<unsigned*>(g) = val;
                    }
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) rn[m][v] += group_sum(yy[v], 32);
            }
            float a, b;
            rows32(dot2, a, b);
            d2t[i] = a + b;
        }
        __syncthreads();  // every wave has left the chunk buffers (the loads, products and stores above did not wait for that)
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (tile_on[i] && h2 == 0) red[(wm * GT + wn * NT + i) * 32 + cl] = d2t[i];
        if (cl == 0) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int v = 0; v < 16; ++v) rowred[wn * (32 * MT) + 32 * (wm * MTW + m) + (v & 3) + 8 * (v >> 2) + 4 * h2] = rn[m][v];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < ng * 32; t += THREADS) {
            float d2 = 0.f;
#pragma unroll
            for (int g = 0; g < MS; ++g) d2 += red[g * GT * 32 + t];
            partial_o2[(size_t)rb * Dpad + col0 + t] = (double)d2;
        }
        if (wv == 0) {  // a block has at most 64 rows: one wave adds them to the totals and folds the trace share
            double s = 0.0;
            if (lane < nrows) {
                float f = 0.f;
#pragma unroll
                for (int g = 0; g < NWN; ++g) f += rowred[g * (32 * MT) + lane];
                s = (double)f;
                atomicAdd(reinterpret_cast<unsigned long long*>(E.dfx) + orow_l[lane], (unsigned long long)__double2ll_rn(s * SDM_FX));
            }
            s = wave_sum(s);
            if (lane == 0) E.tr_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
        }
        if (stamps && lane == 0) {  // diagnostic runs: the same record as below
            const unsigned long long te = __builtin_amdgcn_s_memtime();
            unsigned long long* o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wv) * 8;
            o[0] = t_pro; o[1] = acc_wait; o[2] = acc_issue; o[3] = acc_comp; o[4] = te - tk0; o[5] = (unsigned long long)NC; o[6] = tk0; o[7] = te - tk1;
        }
        return;
    }
    // ---- epilogue on the accumulator layout: lane = column (lane & 31), register v = row (v & 3) + 8 (v >> 2) + 4 (lane >> 5).
    // Rows past the block's last are computed on row 0's address and dropped at the store: no branch around a load, so a tile's
    // 16 loads of u are in flight together.
    const int h2 = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (!tile_on[i]) continue;
        const int col = col0 + (wn * NT + i) * 32 + cl;
        float dot = 0.f, dot2 = 0.f;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            int rows[16];
            float u[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) rows[v] = orow_l[32 * (wm * MTW + m) + (v & 3) + 8 * (v >> 2) + 4 * h2];
            float x2[16];
            if (MODE == SPMM_LANCZOS) {
#pragma unroll
                for (int v = 0; v < 16; ++v) u[v] = U[(size_t)max(rows[v], 0) * Dpad + col];
            }
            if (MODE == SPMM_AXPBY) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    u[v] = E.F[(size_t)max(rows[v], 0) * Dpad + col];
                    x2[v] = E.X2[(size_t)max(rows[v], 0) * Dpad + col];
                }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float o;
                if (MODE == SPMM_LANCZOS) {
                    o = ascale * acc[m][i][v] - shift * u[v];  // shift is 0 unless the recurrence runs on A - mu I
                    if (rows[v] >= 0) {
                        dot += u[v] * o;
                        dot2 += o * o;
                    }
                } else if (MODE == SPMM_AXPBY) {
                    o = ascale * acc[m][i][v] + shift * u[v] + E.c3 * x2[v];
                } else {
                    o = ascale * acc[m][i][v];
                }
                if (rows[v] >= 0) {
                    const size_t at = (size_t)rows[v] * Dpad + col;
                    Out[at] = o;
                    if (MODE != SPMM_LANCZOS && E.out_planes) {
                        const unsigned w = split_bf16(o);
                        E.out_planes[at] = (unsigned short)(w >> 16);
                        *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(E.out_planes) + plane_bytes + at * 2) = (unsigned short)(w & 0xFFFFu);
                    }
                }
            }
        }
        if (MODE == SPMM_LANCZOS) {
            float a, b;
            rows32(dot, a, b);
            const float d1 = a + b;
            rows32(dot2, a, b);
            const float d2 = a + b;
            if (MS == 1) {
                if (h2 == 0) {
                    partial[(size_t)rb * Dpad + col] = (double)d1;
                    if (shifted) partial_o2[(size_t)rb * Dpad + col] = (double)d2;
                }
            } else if (h2 == 0) {  // the row-tile groups meet in LDS below (the chunk buffers are free now)
                float* red = reinterpret_cast<float*>(bufs);
                red[((wm * GT + wn * NT + i) * 32 + cl) * 2] = d1;
                red[((wm * GT + wn * NT + i) * 32 + cl) * 2 + 1] = d2;
            }
        }
    }
    if (MODE == SPMM_LANCZOS && MS > 1) {
        __syncthreads();  // every wave is past its last chunk: its reads of the buffers are done
        const float* red = reinterpret_cast<const float*>(bufs);
        for (int t = threadIdx.x; t < ng * 32; t += THREADS) {
            float d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int g = 0; g < MS; ++g) {
                d1 += red[((g * GT) * 32 + t) * 2];
                d2 += red[((g * GT) * 32 + t) * 2 + 1];
            }
            partial[(size_t)rb * Dpad + col0 + t] = (double)d1;
            if (shifted) partial_o2[(size_t)rb * Dpad + col0 + t] = (double)d2;
        }
    }
    if (stamps && lane == 0) {
        const unsigned long long te = __builtin_amdgcn_s_memtime();
        unsigned long long* o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wv) * 8;
        o[0] = t_pro; o[1] = acc_wait; o[2] = acc_issue; o[3] = acc_comp; o[4] = te - tk0; o[5] = (unsigned long long)NC; o[6] = tk0; o[7] = te - tk1;
    }
}


// ---- X on the pattern (mmw.py:182-194) on the matrix cores ---------------------------------------------------------------------
// The SDDMM of the same blocks: C = Y[rows] Y[union]^T as dense 32 x 32 tiles over the two bf16 halves of Y = exp(L/2) R
// (written by the Lanczos combination), then only the pattern's entries of every tile are divided by the trace and stored.
// Both operands are k-contiguous here (k runs along a row of Y), so fragments are plain 16-byte LDS reads, no transpose.
// One workgroup = (row block, 4 union tiles = 128 union rows): 4 MT waves, wave (wm, wn) owns row tile wm x union tile wn.
// Y comes as bf16 halves interleaved per 32 columns: a row's 128-byte group g holds the hi halves of columns 32 g .. 32 g + 31
// (64 bytes) and then their lo halves (k_lz_combine writes it so), so that a chunk = 32 columns of Y (2 k-steps) is one full
// 128-byte line per row, of the block's 32 MT rows and of the 128 union rows, brought in by LDS-DMA like the SpMM's chunks.
// A row's eight 16-byte slots are rotated by (row >> 1) -- applied on the DMA's source side -- so that the 16 lanes of a
// ds_read_b128 service group, which read 16 different rows at the same slot, hit 16 different bank quads.
// The diagonal comes from the exact row norms the combination made (d / tr), not from the split product.
// Error of an off-diagonal entry: <= 3 * 2^-17 |y_a| |y_b| (the two-half split), i.e. ~1e-5 of the diagonal scale.
struct SdMfmaDev {
    const int* tbase;             // [nb+1] first tile of each block
    const int* tptr;              // [tiles+1] entry ranges
    const unsigned short* trc;    // row in tile << 5 | column in tile
    const unsigned short* tmask;  // [tiles][64] accumulator mask of the listed entries (blocking.h, m_tmask)
    int nedges;                   // listed entries in all = undirected edges of the pattern
};
// X leaves this kernel in TILE ORDER: slot w of the tile lists holds edge w (each undirected edge once -- X is symmetric --, so a wave's
// entries are one contiguous, fully coalesced run), the K diagonal entries follow by row id.  The running sum of X (mmw.py:77) is kept in
// the same order and updated right here, while the value is in a register.  Everything else reads X through the slot map of the blocking
// (m_e2w) or asks the handle for the CSR-ordered copy (Solver::x_to_csr).
constexpr int SDM_GT = 4;   // union tiles per workgroup
constexpr int SDM_KC = 2;   // k-steps per chunk (a 128-byte line per row: 64 bytes of hi halves, 64 of lo halves)
template <int MT> constexpr int sdm_rows() { return 32 * MT + 32 * SDM_GT; }
template <int MT> constexpr int sdm_chunk_bytes() { return 2 * sdm_rows<MT>() * 32 * SDM_KC; }
template <int MT, int NB = 2> constexpr int sdm_lds_bytes() { return sdm_rows<MT>() * 4 + NB * sdm_chunk_bytes<MT>(); }  // NB chunks resident

// The first-order exponential (SPMM_FIRST) is certified after the fact, off the critical path: a few spare workgroups of the next
// iteration's k_dual_h launch (or k_first_verify at the end of a chunk) fold the column sums of o^2 (the product's slabs), of u^2 and
// of (u - fp16(u))^2 (the sketch's slabs) into the largest per-column bound, leave it where the Lanczos steps leave theirs (conv[1],
// m_eff = 1: the host's chunk logic reads the same fields) and raise the replay flag when the bound misses the tolerance or the plan
// does not allow a single substep.  The next plan (the LOSS pass that follows) resets the fields only afterwards.
//
// What is certified, per column u (relative to ||exp(A')u|| >= e^-rho ||u||):
//   truncation      q (rho / 2) e^rho,  q = ||A'u|| / ||u|| measured           (first_order_bound)
//   the fp16 plane  ||A (u - fp16(u))|| <= ||A||_2 ||u - fp16(u)||, with the rounding ||u - fp16(u)|| / ||u|| MEASURED by the sketch
//                   kernel (~0.4 * 2^-11 for Gaussian rows; the format's worst case would be 2^-11 plus the subnormal flush) and
//                   ||A||_2 <= max_i sum_j |a_ij| = absn from the plan
//   the matrix      ||dA u|| <= || |dA| ||_2 ||u|| <= c_A absn ||u|| + n_row 2^-45 ||u||: fp16 unit roundoff 2^-11 per entry of the
//                   one-half image (c_A = 2^-11), 2^-22 for hi + lo; entries of 2^20 L below fp16's normal range (|l| < 2^-34) err
//                   by at most 2^-45 each, at most MF_UNION_ROWS of them in a row.
// The fp32 accumulation of the products is not part of it (it is the arithmetic every fp32 handle computes in).
constexpr int FV_COLS = 8;  // columns per verification workgroup (64 bytes of a slab row per thread)
constexpr double F16_SUBNORMAL_ROW = 640.0 * 2.842170943040401e-14;  // MF_UNION_ROWS * 2^-45
struct FirstVerify {
    ExpmPlan* plan = nullptr;  // nullptr: no verification rides in this launch
    int* viol = nullptr;
    const double* o2 = nullptr;
    const double* u2 = nullptr;
    const double* du2 = nullptr;  // slabs of (u - fp16(u))^2, n_u2 of them; nullptr: not measured -- the format's worst case is used instead
                                  // (|du| <= 2^-11 |u| in fp16's normal range, <= 2^-25 below it: ||du||^2 <= 2^-22 ||u||^2 + rows 2^-50)
    int rows = 0;                 // K
    int n_o2 = 0, n_u2 = 0, Dpad = 0;
    int nwg = 0;               // verification workgroups: Dpad / FV_COLS
    double cA = 0.0;           // relative rounding of the matrix image the product read (see above)
    double du_scale = 1.0;     // tests only (MMW_FV_DU_SCALE): inflates the measured rounding to force a miss
};
// workgroup `wg` (4 to 16 wavefronts) takes the columns [FV_COLS wg, FV_COLS (wg + 1)).  Lane (q = lane & 7, slice = lane >> 3) of wave w sums
// column q over the slab rows 8 w + slice, + 8 waves, ...: a load covers 8 rows x 64 bytes, and a wave's 24 sums meet in three strided
// reductions (lanes of equal q) instead of 24 whole-wave ones.  Fixed order throughout.
__device__ inline void first_verify(const FirstVerify& V, int wg) {
    __shared__ double shv[3 * FV_COLS][16];
    const int BLOCK_V = (int)blockDim.x, NWV = BLOCK_V >> 6;  // 4 ... 16 wavefronts
    const int c0 = wg * FV_COLS;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, q8 = lane & 7, sl = lane >> 3;
    double a_o2 = 0.0, a_u2 = 0.0, a_du = 0.0;
    for (int b = wv * 8 + sl; b < V.n_o2; b += NWV * 8) a_o2 += V.o2[(size_t)b * V.Dpad + c0 + q8];
    for (int b = wv * 8 + sl; b < V.n_u2; b += NWV * 8) {
        a_u2 += V.u2[(size_t)b * V.Dpad + c0 + q8];
        if (V.du2) a_du += V.du2[(size_t)b * V.Dpad + c0 + q8];
    }
    a_o2 = stride_sum(a_o2, 8);
    a_u2 = stride_sum(a_u2, 8);
    a_du = stride_sum(a_du, 8);
    if (lane < FV_COLS) { shv[lane][wv] = a_o2; shv[FV_COLS + lane][wv] = a_u2; shv[2 * FV_COLS + lane][wv] = a_du; }
    __syncthreads();
    if (threadIdx.x < 64) {  // one column per lane: the double-precision tail of the eight columns runs side by side
        double e = 0.0, et = 0.0;  // the whole bound; its truncation part (what the host extrapolates with the norm's growth squared)
        if (threadIdx.x < FV_COLS) {
            const int q = (int)threadIdx.x;
            double o2 = 0.0, u2 = 0.0, du2 = 0.0;
            for (int w = 0; w < NWV; ++w) { o2 += shv[q][w]; u2 += shv[FV_COLS + q][w]; du2 += shv[2 * FV_COLS + q][w]; }
            if (u2 > 0.0) {
                const double rho = V.plan->rho, absn = V.plan->absn;
                et = first_order_bound(sqrt(o2 / u2), rho);
                if (!V.du2) du2 = F16_UNIT * F16_UNIT * u2 + (double)V.rows * 8.881784197001252e-16;  // 2^-50
                e = et + exp(rho) * (absn * (V.du_scale * sqrt(du2 / u2) + V.cA) + F16_SUBNORMAL_ROW);
            }
            if (!(e >= 0.0)) e = et = 1e300;  // NaN
        }
        const double best = wave_max(e), best_t = wave_max(et);
        if (threadIdx.x == 0) {
            auto up = [](double x) {  // float bits, rounded up
                float f = (float)x;
                if (!(f >= 0.0f)) f = __uint_as_float(0x7f800000u);
                if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + 1u);
                return __float_as_uint(f);
            };
            atomicMax(&V.plan->conv[1], up(best));  // maxima: the order of arrival does not matter (the plan zeroed both)
            atomicMax(&V.plan->first_est, up(best_t));
            if (wg == 0) V.plan->m_eff = 1;
            if (!(best <= V.plan->tol) || !V.plan->apost || V.plan->nsub != 1 || V.plan->overflow) atomicOr(V.viol, VIOL_FIRST);
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_first_verify(FirstVerify V) { first_verify(V, (int)blockIdx.x); }

template <int MT, int NB = 2>
__global__ __launch_bounds__(4 * MT * 64)
void k_sddmm_mfma(MfmaDev M, SdMfmaDev S, int K, int Dpad, const char* __restrict__ Ypl, const float* __restrict__ d,
                  const double* __restrict__ tr_part, int ntr, float* __restrict__ xs_val, float* __restrict__ xs_avg, int accumulate,
                  long long* __restrict__ rsfx = nullptr /* [K], zero at launch */, const long long* __restrict__ dfx = nullptr,
                  unsigned long long* __restrict__ stamps = nullptr /* diagnostic runs: 8 shader-clock sums per wave */,
                  FirstVerify V = FirstVerify{}) {
    // dfx: the row norms as 2^-40 fixed-point totals (SPMM_FIRST) instead of `d`
    // V.plan: the launch carries 8 more columns of workgroups (one per XCD and union-tile group) that take no tiles: they certify the
    // first-order exponential this X comes from (first_verify; its slabs are complete once the product has ended) -- short independent
    // work in the slots the one round of tile workgroups leaves free, instead of riding on the DUAL phase's critical path.
    {
        const int nbx = ((M.nb + 7) >> 3) << 3;
        if ((int)blockIdx.x >= nbx) {
            if (!V.plan) return;
            const int nfv = 8 * (int)gridDim.y;
            for (int cg = ((int)blockIdx.x - nbx) + 8 * (int)blockIdx.y; cg < V.nwg; cg += nfv) {
                first_verify(V, cg);
                __syncthreads();  // the next column group reuses first_verify's LDS
            }
            return;
        }
    }
    unsigned long long tk0 = 0, tk1 = 0, acc_wait = 0, acc_issue = 0, acc_comp = 0, t_pro = 0, t_red = 0, t_store = 0;
    if (stamps) tk0 = __builtin_amdgcn_s_memtime();
    const int by = (int)blockIdx.y;  // union-tile group of this workgroup
    // rsfx: the DUAL phase's first step (mmw.py:133-134, the row sums of the off-diagonal X) leaves with the tiles: every wave sums
    // its tile's pattern entries per row straight from the accumulators, the workgroup's union tiles meet in LDS, and the part of
    // this union-tile group is added to the row's total as a 2^-40 fixed-point integer (|sum| < 2^10 for rows of <= 640 entries
    // of magnitude <= 1): integer atomics commute, so the total does not depend on the order the workgroups arrive in.
    // The LOSS pass of every iteration zeroes the totals (k_loss), k_dual_h reads them.
    constexpr int NW = 4 * MT, THREADS = NW * 64;
    constexpr int RA = 32 * MT, R = sdm_rows<MT>();
    constexpr int CHUNK = sdm_chunk_bytes<MT>();
    constexpr int NP = R / 8;  // 1-KiB pieces per chunk: 8 rows x 128 bytes
    constexpr int NJ = (NP + NW - 1) / NW;
    typedef __attribute__((address_space(3))) void* lds_vp;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int* rows_l = reinterpret_cast<int*>(smem_raw);  // [R] global row ids: the block's rows, then this workgroup's union rows
    char* bufs = smem_raw + R * 4;
    const unsigned bufs_l = (unsigned)(size_t)(lds_vp)bufs;
    __shared__ double sh_tr[NW];

    const int per = (M.nb + 7) >> 3;
    const int rb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (rb >= M.nb) return;
    const int* dsc = M.desc + (size_t)rb * 8;
    const int q0 = dsc[0], nrows = dsc[1], nun = dsc[5];
    const int ntile = (nun + 31) >> 5;
    // X is symmetric: an edge is computed by the block of the endpoint that comes first in the blocked order.  The union is sorted by
    // that order, so the tiles before dsc[6] hold only columns of earlier blocks -- their edges are those blocks' -- and are skipped.
    const int ut0 = dsc[6] + by * SDM_GT;  // first union tile of this workgroup
    if (ut0 >= ntile) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 2, wn = wv & 3;

    double tsum = 0.0;
    for (int i = threadIdx.x; i < ntr; i += THREADS) tsum += tr_part[i];
    tsum = wave_sum(tsum);
    if (lane == 0) sh_tr[wv] = tsum;
    for (int i = threadIdx.x; i < R; i += THREADS) {
        int g;
        if (i < RA) g = M.order[q0 + min(i, nrows - 1)];  // rows past the block's last repeat it; their outputs are never referenced
        else g = M.un_fixed[(size_t)rb * MF_UNION_ROWS + min(ut0 * 32 + (i - RA), MF_UNION_ROWS - 1)];
        rows_l[i] = g;
    }
    __syncthreads();
    double tr = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tr += sh_tr[w];
    tr /= (double)K;
    // what the wave's tail needs of the tile lists is requested now: these dependent loads would otherwise sit behind the last product
    int tw0 = 0, tw1 = 0;
    unsigned mk = 0u;
    if (ut0 + wn < ntile) {
        const int t = S.tbase[rb] + (ut0 + wn) * MT + wm;
        tw0 = S.tptr[t];
        tw1 = S.tptr[t + 1];
        if (rsfx) mk = (unsigned)S.tmask[(size_t)t * 64 + lane];
    }

    // DMA pieces of this wave: piece i = wv + NW j covers rows 8 i .. 8 i + 7; lane L -> row 8 i + (L >> 3), position L & 7,
    // which holds source slot (position - (row >> 1)) & 7 of the row's 128-byte chunk line (slots 0-3 hi halves, 4-7 lo halves)
    const unsigned pitch = (unsigned)Dpad * 4u;
    const char* pbase[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = wv + NW * j;
        pbase[j] = Ypl;
        if (i < NP) {
            const int row = 8 * i + (lane >> 3), pos = lane & 7;
            const int slot = (pos - (row >> 1)) & 7;
            pbase[j] = Ypl + ((size_t)(unsigned)rows_l[row] * pitch + (size_t)slot * 16u);
        }
    }
    const int cw = wv < NP ? (NP - wv + NW - 1) / NW : 0;
    auto issue = [&](int c) {
        const unsigned dst_l = bufs_l + (unsigned)((c % NB) * CHUNK);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (j < cw) mf_dma16(pbase[j] + (size_t)c * (64 * SDM_KC), dst_l + (unsigned)((wv + NW * j) * 1024));
    };
    // fragment addresses: lane (r = lane & 31, h = lane >> 5) reads row `ra` (A) / `rbw` (B), slot 4 p + 2 kk + h of half p,
    // stored at position (slot + (row >> 1)) & 7
    const int r = lane & 31, h = lane >> 5;
    const int ra = 32 * wm + r, rbw = RA + 32 * wn + r;
    unsigned offA[SDM_KC][2], offB[SDM_KC][2];
#pragma unroll
    for (int kk = 0; kk < SDM_KC; ++kk)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            offA[kk][p] = (unsigned)(ra * 128 + (((4 * p + 2 * kk + h) + (ra >> 1)) & 7) * 16);
            offB[kk][p] = (unsigned)(rbw * 128 + (((4 * p + 2 * kk + h) + (rbw >> 1)) & 7) * 16);
        }
    mf_f16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const int NC = Dpad / (16 * SDM_KC);
#pragma unroll
    for (int c = 0; c < NB - 1; ++c)
        if (c < NC) issue(c);
    if (stamps) { tk1 = __builtin_amdgcn_s_memtime(); t_pro = tk1 - tk0; }
    for (int c = 0; c < NC; ++c) {
        mf_wait_vmcnt(cw * min(NC - 1 - c, NB - 2));  // this wave's pieces of chunk c have landed: at most the younger chunks' are outstanding
        __builtin_amdgcn_s_barrier();  // chunk c landed for everyone; everyone left the buffer of chunk c - 1
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_wait += t - tk1; tk1 = t; }
        if (c + NB - 1 < NC) issue(c + NB - 1);
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_issue += t - tk1; tk1 = t; }
        const char* cb = bufs + (c % NB) * CHUNK;
#pragma unroll
        for (int kk = 0; kk < SDM_KC; ++kk) {
            const uint4 ah = *reinterpret_cast<const uint4*>(cb + offA[kk][0]);
            const uint4 al = *reinterpret_cast<const uint4*>(cb + offA[kk][1]);
            const uint4 bh = *reinterpret_cast<const uint4*>(cb + offB[kk][0]);
            const uint4 bl = *reinterpret_cast<const uint4*>(cb + offB[kk][1]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, al), __builtin_bit_cast(mf_bf8, bh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, ah), __builtin_bit_cast(mf_bf8, bl), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf8, ah), __builtin_bit_cast(mf_bf8, bh), acc, 0, 0, 0);
        }
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_comp += t - tk1; tk1 = t; }
    }
    static_assert(NW * 4096 + 4 * 32 * MT * 4 <= NB * CHUNK, "tiles and row-sum parts fit into the chunk buffers");
    // the wave's entry list (place in the tile of every listed entry) is static: the first 256 entries are requested here, so that
    // their round trip passes under the reductions, the barriers and the tile's way through LDS
    constexpr int EPL = 4;  // entries per lane and round
    unsigned rc[EPL];
    float xa[EPL];
    auto fetch_list = [&](int base) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int w = base + k * 64 + lane;
            rc[k] = w < tw1 ? (unsigned)S.trc[w] : 0xFFFFu;
            xa[k] = (w < tw1 && accumulate) ? xs_avg[w] : 0.f;
        }
    };
    fetch_list(tw0);
    float rsv[16];
    float csum = 0.f;
    if (rsfx) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float m = ((mk >> v) & 1u) ? acc[v] : 0.f;
            csum += m;                   // the same entries seen from their columns: the mirror entries' share of THOSE rows' sums
            rsv[v] = group_sum(m, 32);   // over the 32 columns of the tile
        }
        float ca, cb;
        rows32(csum, ca, cb);            // a column's 32 rows sit in lanes c and c + 32
        csum = ca + cb;
    }
    if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_red = t - tk1; tk1 = t; }
    __syncthreads();  // every wave is done with the chunk buffers: they now hold the 32 x 32 output tiles, one per wave
    float* rsl = reinterpret_cast<float*>(bufs) + NW * 1024;  // [4 union tiles][32 MT rows] behind the tiles
    if (rsfx && (lane & 31) == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) rsl[wn * RA + 32 * wm + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)] = rsv[v];
    }
    float* tile = reinterpret_cast<float*>(bufs) + wv * 1024;
#pragma unroll
    for (int v = 0; v < 16; ++v) tile[((v & 3) + 8 * (v >> 2) + 4 * h) * 32 + r] = acc[v];  // C layout: column = lane & 31
    if (rsfx) __syncthreads();  // tiles and row-sum parts of every wave
    else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const float inv_tr = (float)(1.0 / tr);
    for (int base = tw0; base < tw1; base += 64 * EPL) {
        if (base != tw0) fetch_list(base);
#pragma unroll
        for (int k = 0; k < EPL; ++k)
            if (rc[k] != 0xFFFFu) {
                const int w = base + k * 64 + lane;
                const float x = tile[rc[k]] * inv_tr;
                xs_val[w] = x;
                if (accumulate) xs_avg[w] = xa[k] + x;
            }
    }
    if (rsfx && lane < 32 && csum != 0.f) {  // (exactly zero where the column has no listed entry in this tile)
        const long long q = __double2ll_rn((double)csum / tr * SDM_FX);
        atomicAdd(reinterpret_cast<unsigned long long*>(rsfx) + rows_l[RA + 32 * wn + lane], (unsigned long long)q);
    }
    if (by == 0)  // the diagonal of the block's rows from the exact row norms
        for (int i = threadIdx.x; i < nrows; i += THREADS) {
            const int row = rows_l[i];
            const float x = (float)((dfx ? (double)dfx[row] * (1.0 / SDM_FX) : (double)d[row]) / tr);
            xs_val[S.nedges + row] = x;
            if (accumulate) xs_avg[S.nedges + row] += x;
        }
    if (rsfx)
        for (int i = threadIdx.x; i < nrows; i += THREADS) {
            const float sum4 = ((rsl[i] + rsl[RA + i]) + rsl[2 * RA + i]) + rsl[3 * RA + i];
            const long long q = __double2ll_rn((double)sum4 / tr * SDM_FX);
            atomicAdd(reinterpret_cast<unsigned long long*>(rsfx) + rows_l[i], (unsigned long long)q);  // integers: the order of arrival does not matter
        }
    if (stamps && lane == 0) {
        const unsigned long long te = __builtin_amdgcn_s_memtime();
        t_store = te - tk1;
        unsigned long long* o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wv) * 8;
        o[0] = t_pro; o[1] = acc_wait; o[2] = acc_issue; o[3] = acc_comp; o[4] = te - tk0; o[5] = t_red; o[6] = tk0; o[7] = t_store;
    }
}

}  // namespace mmw
