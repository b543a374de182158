// Device kernels for the action of exp(A) on a dense K x D block (A symmetric, fixed CSR pattern):
// the CSR SpMM, the per-column Lanczos recurrences, the small tridiagonal exponentials and the
// shifted-Taylor variant.  Replaces scipy.sparse.linalg.expm_multiply as called from
// mmw.expm_half_randsk (sim_src/alg/mmw.py:224-229).
//
// Block layout: row-major [K, Dpad], Dpad a multiple of the 16-byte vector width.  A row is covered by
// LPR = Dpad / VEC lanes, each holding VEC consecutive columns (one 16-B load).  When LPR <= 32 it is a
// power of two and one wavefront processes G = 64/LPR nonzeros of a matrix row at a time (one lane
// group per nonzero: every gather is a fully coalesced LPR*16-byte row segment); for wider blocks one
// lane covers NCH chunks 64 lanes apart.  All HBM-bound: ~0.2-2 flop/byte, so no MFMA here.
#pragma once
#include "blk_config.h"
#include "device_utils.h"

namespace mmw {

struct BlockLayout {
    int D, Dpad, LPR, G, NCH;
};

struct ExpmPlan {      // written by k_plan, read by every expm kernel
    double rho;        // bound on || A - mu I ||_1
    double mu;         // trace(A)/K
    double tol;
    int m;             // Krylov order / Taylor degree per substep (never more than the host launched)
    int nsub;          // substeps (time stepping exp(A) = exp(A/nsub)^nsub)
    int overflow;      // 1 when max_order could not meet tol
    int m_apriori;     // the order the a-priori bound asks for
    int apost;         // 1: the Lanczos steps carry an a-posteriori error estimate and stop as soon as it meets tol
    int m_eff;         // steps the last application actually used (written by the combination)
    int mfma_ok;       // 1: the two-half bf16 split of the matrix-core SpMM (kernels_mfma.h) keeps the product within tol
    int f16_ok;        // 1: the first-order product's single fp16 plane of u is worth launching: F16_PLANE_EXPECT absn <= tol.  A prefilter --
                       //    the form is CERTIFIED after the fact from the plane's measured rounding (kernels_mfma.h, first_verify)
    int f16a_ok;       // 1: ... and with the matrix in ONE fp16 half as well (SPMM_FIRST16): (F16_PLANE_EXPECT + F16_UNIT) absn <= tol
    double absn;       // max_i sum_j |a_ij| of the scaled matrix (the bound behind mfma_ok)
    // Lagged planning (the loop's optimistic chunks): the row sums of the matrix are made one iteration late, inside the DUAL
    // pass that reads the same rows, and the bounds for the matrix that is multiplied are extrapolated from the last two
    // matrices seen; the next plan checks that the extrapolation covered it (else *viol: the chunk is replayed with exact plans).
    int lagged;        // 1: rho / absn / mu above are extrapolated bounds, to be verified by the next plan
    int h_iter;        // iteration index of the last matrix whose sums were seen (-1: none, sums of the zero matrix)
    double h_pp, h_pm, h_mu;  // its sums: max_i (d_i + o_i), max_i (o_i - d_i), trace / K
    double g_pp, g_pm, g_mu;  // growth per iteration of those sums (last finite difference)
    unsigned conv[MAX_ORDER + 2];  // conv[j]: float bits of the largest per-column estimate after j steps (valid once step j's scalars ran)
    unsigned first_est;  // float bits of the largest per-column error bound of the first-order form y = u + (A - mu I) u (first_order_bound)
};
// why an optimistic chunk has to be replayed (bits of the violation word the host reads when it settles the chunk)
enum : int { VIOL_ORDER = 1 /* the launched Lanczos steps did not meet the tolerance */, VIOL_LAGGED = 2 /* an extrapolated plan did not cover its matrix */,
             VIOL_PLAN = 4 /* substeps / overflow */, VIOL_SOFTMAX = 8 /* the fused softmax's shift ran away */, VIOL_OPERANDS = 16 /* 16-bit operand gate */,
             VIOL_FIRST = 32 /* the first-order form's certificate missed */ };
constexpr double F16_UNIT = 4.8828125e-4;                 // 2^-11, the unit roundoff of fp16 (11 significant bits)
constexpr double F16_PLANE_EXPECT = 0.45 * F16_UNIT;      // ||u - fp16(u)|| / ||u|| of a row-normalised Gaussian block: 0.434 * 2^-11 measured
constexpr double F16_CA_TWO = 4.8e-7;                      // the matrix as fp16 hi + lo: 2 * 2^-22 relative per entry
// exp(A')u = u + A'u + R with ||R|| <= sum_{k>=2} rho^(k-1) ||A'u|| / k! <= ||A'u|| (rho/2) e^rho for any rho >= ||A'||_2 (the 1-norm bound of
// the symmetric A' is one), and ||exp(A')u|| >= e^-rho ||u||: the relative error of the first-order form is at most q (rho/2) e^(2 rho),
// q = ||A'u|| / ||u|| measured on the column.
__device__ __forceinline__ double first_order_bound(double q, double rho) { return q * 0.5 * rho * exp(2.0 * rho); }
// Steps that run, given the estimates of the steps <= upto that have completed.  The a-priori order is a bound from the
// 1-norm; the estimate (k_lz_scalars) uses what the recurrence has seen of the operator and typically stops 1-2 steps
// earlier.  Every kernel of a later step evaluates this and returns at once.
__device__ __forceinline__ int plan_steps(const ExpmPlan* p, int upto) {
    int m = p->m;
    if (p->apost) {
        const unsigned t = __float_as_uint((float)p->tol);
        for (int i = 1; i <= upto && i < m; ++i)
            if (p->conv[i] <= t) {
                m = i;
                break;
            }
    }
    return m;
}

enum { SPMM_PLAIN = 0, SPMM_LANCZOS = 1, SPMM_TAYLOR = 2, SPMM_AXPBY = 3, SPMM_FIRST = 4, SPMM_FIRST16 = 5 /* the last two: kernels_mfma.h only */ };

// Out = ascale * A * U (+ mode-specific fused epilogue).  One wavefront per matrix row, grid-stride.
//   SPMM_LANCZOS: also partial[block][col] = sum_rows U[row,col] * Out[row,col]   (alpha numerators)
//   SPMM_TAYLOR : Out = (ascale*A*U - shift*U) * inv_k ;  F += Out                (one Taylor term)
//   SPMM_AXPBY  : Out = ascale*A*U + shift*F + inv_k*X2                           (Chebyshev recurrences)
template <typename T, int NCH, int MODE>
__global__ __launch_bounds__(BLOCK) void k_spmm(int K, BlockLayout lay, const int* __restrict__ indptr,
                                                const int* __restrict__ col, const T* __restrict__ val,
                                                const T* __restrict__ U, T* __restrict__ Out, T* __restrict__ F,
                                                const T* __restrict__ X2, double ascale, double shift, double inv_k,
                                                double* __restrict__ partial, const ExpmPlan* __restrict__ plan, int step,
                                                double* __restrict__ partial_o2) {
    constexpr int VEC = V16<T>::N;
    bool shifted = false;  // Lanczos on A - mu I with column sums of squares of the product (a-posteriori stop, k_lz_scalars)
    if (plan) {  // device-side order: steps beyond the planned Krylov order are no-ops
        if (MODE == SPMM_LANCZOS) {
            if (step > plan_steps(plan, step - 1)) return;
            shifted = plan->apost != 0 && partial_o2 != nullptr;
            if (shifted) shift = plan->mu;
        } else if (step > plan->m) return;
        if (MODE == SPMM_TAYLOR) shift = plan->mu / plan->nsub;
    }
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int LPR = lay.LPR, G = lay.G, Dpad = lay.Dpad;
    const int g = (NCH == 1) ? lane / LPR : 0;
    const int lig = (NCH == 1) ? lane - g * LPR : lane;
    const bool active = g < G;
    double dot[NCH][VEC], dot2[NCH][VEC];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) dot[c][v] = dot2[c][v] = 0.0;

    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < K; row += gridDim.x * WAVES_PER_BLOCK) {
        T acc[NCH][VEC];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[c][v] = T(0);
        const int beg = indptr[row], end = indptr[row + 1];
        if (active) {
            int i = beg + g;
            // 4 independent gathers in flight per lane group
            for (; i + 3 * G < end; i += 4 * G) {
                int cc[4];
                T vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    cc[u] = col[i + u * G];
                    vv[u] = val[i + u * G];
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (lig + 64 * c < LPR) {
                        T x[4][VEC];
#pragma unroll
                        for (int u = 0; u < 4; ++u) load16(U + (size_t)cc[u] * Dpad + (size_t)(lig + 64 * c) * VEC, x[u]);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[c][v] += vv[u] * x[u][v];
                    }
                }
            }
            for (; i < end; i += G) {
                const int cc = col[i];
                const T vv = val[i];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (lig + 64 * c < LPR) {
                        T x[VEC];
                        load16(U + (size_t)cc * Dpad + (size_t)(lig + 64 * c) * VEC, x);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[c][v] += vv * x[v];
                    }
                }
            }
        }
        if (NCH == 1 && G > 1) {  // fold the G lane groups (LPR is a power of two here)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                acc[0][v] = stride_sum(acc[0][v], LPR);
        }
        if (g == 0) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (lig + 64 * c < LPR) {
                    const size_t off = (size_t)row * Dpad + (size_t)(lig + 64 * c) * VEC;
                    T o[VEC];
                    if (MODE == SPMM_PLAIN) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) o[v] = (T)(ascale * (double)acc[c][v]);
                    } else if (MODE == SPMM_LANCZOS) {
                        T u[VEC];
                        load16(U + off, u);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            o[v] = (T)(ascale * (double)acc[c][v] - (MODE == SPMM_LANCZOS ? shift : 0.0) * (double)u[v]);
                            dot[c][v] += (double)u[v] * (double)o[v];
                            dot2[c][v] += (double)o[v] * (double)o[v];
                        }
                    } else if (MODE == SPMM_AXPBY) {
                        T f[VEC], x2[VEC];
                        load16(F + off, f);
                        load16(X2 + off, x2);
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
                            o[v] = (T)(ascale * (double)acc[c][v] + shift * (double)f[v] + inv_k * (double)x2[v]);
                    } else {
                        T u[VEC], f[VEC];
                        load16(U + off, u);
                        load16(F + off, f);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            o[v] = (T)((ascale * (double)acc[c][v] - shift * (double)u[v]) * inv_k);
                            f[v] += o[v];
                        }
                        store16(F + off, f);
                    }
                    store16(Out + off, o);
                }
            }
        }
    }
    if (MODE == SPMM_LANCZOS) {
        extern __shared__ __attribute__((aligned(16))) char smem_raw[];
        double* sh = reinterpret_cast<double*>(smem_raw);  // [WAVES_PER_BLOCK][Dpad]
        if (g == 0) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (lig + 64 * c < LPR)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) sh[wib * Dpad + (lig + 64 * c) * VEC + v] = dot[c][v];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < Dpad; c += BLOCK) {
            double s = 0.0;
            for (int w = 0; w < WAVES_PER_BLOCK; ++w) s += sh[w * Dpad + c];
            partial[(size_t)blockIdx.x * Dpad + c] = s;
        }
        if (shifted) {  // second slab: column sums of squares of the product
            __syncthreads();
            if (g == 0) {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (lig + 64 * c < LPR)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) sh[wib * Dpad + (lig + 64 * c) * VEC + v] = dot2[c][v];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < Dpad; c += BLOCK) {
                double s = 0.0;
                for (int w = 0; w < WAVES_PER_BLOCK; ++w) s += sh[w * Dpad + c];
                partial_o2[(size_t)blockIdx.x * Dpad + c] = s;
            }
        }
    }
}

// ---- SpMM with the Krylov block staged in LDS, for small K without locality (BASELINE configs[1] / [3]: N = 2 000, 5 % dense) -----
// The generic kernel gathers a dense row of U from L2 for every nonzero (400 k nonzeros x 512 B at N = 2 000, D = 64, fp64: 205 MB per
// product for a 1 MB block).  When a column slice of the WHOLE block fits the CU's LDS -- K rows x SB bytes, SB = 64 (8 fp64 / 16 fp32
// columns) up to K = 2 400, SB = 32 up to 4 800 -- a workgroup stages its slice once and serves every nonzero of its row range from
// LDS; the matrix (12 / 8 B per nonzero) is streamed once per slice.  Grid = (slices, row ranges); a wave takes a row (see the loop).
// The Lanczos epilogue reads u[row] from the staged slice.  Same epilogues and the same partial-slab
// layout as k_spmm: range r writes slab r, and range 0 clears the slabs [nranges, nslabs) that the caller folds as well.
constexpr int SLICE_LDS_MAX = 153600;   // bytes of LDS a slice may take (160 KB per CU, a little left for the reductions)
constexpr int SLICE_THREADS = 1024;
template <typename T, int MODE, int SB>
__global__ __launch_bounds__(SLICE_THREADS) void k_spmm_slice(int K, int Dpad, const int* __restrict__ indptr, const int* __restrict__ col, const T* __restrict__ val,
                                                              const T* __restrict__ U, T* __restrict__ Out, T* __restrict__ F, const T* __restrict__ X2,
                                                              double ascale, double shift, double inv_k, double* __restrict__ partial,
                                                              const ExpmPlan* __restrict__ plan, int step, double* __restrict__ partial_o2, int nslabs) {
    constexpr int CS = SB / (int)sizeof(T);  // columns per slice
    constexpr int NG = WAVE / CS;            // nonzeros per wave step
    constexpr int NWV = SLICE_THREADS / WAVE;
    bool shifted = false;
    if (plan) {
        if (MODE == SPMM_LANCZOS) {
            if (step > plan_steps(plan, step - 1)) return;
            shifted = plan->apost != 0 && partial_o2 != nullptr;
            if (shifted) shift = plan->mu;
        } else if (step > plan->m) return;
        if (MODE == SPMM_TAYLOR) shift = plan->mu / plan->nsub;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* us = reinterpret_cast<T*>(smem_raw);  // [K][CS]
    __shared__ double shd[2][NWV][CS];
    const int c0 = (int)blockIdx.x * CS, range = (int)blockIdx.y, nranges = (int)gridDim.y;
    {   // stage the slice: 16-byte pieces, a row's SB bytes contiguous in both places
        constexpr int Q = SB / 16;
        const char* src = reinterpret_cast<const char*>(U + c0);
        const size_t pitch = (size_t)Dpad * sizeof(T);
        for (int i0 = threadIdx.x; i0 < K * Q; i0 += 8 * SLICE_THREADS) {  // eight loads in flight per thread
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * SLICE_THREADS;
                const int row = i / Q, q = i - row * Q;
                t[u] = i < K * Q ? *reinterpret_cast<const float4*>(src + (size_t)row * pitch + q * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * SLICE_THREADS;
                if (i < K * Q) *reinterpret_cast<float4*>(smem_raw + (size_t)i * 16) = t[u];
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = (K + nranges - 1) / nranges;
    const int r0 = range * per, r1 = min(K, r0 + per);
    // A lane takes ONE nonzero at a time -- the row's entries are read 64 at a time, fully coalesced, four such reads in flight -- and
    // multiplies it into all CS columns of the slice (SB bytes of LDS).  The 64 x CS partial sums of a row then meet in a
    // reduce-scatter: each exchange step halves the columns a lane still carries, log2(CS) steps, then log2(64 / CS) plain ones.
    int colsel = 0;  // the column this lane ends up holding
#pragma unroll
    for (int half = CS / 2, bit = 32; half >= 1; half >>= 1, bit >>= 1) colsel += (lane & bit) ? half : 0;
    const bool owner = (lane & (NG - 1)) == 0;
    double dot = 0.0, dot2 = 0.0;
    // two rows per wave in flight (their entry reads are independent: one memory round trip for both)
    for (int rowA = r0 + wv; rowA < r1; rowA += 2 * NWV) {
        const int rowB = rowA + NWV;
        const bool hasB = rowB < r1;
        int beg[2], end[2];
        beg[0] = indptr[rowA]; end[0] = indptr[rowA + 1];
        beg[1] = hasB ? indptr[rowB] : 0; end[1] = hasB ? indptr[rowB + 1] : 0;
        T acc[2][CS];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < CS; ++k) acc[r][k] = T(0);
        const int len = max(end[0] - beg[0], end[1] - beg[1]);
        for (int o0 = lane; o0 - lane < len; o0 += 4 * WAVE) {
            int cc[2][4];
            T vv[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = beg[r] + o0 + u * WAVE;
                    const bool ok = e < end[r];
                    cc[r][u] = ok ? col[e] : 0;
                    vv[r][u] = ok ? val[e] : T(0);
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4* src = reinterpret_cast<const float4*>(smem_raw + (size_t)cc[r][u] * SB);
#pragma unroll
                    for (int q = 0; q < SB / 16; ++q) {
                        const float4 x4 = src[q];
                        constexpr int PER = 16 / (int)sizeof(T);
                        T x[PER];
                        __builtin_memcpy(x, &x4, 16);
#pragma unroll
                        for (int k = 0; k < PER; ++k) acc[r][q * PER + k] += vv[r][u] * x[k];
                    }
                }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int bit = 32;
#pragma unroll
            for (int half = CS / 2; half >= 1; half >>= 1, bit >>= 1) {
                const bool up = (lane & bit) != 0;
#pragma unroll
                for (int k = 0; k < half; ++k) {
                    const T mine = up ? acc[r][k + half] : acc[r][k];
                    const T other = up ? acc[r][k] : acc[r][k + half];
                    acc[r][k] = mine + lane_xor(other, bit);
                }
            }
#pragma unroll
            for (int b2 = NG / 2; b2 >= 1; b2 >>= 1) acc[r][0] += lane_xor(acc[r][0], b2);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = r ? rowB : rowA;
            if (owner && (r == 0 || hasB)) {
                const int c = colsel;
                const size_t off = (size_t)row * Dpad + c0 + c;
                T o;
                if (MODE == SPMM_PLAIN) o = (T)(ascale * (double)acc[r][0]);
                else if (MODE == SPMM_LANCZOS) {
                    const T u = us[(size_t)row * CS + c];
                    o = (T)(ascale * (double)acc[r][0] - shift * (double)u);
                    dot += (double)u * (double)o;
                    dot2 += (double)o * (double)o;
                } else if (MODE == SPMM_AXPBY) o = (T)(ascale * (double)acc[r][0] + shift * (double)F[off] + inv_k * (double)X2[off]);
                else {
                    const T u = us[(size_t)row * CS + c];
                    o = (T)((ascale * (double)acc[r][0] - shift * (double)u) * inv_k);
                    F[off] = F[off] + o;
                }
                Out[off] = o;
            }
        }
    }
    if (MODE == SPMM_LANCZOS) {
        if (owner) { shd[0][wv][colsel] = dot; shd[1][wv][colsel] = dot2; }
        __syncthreads();
        if ((int)threadIdx.x < CS) {
            double s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < NWV; ++w) { s1 += shd[0][w][threadIdx.x]; s2 += shd[1][w][threadIdx.x]; }
            partial[(size_t)range * Dpad + c0 + threadIdx.x] = s1;
            if (shifted) partial_o2[(size_t)range * Dpad + c0 + threadIdx.x] = s2;
        }
        if (range == 0)  // the slabs no range owns
            for (int i = threadIdx.x; i < (nslabs - nranges) * CS; i += SLICE_THREADS) {
                const int sl = nranges + i / CS, cc2 = i % CS;
                partial[(size_t)sl * Dpad + c0 + cc2] = 0.0;
                if (shifted) partial_o2[(size_t)sl * Dpad + c0 + cc2] = 0.0;
            }
    }
}

// ---- LDS-staged SpMM over locality blocks (blocking.h) ------------------------------------------
// One workgroup (8 waves) per (row block, 256-byte column tile):
//   phase 1  gathers the block's union of dense rows into LDS (each row segment one coalesced 256-B read,
//            all of a thread's gathers in flight at once) and copies the block's (local index, value)
//            entries next to it;
//   phase 2  serves every nonzero of the row block from LDS: a 16-lane group owns one nonzero, so one
//            ds_read_b128 wave-instruction reads four full 256-B rows, each service group touching all 64
//            banks once (conflict-free by construction).
// Same fused epilogues as k_spmm.  Values arrive in blocked, chunk-padded order (val_blk).
struct BlkDev {
    int nb;
    const int* rowptr;        // [nb+1] positions into `order`
    const int* order;         // position -> original row
    const int* un_ptr;        // [nb+1]
    const int* un_cols;       // original column ids
    const int* bptr;          // [K+1] blocked entry ranges by position (multiples of 16)
    const unsigned short* lidx;
    const unsigned short* self_li;  // [K] by position: local index of the row itself (its diagonal entry)
    const int* desc;          // [nb][8] {q0, rows, m0, entries, un0, union size, chunks, 0}
    const int* un_fixed;      // [nb][BLK_UNION_ROWS] union column ids at a fixed stride
    int half_tile;            // every block fits the half-tile kernel (k_spmm_blk2)
};
constexpr int BLK_THREADS = MMW_BLK_THREADS;
constexpr int BLK_WAVES = BLK_THREADS / WAVE;
constexpr int BLK_TILE_BYTES = 256;
constexpr int BLK_UNION_ROWS = MMW_BLK_UNION;
constexpr int BLK_META_LDS = MMW_BLK_META;   // staged (offset, value) entries
constexpr int BLK_UNOFF_LDS = BLK_UNION_ROWS * 4;  // element offsets of the union rows (kept out of the register file)
constexpr int BLK_ROWINFO_LDS = 1024;  // 64 rows x {first entry, chunks, output row, own staged row}
template <typename T> struct alignas(2 * sizeof(T)) BlkMeta {  // one staged entry (read with one ds_read_b64 / b128)
    unsigned int li;
    T v;
};
// the 4 consecutive staged entries of one lane group, as wide LDS reads
__device__ __forceinline__ void load_meta4(const BlkMeta<float>* p, unsigned (&li)[4], float (&v)[4]) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    li[0] = a.x; v[0] = __uint_as_float(a.y); li[1] = a.z; v[1] = __uint_as_float(a.w);
    li[2] = b.x; v[2] = __uint_as_float(b.y); li[3] = b.z; v[3] = __uint_as_float(b.w);
}
__device__ __forceinline__ void load_meta4(const BlkMeta<double>* p, unsigned (&li)[4], double (&v)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint4 a = reinterpret_cast<const uint4*>(p)[u];
        li[u] = a.x;
        v[u] = __longlong_as_double(((long long)a.w << 32) | (long long)a.z);
    }
}
template <typename T> constexpr int blk_max_entries() { return BLK_META_LDS / (int)sizeof(BlkMeta<T>); }

template <typename T, int MODE>
__global__ __launch_bounds__(BLK_THREADS) void k_spmm_blk(BlkDev B, int Dpad, int ntiles, int tpw, const T* __restrict__ val_blk,
                                                          const T* __restrict__ U, T* __restrict__ Out, T* __restrict__ F,
                                                          const T* __restrict__ X2, double ascale, double shift, double inv_k,
                                                          double* __restrict__ partial, const ExpmPlan* __restrict__ plan, int step,
                                                          unsigned long long* __restrict__ stamps) {
    constexpr int VEC = V16<T>::N;
    constexpr int CT = BLK_TILE_BYTES / (int)sizeof(T);  // columns per tile
#define MMW_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    MMW_STAMP(0);
    if (plan) {
        if (step > plan->m) return;
        if (MODE == SPMM_TAYLOR) shift = plan->mu / plan->nsub;
    }
    constexpr int RPP = BLK_THREADS / 16;                // union rows gathered per pass
    constexpr int NG = (BLK_UNION_ROWS + RPP - 1) / RPP;             // gathers per thread
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);                                                        // [BLK_UNION_ROWS][CT]
    BlkMeta<T>* meta = reinterpret_cast<BlkMeta<T>*>(smem_raw + (size_t)BLK_UNION_ROWS * BLK_TILE_BYTES);  // [entries]
    int4* rowinfo = reinterpret_cast<int4*>(smem_raw + (size_t)BLK_UNION_ROWS * BLK_TILE_BYTES + BLK_META_LDS);  // [64]
    unsigned* unoff = reinterpret_cast<unsigned*>(smem_raw + (size_t)BLK_UNION_ROWS * BLK_TILE_BYTES + BLK_META_LDS + BLK_ROWINFO_LDS);  // [BLK_UNION_ROWS]
    double* shdot = reinterpret_cast<double*>(smem_raw + (size_t)BLK_UNION_ROWS * BLK_TILE_BYTES + BLK_META_LDS + BLK_ROWINFO_LDS + BLK_UNOFF_LDS);  // [BLK_WAVES][CT]
    // XCD-aware id: consecutive ids of one XCD walk consecutive row blocks of one tile group
    const int ngroups = (ntiles + tpw - 1) / tpw;
    const int total = B.nb * ngroups;
    const int per = (total + 7) / 8;
    const int id = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (id >= total) return;
    const int tg = id / B.nb, rb = id - tg * B.nb;
    const int t0 = tg * tpw, t1 = min(ntiles, t0 + tpw);
    const int l16 = threadIdx.x & 15;
    // everything about the block from one record at a fixed address (no chain of dependent index loads)
    const int* dsc = B.desc + (size_t)rb * 8;
    const int q0 = dsc[0], q1 = q0 + dsc[1];
    const int m0 = dsc[2], nmeta = dsc[3];
    const int nun = dsc[5];
    const int u0 = threadIdx.x >> 4;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int g = lane >> 4;  // lane group = one nonzero per LDS read
    // this thread's share of the union (same rows for every tile)
    unsigned gbase[NG];  // element offsets of the union rows (K * Dpad < 2^32); live only until they are parked in LDS
#pragma unroll
    for (int j = 0; j < NG; ++j) gbase[j] = (unsigned)B.un_fixed[(size_t)rb * BLK_UNION_ROWS + min(u0 + j * RPP, BLK_UNION_ROWS - 1)] * (unsigned)Dpad;
    T x[NG][VEC];
    auto gather0 = [&](int t) {  // first tile: offsets still in registers
        const int c = t * CT + l16 * VEC;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[j][v] = T(0);
            if (u0 + j * RPP < nun && c < Dpad) load16(U + (size_t)gbase[j] + c, x[j]);
        }
    };
    auto gather = [&](int t) {  // later tiles: offsets come back from LDS
        const int c = t * CT + l16 * VEC;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[j][v] = T(0);
            if (u0 + j * RPP < nun && c < Dpad) load16(U + (size_t)unoff[u0 + j * RPP] + c, x[j]);
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int j = 0; j < NG; ++j)
            if (u0 + j * RPP < max(nun, 2)) store16(tile + (size_t)(u0 + j * RPP) * CT + l16 * VEC, x[j]);  // rows 0/1 back the padding entries
    };
    MMW_STAMP(1);
    // the block's entries, once for all its tiles: requested before the gathers so they land first
    constexpr int NM = (BLK_META_LDS / (int)sizeof(BlkMeta<T>) + BLK_THREADS - 1) / BLK_THREADS;
    unsigned mli[NM];
    T mv[NM];
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        const int i = threadIdx.x + k * BLK_THREADS;
        mli[k] = 0;
        mv[k] = T(0);
        if (i < nmeta) {
            mli[k] = (unsigned)B.lidx[m0 + i] * BLK_TILE_BYTES;  // byte offset of the staged row
            mv[k] = val_blk[m0 + i];
        }
    }
    int4 ri = make_int4(0, 0, 0, 0);  // per-row bookkeeping, fetched once per workgroup (not per tile)
    if ((int)threadIdx.x < q1 - q0) {
        const int q = q0 + threadIdx.x;
        const int b0 = B.bptr[q];
        ri = make_int4(b0 - m0, (B.bptr[q + 1] - b0) >> 4, B.order[q], (int)B.self_li[q] * BLK_TILE_BYTES);
    }
    gather0(t0);
    if (l16 == 0)
#pragma unroll
        for (int j = 0; j < NG; ++j) unoff[u0 + j * RPP] = gbase[j];
    if ((int)threadIdx.x < q1 - q0) rowinfo[threadIdx.x] = ri;
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        const int i = threadIdx.x + k * BLK_THREADS;
        if (i < nmeta) {
            BlkMeta<T> e;
            e.li = mli[k];
            e.v = mv[k];
            meta[i] = e;
        }
    }
    MMW_STAMP(2);
    deposit();
    MMW_STAMP(3);
    __syncthreads();
    MMW_STAMP(4);
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) gather(t + 1);  // next tile's rows fly while this tile is consumed from LDS
        const int col0 = t * CT;
        const bool colok = col0 + l16 * VEC < Dpad;
        if (MODE == SPMM_LANCZOS && g == 0) {  // alpha numerators of this wave's rows accumulate in its own LDS row
#pragma unroll
            for (int v = 0; v < VEC; ++v) shdot[wib * CT + l16 * VEC + v] = 0.0;
        }
        for (int q = q0 + wib; q < q1; q += BLK_WAVES) {
            T acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = T(0);
            // wave-uniform row extent -> scalar loop control, no divergence in the hot loop
            int beg, nch, out_row, self_off;  // wave-uniform: scalar registers, scalar loop control
            {
                const int4 rinfo = rowinfo[q - q0];
                beg = __builtin_amdgcn_readfirstlane(rinfo.x);
                nch = __builtin_amdgcn_readfirstlane(rinfo.y);
                out_row = __builtin_amdgcn_readfirstlane(rinfo.z);
                self_off = __builtin_amdgcn_readfirstlane(rinfo.w);
            }
            const BlkMeta<T>* mp = meta + beg + 4 * g;
            const char* tbase = reinterpret_cast<const char*>(tile) + l16 * 16;
            // two register sets (A/B) alternate: the entries of chunk k+1 are requested BEFORE the staged rows of
            // chunk k, so the single wait that covers the rows also covers them and the next step can issue at once
            unsigned liA[4], liB[4];
            T vA[4], vB[4];
            if (nch > 0) load_meta4(mp, liA, vA);
            int c = 0;
            for (; c + 1 < nch; c += 2) {
                T xx[4][VEC];
                load_meta4(mp + 16 * (c + 1), liB, vB);
#pragma unroll
                for (int u = 0; u < 4; ++u) load16(reinterpret_cast<const T*>(tbase + liA[u]), xx[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] += vA[u] * xx[u][v];
                if (c + 2 < nch) load_meta4(mp + 16 * (c + 2), liA, vA);
#pragma unroll
                for (int u = 0; u < 4; ++u) load16(reinterpret_cast<const T*>(tbase + liB[u]), xx[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] += vB[u] * xx[u][v];
            }
            if (c < nch) {
                T xx[4][VEC];
#pragma unroll
                for (int u = 0; u < 4; ++u) load16(reinterpret_cast<const T*>(tbase + liA[u]), xx[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] += vA[u] * xx[u][v];
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[v] = stride_sum(acc[v], 16);
            }
            if (g == 0 && colok) {
                const int row = out_row;
                const size_t off = (size_t)row * Dpad + col0 + l16 * VEC;
                T o[VEC];
                if (MODE == SPMM_PLAIN) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = (T)(ascale * (double)acc[v]);
                } else if (MODE == SPMM_LANCZOS) {
                    T u[VEC];
                    load16(reinterpret_cast<const T*>(tbase + self_off), u);  // U[row] is in the staged union (diagonal entry)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        o[v] = (T)(ascale * (double)acc[v]);
                        shdot[wib * CT + l16 * VEC + v] += (double)u[v] * (double)o[v];
                    }
                } else if (MODE == SPMM_AXPBY) {
                    T f[VEC], x2[VEC];
                    load16(F + off, f);
                    load16(X2 + off, x2);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = (T)(ascale * (double)acc[v] + shift * (double)f[v] + inv_k * (double)x2[v]);
                } else {
                    T u[VEC], f[VEC];
                    load16(reinterpret_cast<const T*>(tbase + self_off), u);
                    load16(F + off, f);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        o[v] = (T)((ascale * (double)acc[v] - shift * (double)u[v]) * inv_k);
                        f[v] += o[v];
                    }
                    store16(F + off, f);
                }
                store16(Out + off, o);
            }
        }
        if (t == t0) MMW_STAMP(5);
        __syncthreads();  // every wave is done with this tile (and shdot is complete)
        if (t == t0) MMW_STAMP(6);
        if (MODE == SPMM_LANCZOS) {
            for (int c = threadIdx.x; c < CT; c += BLK_THREADS)
                if (col0 + c < Dpad) {
                    double s = 0.0;
                    for (int w = 0; w < BLK_WAVES; ++w) s += shdot[w * CT + c];
                    partial[(size_t)rb * Dpad + col0 + c] = s;
                }
        }
        if (t + 1 < t1) {
            deposit();
            if (t == t0) MMW_STAMP(7);
            __syncthreads();
            if (t == t0) MMW_STAMP(8);
        }
    }
    MMW_STAMP(9);
#undef MMW_STAMP
}

// ---- the same product on 128-byte half tiles, two workgroups per CU ------------------------------
// k_spmm_blk keeps one 1024-thread workgroup per CU, so every barrier, the prologue's dependent loads and the
// gather of the next tile stall the whole CU.  Here a workgroup is 8 waves with at most 78 KiB of LDS: two are
// resident per CU and one computes while the other waits.  A staged row is 128 B; an 8-lane group owns one
// nonzero, so one ds_read_b128 wave-instruction serves 8 nonzeros.  The b128 service groups pair lane groups
// 0|3, 1|2, 4|7, 5|6 on the same 16 banks of a bank half; the host orders every 16-entry chunk so that the paired
// groups read rows of opposite parity (blocking.h), which makes every read conflict-free.
// LDS: [rowinfo 64 x int4][shdot 8 x 32 x f64][shdot2 8 x 32 x f64][rows (nun rounded up to 8) x 128 B][entries + 2 chunks].
constexpr int B2_THREADS = 512;
constexpr int B2_WAVES = B2_THREADS / WAVE;
constexpr int B2_ROW_BYTES = 128;
constexpr int B2_LDS_BYTES = 79872;     // == BLK2_LDS_BYTES (blocking.h)
constexpr int B2_HEADER_BYTES = 5120;   // == BLK2_HEADER_BYTES: 1024 + 2048 + 2048
__device__ __forceinline__ void load_meta2(const BlkMeta<float>* p, unsigned (&li)[2], float (&v)[2]) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0];
    li[0] = a.x; v[0] = __uint_as_float(a.y); li[1] = a.z; v[1] = __uint_as_float(a.w);
}
__device__ __forceinline__ void load_meta2(const BlkMeta<double>* p, unsigned (&li)[2], double (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint4 a = reinterpret_cast<const uint4*>(p)[u];
        li[u] = a.x;
        v[u] = __longlong_as_double(((long long)a.w << 32) | (long long)a.z);
    }
}

// acc += a * x over one 16-byte lane slice.  float: two v_pk_fma_f32 with the scalar broadcast by op_sel (the plain
// loop is SLP-packed into pk_mul + pk_add + register shuffles, and this kernel is VALU-issue bound)
template <typename T> struct Axpy16;
template <> struct Axpy16<float> {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
    __device__ __forceinline__ void add(float a, const float (&x)[4]) {
        const f2 a2 = {a, a};
        lo = __builtin_elementwise_fma((f2){x[0], x[1]}, a2, lo);
        hi = __builtin_elementwise_fma((f2){x[2], x[3]}, a2, hi);
    }
    __device__ __forceinline__ void get(float (&o)[4]) const { o[0] = lo.x; o[1] = lo.y; o[2] = hi.x; o[3] = hi.y; }
};
template <> struct Axpy16<double> {
    double s0 = 0.0, s1 = 0.0;
    __device__ __forceinline__ void add(double a, const double (&x)[2]) {
        s0 = __builtin_fma(a, x[0], s0);
        s1 = __builtin_fma(a, x[1], s1);
    }
    __device__ __forceinline__ void get(double (&o)[2]) const { o[0] = s0; o[1] = s1; }
};

// the four entries held by the lanes of a quad, broadcast to every lane of the quad (DPP quad_perm, no LDS traffic)
template <int Q> __device__ __forceinline__ unsigned quad_lane(unsigned w) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)w, Q * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ void quad_bcast(const BlkMeta<float>& e, unsigned (&li)[4], float (&v)[4]) {
    const unsigned w = __float_as_uint(e.v);
    li[0] = quad_lane<0>(e.li); li[1] = quad_lane<1>(e.li); li[2] = quad_lane<2>(e.li); li[3] = quad_lane<3>(e.li);
    v[0] = __uint_as_float(quad_lane<0>(w)); v[1] = __uint_as_float(quad_lane<1>(w));
    v[2] = __uint_as_float(quad_lane<2>(w)); v[3] = __uint_as_float(quad_lane<3>(w));
}
__device__ __forceinline__ void quad_bcast(const BlkMeta<double>& e, unsigned (&li)[4], double (&v)[4]) {
    const unsigned long long w = (unsigned long long)__double_as_longlong(e.v);
    const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
    li[0] = quad_lane<0>(e.li); li[1] = quad_lane<1>(e.li); li[2] = quad_lane<2>(e.li); li[3] = quad_lane<3>(e.li);
    v[0] = __longlong_as_double((long long)(((unsigned long long)quad_lane<0>(hi) << 32) | quad_lane<0>(lo)));
    v[1] = __longlong_as_double((long long)(((unsigned long long)quad_lane<1>(hi) << 32) | quad_lane<1>(lo)));
    v[2] = __longlong_as_double((long long)(((unsigned long long)quad_lane<2>(hi) << 32) | quad_lane<2>(lo)));
    v[3] = __longlong_as_double((long long)(((unsigned long long)quad_lane<3>(hi) << 32) | quad_lane<3>(lo)));
}

struct Blk2Sched {
    int nfull;        // row blocks [0, nfull) run whole (all tiles), one workgroup each
    int grid1;        // nfull rounded up to 8 (the XCD interleave)
    int tpw_tail;     // tiles per workgroup for the remaining blocks
    int groups_tail;  // ceil(ntiles / tpw_tail)
};
template <typename T, int MODE>
__global__ __launch_bounds__(B2_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_spmm_blk2(BlkDev B, int Dpad, int ntiles, Blk2Sched sched, const T* __restrict__ val_blk, const T* __restrict__ U, T* __restrict__ Out,
                 T* __restrict__ F, const T* __restrict__ X2, double ascale, double shift, double inv_k, double* __restrict__ partial,
                 double* __restrict__ partial_o2, const ExpmPlan* __restrict__ plan, int step, unsigned long long* __restrict__ stamps) {
    constexpr int VEC = V16<T>::N;
    constexpr int CT = B2_ROW_BYTES / (int)sizeof(T);  // columns per tile
#define MMW_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    MMW_STAMP(0);
    if (stamps && threadIdx.x == 0) {
        stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg(63492);
        stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_getreg(63508);
    }
    bool shifted = false;  // Lanczos on A - mu I with column sums of squares of the product (a-posteriori stop, k_lz_scalars)
    if (plan) {
        if (MODE == SPMM_LANCZOS) {
            if (step > plan_steps(plan, step - 1)) return;
            shifted = plan->apost != 0 && partial_o2 != nullptr;
            if (shifted) shift = plan->mu;
        } else if (step > plan->m) return;
        if (MODE == SPMM_TAYLOR) shift = plan->mu / plan->nsub;
    }
    constexpr int RPP = B2_THREADS / 8;        // union rows gathered per pass
    constexpr int NG = (BLK_UNION_ROWS + RPP - 1) / RPP;   // gathers per thread
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int4* rowinfo = reinterpret_cast<int4*>(smem_raw);                          // [64]
    double* shdot = reinterpret_cast<double*>(smem_raw + 1024);                // [B2_WAVES][CT]
    double* shdot2 = reinterpret_cast<double*>(smem_raw + 1024 + 2048);        // [B2_WAVES][CT]
    char* tile = smem_raw + B2_HEADER_BYTES;                                   // [max(nun,2)][128 B]
    // Two-phase static schedule (Blk2Sched): whole row blocks first, one workgroup each; the blocks that would start a
    // mostly empty last round are cut into tile groups so that their pieces fill the chip.  Ids are XCD-aware within a phase.
    int rb, t0, t1;
    if ((int)blockIdx.x < sched.grid1) {
        const int id = (blockIdx.x & 7) * (sched.grid1 >> 3) + (blockIdx.x >> 3);
        if (id >= sched.nfull) return;
        rb = id; t0 = 0; t1 = ntiles;
    } else {
        const int b2 = blockIdx.x - sched.grid1, rem = B.nb - sched.nfull;
        const int total = rem * sched.groups_tail;
        const int id = (b2 & 7) * ((total + 7) >> 3) + (b2 >> 3);
        if (id >= total) return;
        const int tg = id / rem;
        rb = sched.nfull + id - tg * rem;
        t0 = tg * sched.tpw_tail; t1 = min(ntiles, t0 + sched.tpw_tail);
        if (t0 >= t1) return;
    }
    const int* dsc = B.desc + (size_t)rb * 8;
    const int q0 = dsc[0], nrows = dsc[1];
    const int m0 = dsc[2], nmeta = dsc[3];
    const int nun = dsc[5];
    BlkMeta<T>* meta = reinterpret_cast<BlkMeta<T>*>(tile + (size_t)((nun + 7) & ~7) * B2_ROW_BYTES);
    const int l8 = threadIdx.x & 7, u0 = threadIdx.x >> 3;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int g = lane >> 3;  // lane group = one nonzero per LDS read
    // Byte offsets of this thread's union rows, kept in registers for all tiles (K * Dpad * sizeof(T) < 4 GiB, checked by
    // the host).  The union is staged in whole groups of 8 rows (un_fixed pads with the first column), so a wave's 8 rows
    // are gathered or skipped together: the row test is scalar, and only the last, partial column tile tests lanes.
    const int nun8 = (nun + 7) & ~7;
    unsigned gbase[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j)
        gbase[j] = (unsigned)B.un_fixed[(size_t)rb * BLK_UNION_ROWS + min(u0 + j * RPP, BLK_UNION_ROWS - 1)] * (unsigned)(Dpad * (int)sizeof(T)) + (unsigned)(l8 * 16);
    const int wrow = (threadIdx.x >> 6) * 8;  // first union row of this wave in pass 0
    const char* Ub = reinterpret_cast<const char*>(U);
    T x[NG][VEC];
    auto gather = [&](int t) {
        const unsigned cb = (unsigned)(t * B2_ROW_BYTES);
        const bool edge = (t + 1) * CT > Dpad;
        if (!edge) {
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if (wrow + j * RPP < nun8) load16(reinterpret_cast<const T*>(Ub + (gbase[j] + cb)), x[j]);
        } else {
            const bool ok = t * CT + l8 * VEC < Dpad;
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if (wrow + j * RPP < nun8 && ok) load16(reinterpret_cast<const T*>(Ub + (gbase[j] + cb)), x[j]);
        }
    };
    auto deposit = [&]() {  // lanes past the last column deposit stale registers; those columns are never stored
#pragma unroll
        for (int j = 0; j < NG; ++j)
            if (wrow + j * RPP < nun8) store16(reinterpret_cast<T*>(tile + (size_t)(u0 + j * RPP) * B2_ROW_BYTES) + l8 * VEC, x[j]);
    };
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) x[j][v] = T(0);
    MMW_STAMP(1);
    gather(t0);
    // the block's entries, once for all its tiles
    for (int i = threadIdx.x; i < nmeta; i += B2_THREADS) {
        BlkMeta<T> e;
        e.li = (unsigned)B.lidx[m0 + i] * B2_ROW_BYTES;  // byte offset of the staged row
        e.v = val_blk[m0 + i];
        meta[i] = e;
    }
    if ((int)threadIdx.x < nrows) {
        const int q = q0 + threadIdx.x;
        const int b0 = B.bptr[q];
        rowinfo[threadIdx.x] = make_int4(b0 - m0, (B.bptr[q + 1] - b0) >> 4, B.order[q], (int)B.self_li[q] * B2_ROW_BYTES);
    }
    MMW_STAMP(2);
    deposit();
    MMW_STAMP(3);
    __syncthreads();
    MMW_STAMP(4);
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) gather(t + 1);  // next tile's rows fly while this tile is consumed from LDS
        const int col0 = t * CT;
        const bool colok = col0 + l8 * VEC < Dpad;
        T dotw[VEC], dotw2[VEC];  // alpha numerators / squares of the product, this wave's rows (lanes of group 0)
#pragma unroll
        for (int v = 0; v < VEC; ++v) dotw[v] = dotw2[v] = T(0);
        for (int r = wib; r < nrows; r += B2_WAVES) {
            T acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = T(0);
            int beg, nch, out_row, self_off;  // wave-uniform: scalar registers, scalar loop control
            {
                const int4 rinfo = rowinfo[r];
                beg = __builtin_amdgcn_readfirstlane(rinfo.x);
                nch = __builtin_amdgcn_readfirstlane(rinfo.y);
                out_row = __builtin_amdgcn_readfirstlane(rinfo.z);
                self_off = __builtin_amdgcn_readfirstlane(rinfo.w);
            }
            // The kernel is VALU-issue bound (a wave64 instruction holds its SIMD for 4 cycles; LDS runs at a third of its
            // rate), so the loop is written for the fewest vector instructions per product: a lane group reads its two
            // entries of a chunk with ONE broadcast ds_read_b128 (no cross-lane moves), then per entry one address add,
            // one ds_read_b128 of the staged row and two v_pk_fma_f32.  Two chunks per step; register sets A / B alternate,
            // each requested one step ahead (reads run up to two chunks past the row: the next row, or the block's slack).
            const BlkMeta<T>* mp = meta + beg + 2 * g;
            const char* tbase = tile + l8 * 16;
            const int npair = nch >> 1;
            Axpy16<T> sum;
            unsigned liA[4], liB[4];
            T vA[4], vB[4];
            auto fetch = [&](int pair, unsigned (&li)[4], T (&vv)[4]) {
                unsigned l0[2], l1[2];
                T v0[2], v1[2];
                load_meta2(mp + pair * 32, l0, v0);
                load_meta2(mp + pair * 32 + 16, l1, v1);
                li[0] = l0[0]; li[1] = l0[1]; li[2] = l1[0]; li[3] = l1[1];
                vv[0] = v0[0]; vv[1] = v0[1]; vv[2] = v1[0]; vv[3] = v1[1];
            };
            auto consume4 = [&](const unsigned (&li)[4], const T (&vv)[4]) {
                T xx[4][VEC];
#pragma unroll
                for (int u = 0; u < 4; ++u) load16(reinterpret_cast<const T*>(tbase + li[u]), xx[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) sum.add(vv[u], xx[u]);
            };
            fetch(0, liA, vA);
            int p = 0;
            for (; p + 2 <= npair; p += 2) {
                fetch(p + 1, liB, vB);
                consume4(liA, vA);
                fetch(p + 2, liA, vA);
                consume4(liB, vB);
            }
            if (p < npair) {
                fetch(p + 1, liB, vB);
                consume4(liA, vA);
#pragma unroll
                for (int u = 0; u < 2; ++u) { liA[u] = liB[u]; vA[u] = vB[u]; }
            }
            if (nch & 1) {  // odd tail: the first chunk of the pending pair
                T xx[2][VEC];
#pragma unroll
                for (int u = 0; u < 2; ++u) load16(reinterpret_cast<const T*>(tbase + liA[u]), xx[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u) sum.add(vA[u], xx[u]);
            }
            sum.get(acc);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[v] = stride_sum(acc[v], 8);
            }
            if (g == 0 && colok) {
                const size_t off = (size_t)out_row * Dpad + col0 + l8 * VEC;
                T o[VEC];
                if (MODE == SPMM_PLAIN) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = (T)ascale * acc[v];
                } else if (MODE == SPMM_LANCZOS) {
                    T u[VEC];
                    load16(reinterpret_cast<const T*>(tbase + self_off), u);  // U[row] is in the staged union (diagonal entry)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        o[v] = (T)ascale * acc[v] - (T)shift * u[v];  // shift is 0 unless the recurrence runs on A - mu I
                        dotw[v] += u[v] * o[v];  // this wave's 2-3 rows of the tile; widened once per tile below
                        dotw2[v] += o[v] * o[v];
                    }
                } else if (MODE == SPMM_AXPBY) {
                    T f[VEC], x2[VEC];
                    load16(F + off, f);
                    load16(X2 + off, x2);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = (T)(ascale * (double)acc[v] + shift * (double)f[v] + inv_k * (double)x2[v]);
                } else {
                    T u[VEC], f[VEC];
                    load16(reinterpret_cast<const T*>(tbase + self_off), u);
                    load16(F + off, f);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        o[v] = (T)((ascale * (double)acc[v] - shift * (double)u[v]) * inv_k);
                        f[v] += o[v];
                    }
                    store16(F + off, f);
                }
                store16(Out + off, o);
            }
        }
        if (MODE == SPMM_LANCZOS && g == 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                shdot[wib * CT + l8 * VEC + v] = (double)dotw[v];
                shdot2[wib * CT + l8 * VEC + v] = (double)dotw2[v];
            }
        }
        if (t == t0) MMW_STAMP(5);
        __syncthreads();  // every wave is done with this tile (and shdot is complete)
        if (t == t0) MMW_STAMP(6);
        if (MODE == SPMM_LANCZOS) {
            for (int c = threadIdx.x; c < CT; c += B2_THREADS)
                if (col0 + c < Dpad) {
                    double s = 0.0, s2 = 0.0;
                    for (int w = 0; w < B2_WAVES; ++w) {
                        s += shdot[w * CT + c];
                        s2 += shdot2[w * CT + c];
                    }
                    partial[(size_t)rb * Dpad + col0 + c] = s;
                    if (shifted) partial_o2[(size_t)rb * Dpad + col0 + c] = s2;
                }
        }
        if (t + 1 < t1) {
            deposit();
            if (t == t0) MMW_STAMP(7);
            __syncthreads();
            if (t == t0) MMW_STAMP(8);
        }
    }
    MMW_STAMP(9);
#undef MMW_STAMP
}

// Column sums of squares of a block: partial[block][col] = sum over the block's rows of X^2.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_colsq(int K, int Dpad, const T* __restrict__ X, double* __restrict__ partial) {
    // thread t owns column (t % Dpad) for rows t / Dpad + n * rows_per_pass
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sh = reinterpret_cast<double*>(smem_raw);  // [BLOCK]
    const int rpp = BLOCK / Dpad > 0 ? BLOCK / Dpad : 1;
    for (int c0 = 0; c0 < Dpad; c0 += BLOCK) {
        const int c = c0 + (Dpad >= BLOCK ? threadIdx.x : threadIdx.x % Dpad);
        const int r0 = Dpad >= BLOCK ? 0 : threadIdx.x / Dpad;
        double s = 0.0;
        if (c < Dpad && r0 < rpp)
            for (int row = blockIdx.x * rpp + r0; row < K; row += gridDim.x * rpp) {
                const double x = (double)X[(size_t)row * Dpad + c];
                s += x * x;
            }
        sh[threadIdx.x] = s;
        __syncthreads();
        if (Dpad >= BLOCK) {
            if (c < Dpad) partial[(size_t)blockIdx.x * Dpad + c] = s;
        } else if (threadIdx.x < Dpad) {
            double t = 0.0;
            for (int r = 0; r < rpp; ++r) t += sh[r * Dpad + threadIdx.x];
            partial[(size_t)blockIdx.x * Dpad + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

// out[col] = sum_b partial[b][col] in a fixed order: one workgroup per 16 columns (128-byte row segments), 64 row
// slices per column.  The Lanczos scalar update that consumes the sum is fused in (OP).
enum { LZ_NONE = 0, LZ_INIT = 1, LZ_ALPHA = 2, LZ_BETA = 3 };
// Lanczos scalars, per column c (all arrays [MAX_ORDER+2][Dpad], index j = Lanczos step starting at 1):
//   beta[0] = ||b_c||, sinv[j] = 1/beta[j-1] (0 when the Krylov space is exhausted), alpha[j].
struct LanczosScalars {
    double* alpha;
    double* beta;
    double* sinv;
    double* coef;  // [MAX_ORDER+1][Dpad]: y = sum_j coef[j] U_j
};

template <int OP>
__global__ __launch_bounds__(1024) void k_colreduce(int nb, int Dpad, const double* __restrict__ partial, double* __restrict__ out, int j,
                                                    double eps, LanczosScalars S, const ExpmPlan* __restrict__ plan) {
    __shared__ double sh[64][17];
    if (plan && ((OP == LZ_ALPHA && j > plan->m) || (OP == LZ_BETA && j >= plan->m))) return;
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (c < Dpad)
        for (int b = sl; b < nb; b += 64) s += partial[(size_t)b * Dpad + c];
    sh[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < Dpad) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < 64; ++p) t += sh[p][cl];
        if (OP == LZ_NONE) {
            out[c] = t;
        } else if (OP == LZ_INIT) {  // beta0 = ||b_c||, sinv1
            const double b = sqrt(t);
            S.beta[c] = b;
            S.sinv[1 * Dpad + c] = b > 0.0 ? 1.0 / b : 0.0;
        } else if (OP == LZ_ALPHA) {  // alpha_j = sinv_j^2 (U_j . A U_j)
            const double si = S.sinv[j * Dpad + c];
            S.alpha[j * Dpad + c] = si * si * t;
        } else {  // beta_j = ||U_{j+1}||, sinv_{j+1}; a column at rounding level has exhausted its Krylov space
            const double b = sqrt(t);
            const double scale = fabs(S.alpha[j * Dpad + c]) + (j > 1 ? S.beta[(j - 1) * Dpad + c] : 0.0);
            const bool dead = S.sinv[j * Dpad + c] == 0.0 || !(b > eps * scale) || !(b > 0.0);
            S.beta[j * Dpad + c] = dead ? 0.0 : b;
            S.sinv[(j + 1) * Dpad + c] = dead ? 0.0 : 1.0 / b;
        }
    }
}
// U_{j+1} = sinv_j * t - alpha_j sinv_j U_j - beta_{j-1} sinv_{j-1} U_{j-1};  partial col sums of U_{j+1}^2
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_lz_update(int K, int Dpad, int j, const T* __restrict__ Tm, const T* __restrict__ Uj,
                                                     const T* __restrict__ Ujm1, T* __restrict__ Unext, LanczosScalars S,
                                                     double* __restrict__ partial, const ExpmPlan* __restrict__ plan,
                                                     unsigned short* __restrict__ planes = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (plan && j >= plan_steps(plan, j)) return;  // U_{m+1} is never formed: the last product feeds the combination directly
    double* sh = reinterpret_cast<double*>(smem_raw);  // [BLOCK]
    const int rpp = BLOCK / Dpad > 0 ? BLOCK / Dpad : 1;
    for (int c0 = 0; c0 < Dpad; c0 += BLOCK) {
        const int c = c0 + (Dpad >= BLOCK ? threadIdx.x : threadIdx.x % Dpad);
        const int r0 = Dpad >= BLOCK ? 0 : threadIdx.x / Dpad;
        double s = 0.0;
        if (c < Dpad && r0 < rpp) {
            const double sj = S.sinv[j * Dpad + c];
            const double c2 = S.alpha[j * Dpad + c] * sj;
            const double c3 = j > 1 ? S.beta[(j - 1) * Dpad + c] * S.sinv[(j - 1) * Dpad + c] : 0.0;
            for (int row = blockIdx.x * rpp + r0; row < K; row += gridDim.x * rpp) {
                const size_t o = (size_t)row * Dpad + c;
                double w = sj * (double)Tm[o] - c2 * (double)Uj[o];
                if (j > 1) w -= c3 * (double)Ujm1[o];
                const T wt = (T)w;
                Unext[o] = wt;
                if (planes) {  // fp32 handles: the next product reads U_{j+1} as two bf16 halves (kernels_mfma.h)
                    const float wf = (float)wt;
                    const unsigned short hi = __builtin_bit_cast(unsigned short, (__bf16)wf);
                    const unsigned short lo = __builtin_bit_cast(unsigned short, (__bf16)(wf - __uint_as_float((unsigned)hi << 16)));
                    planes[o] = hi;
                    planes[(size_t)K * Dpad + o] = lo;
                }
                s += (double)wt * (double)wt;
            }
        }
        sh[threadIdx.x] = s;
        __syncthreads();
        if (Dpad >= BLOCK) {
            if (c < Dpad) partial[(size_t)blockIdx.x * Dpad + c] = s;
        } else if (threadIdx.x < Dpad) {
            double t = 0.0;
            for (int r = 0; r < rpp; ++r) t += sh[r * Dpad + threadIdx.x];
            partial[(size_t)blockIdx.x * Dpad + threadIdx.x] = t;
        }
        __syncthreads();
    }
}
// per column: the corrected Krylov approximation (Saad 1992) of exp(A/nsub) b from m Lanczos steps:
//   y = beta0 * [ V_m exp(T) e_1 + (1/nsub) phi * w ],   T = T_m/nsub,  phi = e_m^T phi_1(T) e_1,
//   w = A v_m - alpha_m v_m - beta_{m-1} v_{m-1}  (= beta_m v_{m+1}; never formed, never normalised),
// [exp(T) e_1; phi] being the first column of exp of the augmented matrix [[T, 0], [e_m^T, 0]].
// It spends the product A v_m that the m-th SpMM made anyway, which buys one polynomial degree: m SpMMs
// reach the accuracy of a degree-m Taylor polynomial.  Output, so that y = sum_{j<=m} coef[j] U_j + coef[m+1] * t_m
// with t_m = A U_m (the last SpMM's output, U_j unnormalised):
//   coef[j] = beta0 g_j sinv_j  minus the w-terms folded onto U_m and U_{m-1};  coef[m+1] = beta0 phi sinv_m / nsub.
// exp(M) e_1 by a scaled Taylor series on the small vector (repeated application keeps it valid for any norm).
// With `est` the matrix gets one more row, [0 .. 0 1 0]: the extra component is e_m^T phi_2(T) e_1, the leading coefficient of
// the corrected scheme's error  beta0 beta_m sum_{k>=2} (e_m^T phi_k(T) e_1) A^{k-1} v_{m+1}  (Saad 1992, Thm 5.1); returned.
// `scale` multiplies the result (e^{mu} when the recurrence ran on the shifted operator).
template <int NMAX>  // NMAX >= m + 2; all loops fully unrolled so the small vectors live in registers
__device__ __forceinline__ double texp_core(int Dpad, int c, int m, double inv_nsub, LanczosScalars S, double scale, bool est) {
    double a[NMAX], b[NMAX], g[NMAX], t[NMAX], f[NMAX];
    double nrm = 0.0;
    int mm = m;  // Krylov dimension actually reached
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        a[j] = j < m ? S.alpha[(j + 1) * Dpad + c] * inv_nsub : 0.0;
        b[j] = j + 1 < m ? S.beta[(j + 1) * Dpad + c] * inv_nsub : 0.0;  // b[j] couples j and j+1
    }
#pragma unroll
    for (int j = NMAX - 1; j >= 0; --j)
        if (j + 1 < m && b[j] == 0.0) mm = j + 1;  // breakdown: the leading block is exact, no correction
    const bool corrected = mm == m && S.sinv[m * Dpad + c] != 0.0;
    const int n = corrected ? (est ? m + 2 : m + 1) : mm;
#pragma unroll
    for (int j = 0; j < NMAX; ++j)
        if (j < mm) {
            const double r = fabs(a[j]) + (j > 0 ? fabs(b[j > 0 ? j - 1 : 0]) : 0.0) + (j + 1 < mm ? fabs(b[j]) : 0.0);
            nrm = r > nrm ? r : nrm;
        }
    if (corrected && nrm < 1.0) nrm = 1.0;  // the augmented row e_m^T
    int sq = 1;
    while (nrm / sq > 0.5) sq *= 2;
    const double isq = 1.0 / sq;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) g[j] = j == 0 ? 1.0 : 0.0;
    for (int rep = 0; rep < sq; ++rep) {  // g <- exp(M/sq) g
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            t[j] = g[j];
            f[j] = g[j];
        }
        for (int k = 1; k <= 30; ++k) {
            double tn[NMAX];
            double big = 0.0;
            const double isq_k = isq / (double)k;  // ONE division per term (a double-precision division per element was most of this launch's time)
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                double v = 0.0;
                if (j < mm) {
                    v = a[j] * t[j];
                    if (j > 0) v += b[j > 0 ? j - 1 : 0] * t[j > 0 ? j - 1 : 0];
                    if (j + 1 < mm && j + 1 < NMAX) v += b[j] * t[j + 1 < NMAX ? j + 1 : j];
                } else if (j < n) {
                    v = t[j > 0 ? j - 1 : 0];  // the augmented rows e_m^T and e_{m+1}^T
                }
                tn[j] = v * isq_k;
                big = fabs(tn[j]) > big ? fabs(tn[j]) : big;
            }
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                t[j] = tn[j];
                f[j] += tn[j];
            }
            if (big < 1e-19) break;
        }
#pragma unroll
        for (int j = 0; j < NMAX; ++j) g[j] = f[j];
    }
    const double e = S.beta[c] * scale;
    double gm = 0.0, gm2 = 0.0;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        if (j < m) S.coef[(j + 1) * Dpad + c] = j < mm ? e * g[j] * S.sinv[(j + 1) * Dpad + c] : 0.0;
        if (j == m) gm = g[j];
        if (j == m + 1) gm2 = g[j];
    }
    double ct = 0.0;
    if (corrected) {
        const double sm = S.sinv[m * Dpad + c];
        const double w = e * gm * inv_nsub;  // weight of w = sm*t_m - alpha_m sm U_m - beta_{m-1} s_{m-1} U_{m-1}
        ct = w * sm;
        S.coef[m * Dpad + c] -= w * S.alpha[m * Dpad + c] * sm;
        if (m > 1) S.coef[(m - 1) * Dpad + c] -= w * S.beta[(m - 1) * Dpad + c] * S.sinv[(m - 1) * Dpad + c];
    }
    S.coef[(m + 1) * Dpad + c] = ct;
    return corrected ? fabs(gm2) : 0.0;
}
// m == 1 in closed form (the common case once the steps stop on the estimate): T is the scalar alpha, so
// exp(T) e_1 = e^alpha, e_1^T phi_1 e_1 = (e^alpha - 1)/alpha, e_1^T phi_2 e_1 = (e^alpha - 1 - alpha)/alpha^2.  Same outputs as texp_core.
__device__ __forceinline__ double texp_order1(int Dpad, int c, double inv_nsub, LanczosScalars S, double scale) {
    const double si = S.sinv[1 * Dpad + c];
    const double a = S.alpha[1 * Dpad + c] * inv_nsub;
    const double e = S.beta[c] * scale;
    const double g = exp(a);
    double phi1, phi2;
    if (fabs(a) < 1e-2) {  // series: the closed forms cancel
        phi1 = 1.0 + a * (0.5 + a * (1.0 / 6.0 + a * (1.0 / 24.0 + a * (1.0 / 120.0 + a * (1.0 / 720.0)))));
        phi2 = 0.5 + a * (1.0 / 6.0 + a * (1.0 / 24.0 + a * (1.0 / 120.0 + a * (1.0 / 720.0 + a * (1.0 / 5040.0)))));
    } else {
        const double em1 = expm1(a);
        phi1 = em1 / a;
        phi2 = (em1 - a) / (a * a);
    }
    if (si == 0.0) {  // empty column
        S.coef[1 * Dpad + c] = 0.0;
        S.coef[2 * Dpad + c] = 0.0;
        return 0.0;
    }
    const double w = e * phi1 * inv_nsub;  // weight of w = s1*t1 - alpha1 s1 U1
    S.coef[1 * Dpad + c] = e * g * si - w * S.alpha[1 * Dpad + c] * si;
    S.coef[2 * Dpad + c] = w * si;
    return fabs(phi2);
}

// All Lanczos scalars of step j in ONE launch (fixed summation order, like k_colreduce):
//   partB  holds the column sums of squares of U_j (j == 1: of the start block)  -> beta_{j-1}, sinv_j
//   partA  holds the alpha numerators U_j . A U_j of the product just made        -> alpha_j
// and, on the last step (j == m), the small exponentials (texp_core) of the columns this workgroup owns.
constexpr int LZS_COLS = 4;  // columns per workgroup: 32-byte slab segments, 256 slab slices in flight per column
// NMAX bounds the small exponentials this instantiation can run (steps j <= NMAX - 2): the host picks 4, 8 or MAX_ORDER + 2 by
// the step number, so the common one- and two-step launches keep their five small vectors in registers (one kernel for every
// order kept 18-entry arrays with dynamic bounds in scratch memory: 387 scratch instructions, 11 us per launch).
template <int NMAX>
__global__ __launch_bounds__(1024) void k_lz_scalars(int nbA, const double* __restrict__ partA, const double* __restrict__ partO2, int nbB,
                                                     const double* __restrict__ partB, int Dpad, int j, int m, double inv_nsub, double eps,
                                                     LanczosScalars S, ExpmPlan* __restrict__ plan) {
    constexpr int SL = 1024 / LZS_COLS;
    __shared__ double shA[SL][LZS_COLS + 1], shB[SL][LZS_COLS + 1], shO[SL][LZS_COLS + 1];
    __shared__ double s2A[16][LZS_COLS + 1], s2B[16][LZS_COLS + 1], s2O[16][LZS_COLS + 1];
    bool apost = false;
    double scale = 1.0;
    if (plan) {
        inv_nsub = 1.0 / plan->nsub;
        if (j > plan_steps(plan, j - 1)) return;  // beyond the a-priori order, or an earlier step already met the tolerance
        apost = plan->apost != 0 && partO2 != nullptr;
        if (apost) scale = exp(plan->mu * inv_nsub);  // the recurrence ran on A - mu I
    }
    const int cl = threadIdx.x % LZS_COLS, sl = threadIdx.x / LZS_COLS;
    const int c = blockIdx.x * LZS_COLS + cl;
    double sa = 0.0, sb = 0.0, so = 0.0;
    if (c < Dpad) {
        for (int b = sl; b < nbB; b += SL) sb += partB[(size_t)b * Dpad + c];
        for (int b = sl; b < nbA; b += SL) sa += partA[(size_t)b * Dpad + c];
        if (apost)
            for (int b = sl; b < nbA; b += SL) so += partO2[(size_t)b * Dpad + c];
    }
    shA[sl][cl] = sa;
    shB[sl][cl] = sb;
    shO[sl][cl] = so;
    __syncthreads();
    if (sl < 16) {  // fixed two-level order: 16 runs of SL/16 slices, then the 16 run totals
        double ta = 0.0, tb = 0.0, to = 0.0;
        for (int p = 0; p < SL / 16; ++p) {
            ta += shA[sl * (SL / 16) + p][cl];
            tb += shB[sl * (SL / 16) + p][cl];
            to += shO[sl * (SL / 16) + p][cl];
        }
        s2A[sl][cl] = ta;
        s2B[sl][cl] = tb;
        s2O[sl][cl] = to;
    }
    __syncthreads();
    if (sl != 0 || c >= Dpad) return;
    double ta = 0.0, tb = 0.0, to = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        ta += s2A[p][cl];
        tb += s2B[p][cl];
        to += s2O[p][cl];
    }
    double si, bprev = 0.0;  // bprev = beta_{j-1}
    if (j == 1) {  // beta0 = ||b_c||, sinv_1
        const double b = sqrt(tb);
        si = b > 0.0 ? 1.0 / b : 0.0;
        S.beta[c] = b;
        S.sinv[1 * Dpad + c] = si;
    } else {  // beta_{j-1} = ||U_j||, sinv_j; a column at rounding level has exhausted its Krylov space
        const int i = j - 1;
        const double b = sqrt(tb);
        const double sc = fabs(S.alpha[i * Dpad + c]) + (i > 1 ? S.beta[(i - 1) * Dpad + c] : 0.0);
        const bool dead = S.sinv[i * Dpad + c] == 0.0 || !(b > eps * sc) || !(b > 0.0);
        si = dead ? 0.0 : 1.0 / b;
        bprev = dead ? 0.0 : b;
        S.beta[i * Dpad + c] = bprev;
        S.sinv[j * Dpad + c] = si;
    }
    const double alpha = si * si * ta;  // alpha_j = sinv_j^2 (U_j . A U_j)
    S.alpha[j * Dpad + c] = alpha;
    const int mlast = plan ? plan->m : m;
    if (!apost) {
        if (j == mlast) {
            __threadfence_block();
            texp_core<NMAX>(Dpad, c, j, inv_nsub, S, scale, false);
        }
        return;
    }
    // A-posteriori stop.  The combination coefficients for "stop after this step" are made at every step, together with
    //   est = e^rho * beta_j * |e_j^T phi_2(T_j) e_1| * rho / (1 - rho)  >=  ||error|| / ||exp(A)b||   for this column,
    // the leading term of Saad's expansion of the corrected scheme's error with ||A - mu I|| replaced by its 1-norm bound rho
    // and the remaining terms by a geometric tail.  beta_j = ||A v_j - alpha_j v_j - beta_{j-1} v_{j-1}|| comes from the
    // column sums of squares of the product the SpMM just made: ||A v_j||^2 = alpha_j^2 + beta_{j-1}^2 + beta_j^2.
    __threadfence_block();
    double phi2;
    if (j == 1) phi2 = texp_order1(Dpad, c, inv_nsub, S, scale);
    else phi2 = texp_core<NMAX>(Dpad, c, j, inv_nsub, S, scale, true);
    const double n2 = si * si * to;
    if (j == 1) {  // what the first-order form would have cost on this column (the host decides from it whether the next chunk takes that form)
        float ff = (float)first_order_bound(sqrt(n2 > 0.0 ? n2 : 0.0), plan->rho);
        if (!(ff >= 0.0f)) ff = __uint_as_float(0x7f800000u);
        atomicMax(&plan->first_est, __float_as_uint(ff));
    }
    double bj2 = n2 - alpha * alpha - bprev * bprev;
    const double floor2 = 4.0 * eps * n2;  // cancellation floor of the difference (the products were rounded to T)
    if (!(bj2 > floor2)) bj2 = floor2;
    const double rho = plan->rho;
    const double est = si == 0.0 ? 0.0 : exp(rho) * sqrt(bj2) * phi2 * rho / (1.0 - rho);
    float ef = (float)est;
    if (!(ef >= 0.0f)) ef = __uint_as_float(0x7f800000u);  // NaN -> +inf: never stops early
    if ((double)ef < est) ef = __uint_as_float(__float_as_uint(ef) + 1u);  // round up
    atomicMax(&plan->conv[j], __float_as_uint(ef));  // a maximum: the result does not depend on the order of arrival
}
// y[row,:] = sum_{j=1..m} coef[j][:] * U_j[row,:] + coef[m+1][:] * Tm[row,:]   (U_j = Ubase + (j-1)*stride),
// one wavefront per row; optionally also d[row] = ||y_row||^2 and per-block partial sums of d (the trace).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_lz_combine(int K, int Dpad, int m, const T* __restrict__ Ubase, size_t stride,
                                                      const T* __restrict__ Tm, const double* __restrict__ coef, T* __restrict__ Yout,
                                                      ExpmPlan* __restrict__ plan, T* __restrict__ d,
                                                      double* __restrict__ dpart, int* __restrict__ viol,
                                                      unsigned short* __restrict__ planes = nullptr, int planes_only = 0) {
    // planes_only: the only reader of this y is the matrix-core SDDMM (its two bf16 planes); the fp32 copy is not stored
    constexpr int VEC = V16<T>::N;
    __shared__ double sh[WAVES_PER_BLOCK];
    if (plan) {
        m = plan_steps(plan, plan->m);  // the last step that ran made the coefficients for stopping there
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            plan->m_eff = m;
            // the host launched fewer steps than the a-priori order and the estimate did not accept them either: replay
            if (plan->apost && plan->m_apriori > plan->m && m == plan->m && plan->conv[m] > __float_as_uint((float)plan->tol) && viol) atomicOr(viol, VIOL_ORDER);
        }
    }
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int ngroups = Dpad / VEC;
    double tot = 0.0;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < K; row += gridDim.x * WAVES_PER_BLOCK) {
        double ss = 0.0;
        for (int p = lane; p < ngroups; p += WAVE) {
            const size_t o = (size_t)row * Dpad + (size_t)p * VEC;
            T x[VEC];
            double s[VEC];
            load16(Tm + o, x);
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[v] = coef[(m + 1) * Dpad + p * VEC + v] * (double)x[v];
            for (int j = 1; j <= m; ++j) {
                load16(Ubase + (size_t)(j - 1) * stride + o, x);
#pragma unroll
                for (int v = 0; v < VEC; ++v) s[v] += coef[j * Dpad + p * VEC + v] * (double)x[v];
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                x[v] = (T)s[v];
                ss += (double)x[v] * (double)x[v];
            }
            if (!planes_only) store16(Yout + o, x);
            if constexpr (sizeof(T) == 4) {
                if (planes) {  // the matrix-core SDDMM reads y as two bf16 halves (kernels_mfma.h)
                    unsigned w4[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float xf = x[v];
                        const unsigned short hi = __builtin_bit_cast(unsigned short, (__bf16)xf);
                        const unsigned short lo = __builtin_bit_cast(unsigned short, (__bf16)(xf - __uint_as_float((unsigned)hi << 16)));
                        w4[v] = ((unsigned)hi << 16) | lo;
                    }
                    // interleaved per 32 columns: 32 hi halves, then the 32 lo halves (the layout k_sddmm_mfma gathers whole lines of)
                    const int col = p * VEC;
                    unsigned short* g = planes + (size_t)row * Dpad * 2 + (size_t)(col >> 5) * 64 + (col & 31);
                    *reinterpret_cast<uint2*>(g) = make_uint2((w4[0] >> 16) | (w4[1] & 0xFFFF0000u), (w4[2] >> 16) | (w4[3] & 0xFFFF0000u));
                    *reinterpret_cast<uint2*>(g + 32) = make_uint2((w4[0] & 0xFFFFu) | (w4[1] << 16), (w4[2] & 0xFFFFu) | (w4[3] << 16));
                }
            }
        }
        if (d) {
            ss = wave_sum(ss);
            if (lane == 0) {
                d[row] = (T)ss;
                tot += (double)(T)ss;
            }
        }
    }
    if (d) {
        tot = block_sum(tot, sh);
        if (threadIdx.x == 0) dpart[blockIdx.x] = tot;
    }
}
// y *= scale (Taylor: e^{mu})
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_scale(size_t n, T* __restrict__ y, double scale, const ExpmPlan* __restrict__ plan) {
    if (plan) scale = exp(plan->mu / plan->nsub);
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK)
        y[o] = (T)((double)y[o] * scale);
}
template <typename T> __global__ __launch_bounds__(BLOCK) void k_copy(size_t n, const T* __restrict__ a, T* __restrict__ b) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) b[o] = a[o];
}
// float64 host block [K, D] -> device block [K, Dpad] of T (zero padded)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_import_block(int K, int D, int Dpad, const double* __restrict__ src, T* __restrict__ dst) {
    const size_t n = (size_t)K * Dpad;
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) {
        const int c = (int)(o % Dpad);
        const size_t r = o / Dpad;
        dst[o] = c < D ? (T)src[r * D + c] : T(0);
    }
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_export_block(int K, int D, int Dpad, const T* __restrict__ src, double* __restrict__ dst) {
    const size_t n = (size_t)K * D;
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) {
        const int c = (int)(o % D);
        const size_t r = o / D;
        dst[o] = (double)src[r * Dpad + c];
    }
}

// ---- planning: 1-norm bound of A - mu I from the values on a symmetric pattern ----------------
// partial[block] = max over the block's rows of sum_e |ascale*val[e] - mu*[e is diag]|
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_rowabs(int K, const int* __restrict__ indptr, const int* __restrict__ col,
                                                  const T* __restrict__ val, double ascale, const double* __restrict__ trace_part,
                                                  int ntrace, double* __restrict__ partial) {
    __shared__ double sh[WAVES_PER_BLOCK];
    // mu from the per-block diagonal sums of the producing kernel (every block re-reduces the same slab)
    double tr = 0.0;
    for (int i = threadIdx.x; i < ntrace; i += BLOCK) tr += trace_part[i];
    tr = block_sum(tr, sh);
    const double mu = ascale * tr / K;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    double best = 0.0;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < K; row += gridDim.x * WAVES_PER_BLOCK) {
        double s = 0.0;
        for (int e = indptr[row] + lane; e < indptr[row + 1]; e += WAVE) {
            double v = ascale * (double)val[e];
            if (col[e] == row) v -= mu;
            s += fabs(v);
        }
        s = wave_sum(s);
        best = s > best ? s : best;
    }
    best = block_max(best, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = best;
}
// Per workgroup (4 rows, one wave each): {sum d_i, max(d_i + o_i), max(o_i - d_i)} with d_i = ascale*val[i,i] and
// o_i = sum_{j != i} |ascale*val[i,j]|.  Since |x| = max(x, -x), the 1-norm bound of A - mu I is
// max_i(|d_i - mu| + o_i) = max(max_i(d_i + o_i) - mu, max_i(o_i - d_i) + mu): k_plan gets mu = tr/K and the bound from
// these three slabs in one short pass, and no kernel has to re-reduce a slab to know mu first.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_rowsums(int K, const int* __restrict__ indptr, const int* __restrict__ col,
                                                   const T* __restrict__ val, double ascale, double* __restrict__ part /* [3][grid] */) {
    __shared__ double sh[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    double sd = 0.0, pp = -1e300, pm = -1e300;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < K; row += gridDim.x * WAVES_PER_BLOCK) {
        double s = 0.0, d = 0.0;
        for (int e = indptr[row] + lane; e < indptr[row + 1]; e += WAVE) {
            const double v = ascale * (double)val[e];
            if (col[e] == row) d = v;
            else s += fabs(v);
        }
        s = wave_sum(s);
        d = wave_sum(d);  // exactly one lane holds the diagonal
        if (lane == 0) {
            sd += d;
            pp = d + s > pp ? d + s : pp;
            pm = s - d > pm ? s - d : pm;
        }
    }
    sd = block_sum(sd, sh);
    pp = block_max(pp, sh);
    pm = block_max(pm, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = sd;
        part[gridDim.x + blockIdx.x] = pp;
        part[2 * gridDim.x + blockIdx.x] = pm;
    }
}

__host__ __device__ inline int plan_order(int method, double rho, double tol, int max_order) {
    // smallest m with a remainder bound below tol.  Corrected Lanczos with m steps reproduces polynomials of
    // degree m: the error is at most twice the best degree-m polynomial error on [-rho, rho], bounded by the
    // Chebyshev-shifted Taylor remainder 2 (rho/2)^(m+1) / (m+1)! e^{rho};  Taylor degree m: rho^{m+1}/(m+1)! e^{rho}.
    double term = 1.0;
    const double er = exp(2.0 * rho);  // also covers the e^{-rho} lower bound on ||exp(A')b|| / ||b||
    for (int m = 1; m <= max_order; ++m) {
        if (method == 0) {
            term *= (rho * 0.5) / m;
            if (4.0 * term * (rho * 0.5) / (m + 1) * er <= tol) return m;
        } else {
            term *= rho / m;
            if (term * rho / (m + 1) * er <= tol) return m;
        }
    }
    return -1;
}

// m_launch > 0: the host has already decided to launch m_launch steps with a single substep (no readback);
// if the matrix needs more, the sticky flag *viol is raised and the caller replays the batch synchronously.
// iter_seen: iteration index of the matrix the sums in `part` were taken from; lagged != 0: that matrix is one iteration older
// than the one about to be multiplied (see ExpmPlan).
constexpr int PLAN_THREADS = 256;
__device__ __forceinline__ void plan_body(int K, int method, int max_order, double tol, const double* __restrict__ part, int np,
                                          ExpmPlan* __restrict__ plan, int m_launch, int* __restrict__ viol, int apost, int iter_seen, int lagged,
                                          double* sh /* one entry per wave of the workgroup */) {
    double tr = 0.0, pp = -1e300, pm = -1e300;
    for (int i = threadIdx.x; i < np; i += (int)blockDim.x) {
        tr += part[i];
        pp = part[np + i] > pp ? part[np + i] : pp;
        pm = part[2 * np + i] > pm ? part[2 * np + i] : pm;
    }
    tr = block_sum(tr, sh);
    pp = block_max(pp, sh);
    pm = block_max(pm, sh);
    double mu = tr / K;
    if (threadIdx.x == 0) {
        ExpmPlan p;
        const ExpmPlan old = *plan;
        // the previous plan's extrapolated bounds against the sums of the matrix it was used on (seen only now)
        if (old.lagged && iter_seen == old.h_iter + 1) {
            const double need = pp - old.mu > pm + old.mu ? pp - old.mu : pm + old.mu;
            const double absn_seen = pp > pm ? pp : pm;
            // ... and the row-sum bound every gate and certificate of that product used (mfma_ok, f16_ok, f16a_ok, first_verify)
            if (need > old.rho || absn_seen > old.absn) atomicOr(viol, VIOL_LAGGED);
        }
        // history and growth per iteration
        p.h_iter = old.h_iter; p.h_pp = old.h_pp; p.h_pm = old.h_pm; p.h_mu = old.h_mu;
        p.g_pp = old.g_pp; p.g_pm = old.g_pm; p.g_mu = old.g_mu;
        if (iter_seen > old.h_iter) {
            const double dt = (double)(iter_seen - old.h_iter);
            p.g_pp = (pp - old.h_pp) / dt; p.g_pm = (pm - old.h_pm) / dt; p.g_mu = (mu - old.h_mu) / dt;
            p.h_iter = iter_seen; p.h_pp = pp; p.h_pm = pm; p.h_mu = mu;
        }
        p.lagged = lagged;
        if (lagged) {
            // Bounds for the next matrix: one more step of growth, with a factor 2.  The matrix is a running sum of bounded
            // increments, so its sums grow by about 1 / (iterations so far) of themselves per step: that average rate backs the
            // last finite difference (which is noisy: the maxima move between rows).
            const double avg = 1.0 / (double)(iter_seen + 2);
            const double spread = (fabs(pp) > fabs(pm) ? fabs(pp) : fabs(pm)) + fabs(mu);
            const double gp = p.g_pp > avg * spread ? p.g_pp : avg * spread, gm = p.g_pm > avg * spread ? p.g_pm : avg * spread;
            const double mu1 = mu + p.g_mu;
            pp += 2.0 * gp + 1e-3 * spread;
            pm += 2.0 * gm + 1e-3 * spread;
            mu = mu1;
        }
        const double r = pp - mu > pm + mu ? pp - mu : pm + mu;
        p.rho = r;
        p.mu = mu;
        p.tol = tol;
        p.overflow = 0;
        p.m_eff = 0;
        int nsub = 1, m = -1;
        for (; nsub <= 4096; nsub *= 2) {
            m = plan_order(method, r / nsub, tol / nsub, max_order);
            if (m > 0) break;
        }
        if (m <= 0) {
            m = max_order;
            nsub = 4096;
            p.overflow = 1;
        }
        p.m = m;
        p.m_apriori = m;
        p.nsub = nsub;
        // matrix-core SpMM: ||dT||_F <= 3 * 2^-17 || |A| ||_2 ||U||_F and || |A| ||_2 <= max_i sum_j |a_ij| = max(pp, pm)
        p.absn = pp > pm ? pp : pm;
        p.mfma_ok = 2.3e-5 * p.absn <= tol ? 1 : 0;
        p.f16_ok = F16_PLANE_EXPECT * p.absn <= tol && p.absn < 0.03 ? 1 : 0;  // ... and the entries times 2^20 stay inside fp16's range
        p.f16a_ok = p.f16_ok && (F16_PLANE_EXPECT + F16_UNIT) * p.absn <= tol ? 1 : 0;
        // the a-posteriori stop needs the shifted recurrence of the half-tile SpMM, a single substep and a geometric tail
        p.apost = apost && method == 0 && nsub == 1 && !p.overflow && r < 0.5;
        for (int i = 0; i < MAX_ORDER + 2; ++i) p.conv[i] = 0u;  // identity of the maximum; a step's entry is read only after its k_lz_scalars ran
        p.first_est = 0u;
        if (m_launch > 0 && (nsub > 1 || m > m_launch || p.overflow)) {
            // more than the host launched: with the estimate on, the combination decides whether the launched steps were
            // enough after all; without it the batch is replayed
            if (!(p.apost && nsub == 1 && !p.overflow)) atomicOr(viol, VIOL_PLAN);
            p.m = m < m_launch ? m : m_launch;  // keep the launched kernels in range
            p.nsub = 1;
        }
        *plan = p;
    }
}
__global__ __launch_bounds__(PLAN_THREADS) void k_plan(int K, int method, int max_order, double tol, const double* __restrict__ part, int np,
                                                       ExpmPlan* __restrict__ plan, int m_launch, int* __restrict__ viol, int apost, int iter_seen) {
    __shared__ double sh[PLAN_THREADS / WAVE];
    plan_body(K, method, max_order, tol, part, np, plan, m_launch, viol, apost, iter_seen, 0, sh);
}
// End of a chunk: the last iteration's extrapolated bounds against the sums of the matrix they were used on (part = k_rowsums of
// it), since no later plan of the chunk will see that matrix; also brings the history up to it.
__global__ __launch_bounds__(PLAN_THREADS) void k_plan_verify(int K, const double* __restrict__ part, int np, ExpmPlan* __restrict__ plan,
                                                              int* __restrict__ viol, int iter_seen) {
    __shared__ double sh[PLAN_THREADS / WAVE];
    double tr = 0.0, pp = -1e300, pm = -1e300;
    for (int i = threadIdx.x; i < np; i += PLAN_THREADS) {
        tr += part[i];
        pp = part[np + i] > pp ? part[np + i] : pp;
        pm = part[2 * np + i] > pm ? part[2 * np + i] : pm;
    }
    tr = block_sum(tr, sh);
    pp = block_max(pp, sh);
    pm = block_max(pm, sh);
    if (threadIdx.x == 0) {
        ExpmPlan p = *plan;
        const double mu = tr / K;
        if (p.lagged && iter_seen == p.h_iter + 1) {
            const double need = pp - p.mu > pm + p.mu ? pp - p.mu : pm + p.mu;
            if (need > p.rho || (pp > pm ? pp : pm) > p.absn) atomicOr(viol, VIOL_LAGGED);  // see plan_body
        }
        if (iter_seen > p.h_iter) {
            const double dt = (double)(iter_seen - p.h_iter);
            p.g_pp = (pp - p.h_pp) / dt; p.g_pm = (pm - p.h_pm) / dt; p.g_mu = (mu - p.h_mu) / dt;
            p.h_iter = iter_seen; p.h_pp = pp; p.h_pm = pm; p.h_mu = mu;
        }
        p.lagged = 0;
        *plan = p;
    }
}
// history of a fresh run: the zero matrix at iteration -1 (keep_sums: a warm restart keeps the sums of the matrix it continues from)
__global__ void k_plan_reset(ExpmPlan* __restrict__ plan, int keep_sums) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ExpmPlan p = *plan;
        p.lagged = 0;
        p.h_iter = -1;
        if (!keep_sums) { p.h_pp = p.h_pm = p.h_mu = 0.0; p.g_pp = p.g_pm = p.g_mu = 0.0; }
        *plan = p;
    }
}

}  // namespace mmw
