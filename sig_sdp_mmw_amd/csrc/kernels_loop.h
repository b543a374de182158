// Device kernels for the per-iteration MMW phases on the fixed L/X pattern:
//   DUAL  (sim_src/alg/mmw.py:124-142): violations e(X), e_accu += eta e, Y = softmax(e_accu)
//   LOSS  (mmw.py:144-170):             L_accu -= eta (LD + LF + LH)
//   X on the pattern (mmw.py:182-194):  row norms, trace, SDDMM over the edges
//   averaging (mmw.py:77-78) is folded into the producers of X and Y.
// Reductions are wavefront shuffles + fixed-order per-block slabs (bitwise reproducible, no atomics).
#pragma once
#include <type_traits>
#include "blocking.h"
#include "device_utils.h"
#include "kernels_expm.h"
#include "kernels_mfma.h"

namespace mmw {

template <typename T> struct PatternDev {
    int K, Z, E_asso, C;
    int nnzL;
    const int* indptr;
    const int* col;
    const int* pid;       // association pair id or -1
    const int* mirror;
    const int* diag_pos;
    const int* asso_pos;
    const T* sab;         // S_T'[row,col]
    const T* sba;         // S_T'[col,row]
    const T* h_max;
    const T* S_sum;
    const T* inv_norm_H;  // 1/norm_H
    const T* cH;          // h_max/K - S_sum/(K Z)
    // where X lives.  nullptr / -1: `xval` is CSR-ordered like the pattern.  Otherwise X is in the matrix-core SDDMM's tile order
    // (kernels_mfma.h): entry e sits in slot e2w[e], the diagonal of row k in slot xdiag_base + k, association pair p in slot xasso[p]
    const int* e2w = nullptr;
    const int* xasso = nullptr;
    int xdiag_base = -1;
    __device__ __forceinline__ int xslot(int e) const { return e2w ? e2w[e] : e; }
    __device__ __forceinline__ int xdiag(int k) const { return xdiag_base >= 0 ? xdiag_base + k : diag_pos[k]; }
    __device__ __forceinline__ int xpair(int p) const { return xasso ? xasso[p] : asso_pos[p]; }
};

// The sketch of an iteration is pure VALU work that depends on nothing but (seed, iteration): it rides as extra workgroups in
// the launch of a memory-latency-bound kernel -- the LOSS pass of the same iteration (default), or the SDDMM of the previous
// one (MMW_FUSED_SKETCH=1).  nblocks == 0 disables it.
template <typename T> struct SketchArgs {
    int nblocks, K, D;
    uint64_t seed;
    uint32_t iter;
    T* R;
    double* colsq_part;
    unsigned short* planes;  // optional (fp32 only): bf16 hi plane, then the lo plane K * Dpad elements further (kernels_mfma.h)
    int planes_f16;          // 1: ONE fp16 plane instead (the first-order product's operand)
    double* dusq_part;       // with planes_f16: per-slab column sums of (u - fp16(u))^2, the MEASURED rounding of that plane (first_verify)
};
template <typename T, int NWAVES>
__device__ __forceinline__ void sketch_rows(int K, int D, int Dpad, uint64_t seed, uint32_t iter, T* __restrict__ R,
                                            double* __restrict__ colsq_part, int bid, int nblocks, double* shc,
                                            unsigned short* __restrict__ planes = nullptr, int planes_f16 = 0, double* __restrict__ dusq_part = nullptr);

// ---- DUAL, step 1: per row r[k] = sum of off-diagonal X, eD; per pair eF -------------------------
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dual_rows(PatternDev<T> P, const T* __restrict__ xval, T* __restrict__ rsum,
                                                     T* __restrict__ e_this) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int K = P.K;
    const double invK = 1.0 / (double)K;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < K; row += gridDim.x * WAVES_PER_BLOCK) {
        double s = 0.0;
        const int dp = P.diag_pos[row];
        for (int e = P.indptr[row] + lane; e < P.indptr[row + 1]; e += WAVE)
            if (e != dp) s += (double)xval[P.xslot(e)];
        s = wave_sum(s);
        if (lane == 0) {
            rsum[row] = (T)s;
            e_this[row] = (T)(((double)xval[P.xdiag(row)] - 1.0) / (1.0 - invK));  // mmw.py:127
        }
    }
    const double Zm1 = (double)(P.Z - 1);
    const double den = 1.0 / ((double)K * Zm1) + 0.5;  // mmw.py:131
    for (int p = blockIdx.x * BLOCK + threadIdx.x; p < P.E_asso; p += gridDim.x * BLOCK)
        e_this[K + p] = (T)(((double)xval[P.xpair(p)] + 1.0 / Zm1) / den);
}

// ---- DUAL, step 2: eH = (S_T' r (Z-1)/Z - (h - S_sum/Z)) / norm_H ; e_accu += eta e ; block max ----
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dual_h(PatternDev<T> P, const T* __restrict__ rsum, T* __restrict__ e_this,
                                                  T* __restrict__ e_accu, double eta, double* __restrict__ max_part,
                                                  const T* __restrict__ lval = nullptr, double lscale = 0.0,
                                                  double* __restrict__ lpart = nullptr /* [3][grid], as k_rowsums */,
                                                  const double* __restrict__ mref = nullptr, T* __restrict__ Yun = nullptr,
                                                  T* __restrict__ wun = nullptr, double* __restrict__ sum_part = nullptr /* [4][grid] */,
                                                  const long long* __restrict__ rsfx = nullptr /* [K] */, const T* __restrict__ xval = nullptr,
                                                  FirstVerify V = FirstVerify{}, unsigned long long* __restrict__ stamps = nullptr /* diagnostic runs */) {
    unsigned long long tk0 = 0, tk_ptr = 0, tk_rows = 0, tk_tail = 0, tk_c = 0;
    if (stamps) tk0 = __builtin_amdgcn_s_memtime();
    // V.plan: the last V.nwg workgroups of the launch do not take rows; they certify the first-order exponential of the iteration
    // before (kernels_mfma.h, first_verify) -- short independent work under a latency-bound pass.
    const int G = (int)gridDim.x - (V.plan ? V.nwg : 0);  // workgroups of the pass itself: the slabs' stride
    if ((int)blockIdx.x >= G) {
        first_verify(V, (int)blockIdx.x - G);
        return;
    }
    // With `rsfx` the row sums of X come from the matrix-core SDDMM (k_sddmm_mfma: 2^-40 fixed point) instead of k_dual_rows, and
    // the D / F violations that kernel made are taken here from `xval` (mmw.py:127,131).
    // With `mref` (the maximum of e_accu one iteration ago, left in scal[4]) the pass also does softmax pass A's work with that
    // shift instead of this iteration's maximum: the softmax does not depend on the shift, and e_accu's maximum moves by
    // eta * max e per iteration, so the exponentials stay far from overflow (k_dual_scal checks exactly that).  Yun / wun get
    // exp(e_accu - mref) and the same over norm_H; k_dual_scal and the LOSS pass apply the normalisation.
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int K = P.K, Z = P.Z;
    double best = -1e300;
    const double m0 = mref ? *mref : 0.0;
    double sD = 0.0, sF = 0.0, sH = 0.0, sW = 0.0;
    double sd = 0.0, pp = -1e300, pm = -1e300;  // sums of L for the (lagged) plan of the exponential, when lval is given
    const int baseH = K + P.E_asso;
    // TWO rows per wavefront, interleaved: both rows' entries are requested together, then both rows' gathers, so the pair costs the
    // memory round trips of one row, and K / 8 workgroups do the rows' work (measured: 14.4 -> 13.6 us; the pass is bound by its chain of
    // dependent loads and the double-precision tail, not by the number of rounds).  Each row's own sum keeps its order (one entry per
    // lane and slice of 64).
    for (int rb = (blockIdx.x * WAVES_PER_BLOCK + wib) * 2; rb < K; rb += G * WAVES_PER_BLOCK * 2) {
        const bool hasB = rb + 1 < K;
        const int rowA = rb, rowB = hasB ? rb + 1 : rb;
        const int myrow = (lane & 1) ? rowB : rowA;  // the row whose tail this lane computes (lanes 0 and 1)
        // the rows' own scalars do not depend on the sums: requested up front, they arrive under the entries' loads
        const double r_hmax = (double)P.h_max[myrow], r_ssum = (double)P.S_sum[myrow], r_invn = (double)P.inv_norm_H[myrow];
        const double r_eacc = (double)e_accu[baseH + myrow], r_cH = mref ? (double)P.cH[myrow] : 0.0;
        const int a0 = P.indptr[rowA], a1 = P.indptr[rowA + 1];
        const int b0 = hasB ? a1 : a0, b1 = hasB ? P.indptr[rowB + 1] : a0;  // consecutive rows: B starts where A ends
        const int nmax = max(a1 - a0, b1 - b0);
        if (stamps && nmax >= 0) tk_ptr = __builtin_amdgcn_s_memtime();  // (nmax: the row pointers have arrived)
        // the |L| row sums of an fp32 handle (they feed the exponential's norm bound, which carries a margin of 1e-3) are formed in fp32:
        // conversions, double-precision adds and cross-lane steps on these four sums were a fifth of the pass's vector instructions.
        // The violation sums stay in double: e_accu feeds the softmax, and with a large step size an fp32 sum's rounding (7e-7) was
        // amplified into 3e-3 of X_half within 24 iterations (test_matrix_core_spmm_steps_aside_when_the_norm_outgrows_the_split).
        typedef typename std::conditional<sizeof(T) == 4, float, double>::type A;
        double sA = 0.0, sB = 0.0;
        A laA = 0, ldA = 0, laB = 0, ldB = 0;
        for (int k = lane; k - lane < nmax; k += WAVE) {
            const bool onA = a0 + k < a1, onB = b0 + k < b1;
            const int ea = onA ? a0 + k : a0, eb = onB ? b0 + k : a0;
            const T wa = P.sab[ea], wb = P.sab[eb];
            const int ca = P.col[ea], cb = P.col[eb];
            const T va = lval ? lval[ea] : T(0), vb = lval ? lval[eb] : T(0);
            const bool ga = onA && wa != T(0), gb = onB && wb != T(0);
            const int cca = ga ? ca : rowA, ccb = gb ? cb : rowA;
            const double ra = rsfx ? (double)rsfx[cca] * (1.0 / SDM_FX) : (double)rsum[cca];
            const double rbv = rsfx ? (double)rsfx[ccb] * (1.0 / SDM_FX) : (double)rsum[ccb];
            if (ga) sA += (double)wa * ra;
            if (gb) sB += (double)wb * rbv;
            if (lval) {  // the pass reads these rows anyway: |L| row sums and the diagonal, like k_rowsums
                if (onA) {
                    const A v = (A)lscale * (A)va;
                    if (ca == rowA) ldA = v;
                    else laA += v < 0 ? -v : v;
                }
                if (onB) {
                    const A v = (A)lscale * (A)vb;
                    if (cb == rowB) ldB = v;
                    else laB += v < 0 ? -v : v;
                }
            }
        }
        sA = wave_sum(sA);
        sB = wave_sum(sB);
        if (stamps) tk_rows = __builtin_amdgcn_s_memtime();
        const bool tail = lane == 0 || (lane == 1 && hasB);
        if (lval) {
            laA = wave_sum(laA); ldA = wave_sum(ldA);  // exactly one lane holds a row's diagonal
            laB = wave_sum(laB); ldB = wave_sum(ldB);
            if (tail) {
                const double la = (double)((lane & 1) ? laB : laA), ld = (double)((lane & 1) ? ldB : ldA);
                sd += ld;
                pp = ld + la > pp ? ld + la : pp;
                pm = la - ld > pm ? la - ld : pm;
            }
        }
        if (tail) {
            const double s = (lane & 1) ? sB : sA;
            const double eh = (s * (double)(Z - 1) / (double)Z - (r_hmax - (1.0 / (double)Z) * r_ssum)) * r_invn;  // mmw.py:134
            e_this[baseH + myrow] = (T)eh;
            const T a = (T)(r_eacc + (double)(T)eh * eta);
            e_accu[baseH + myrow] = a;
            best = (double)a > best ? (double)a : best;
            if (mref) {
                const T ex = (T)exp((double)a - m0);
                const double wn = (double)ex * r_invn;
                Yun[baseH + myrow] = ex;
                wun[myrow] = (T)wn;
                sH += (double)ex;
                sW += r_cH * wn;
            }
        }
    }
    if (stamps) tk_tail = __builtin_amdgcn_s_memtime();
    const double invK = 1.0 / (double)K, Zm1 = (double)(Z - 1), denF = 1.0 / ((double)K * Zm1) + 0.5;
    for (int c = blockIdx.x * BLOCK + threadIdx.x; c < baseH; c += G * BLOCK) {
        T et;
        if (rsfx) {
            et = c < K ? (T)(((double)xval[P.xdiag(c)] - 1.0) / (1.0 - invK)) : (T)(((double)xval[P.xpair(c - K)] + 1.0 / Zm1) / denF);
            e_this[c] = et;
        } else et = e_this[c];
        const T a = (T)((double)e_accu[c] + (double)et * eta);
        e_accu[c] = a;
        best = (double)a > best ? (double)a : best;
        if (mref) {
            const T ex = (T)exp((double)a - m0);
            Yun[c] = ex;
            if (c < K) sD += (double)ex;
            else sF += (double)ex;
        }
    }
    if (stamps) tk_c = __builtin_amdgcn_s_memtime();
    // one LDS round for everything the block hands on: {max e_accu | L sums sd, pp, pm | softmax sums sD, sF, sH, sW}
    __shared__ double shr[8][WAVES_PER_BLOCK];
    best = wave_max(best);
    if (lval) { sd = wave_sum(sd); pp = wave_max(pp); pm = wave_max(pm); }
    if (mref) { sD = wave_sum(sD); sF = wave_sum(sF); sH = wave_sum(sH); sW = wave_sum(sW); }
    if (lane == 0) {
        shr[0][wib] = best;
        shr[1][wib] = sd; shr[2][wib] = pp; shr[3][wib] = pm;
        shr[4][wib] = sD; shr[5][wib] = sF; shr[6][wib] = sH; shr[7][wib] = sW;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int q = threadIdx.x;
        const bool is_max = q == 0 || q == 2 || q == 3;
        double t = shr[q][0];
        for (int w = 1; w < WAVES_PER_BLOCK; ++w) t = is_max ? (shr[q][w] > t ? shr[q][w] : t) : t + shr[q][w];
        if (q == 0) max_part[blockIdx.x] = t;
        else if (q < 4) { if (lval) lpart[(q - 1) * G + blockIdx.x] = t; }
        else if (mref) sum_part[(q - 4) * G + blockIdx.x] = t;
    }
    if (stamps && lane == 0) {  // per wave: start, pointers, rows summed, rows' tails done, violation part done, end
        const unsigned long long te = __builtin_amdgcn_s_memtime();
        unsigned long long* o = stamps + ((size_t)blockIdx.x * WAVES_PER_BLOCK + wib) * 8;
        o[0] = tk0; o[1] = tk_ptr; o[2] = tk_rows; o[3] = tk_tail; o[4] = tk_c; o[5] = te;
    }
}

// ---- softmax pass A: Y <- exp(e_accu - max); per block sums {all, D block, F block, sum_H cH*Y/norm_H}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_softmax_a(PatternDev<T> P, const T* __restrict__ e_accu, T* __restrict__ Y,
                                                     const double* __restrict__ max_part, int nmax,
                                                     double* __restrict__ sum_part /* [4][grid] */) {
    __shared__ double sh[WAVES_PER_BLOCK];
    double m = -1e300;
    for (int i = threadIdx.x; i < nmax; i += BLOCK) m = max_part[i] > m ? max_part[i] : m;
    m = block_max(m, sh);
    const int K = P.K, baseH = K + P.E_asso, C = P.C;
    double sD = 0.0, sF = 0.0, sH = 0.0, sW = 0.0;
    for (int c = blockIdx.x * BLOCK + threadIdx.x; c < C; c += gridDim.x * BLOCK) {
        const T ex = (T)exp((double)e_accu[c] - m);
        Y[c] = ex;
        if (c < K) sD += (double)ex;
        else if (c < baseH) sF += (double)ex;
        else {
            sH += (double)ex;
            sW += (double)P.cH[c - baseH] * (double)ex * (double)P.inv_norm_H[c - baseH];
        }
    }
    sD = block_sum(sD, sh);
    sF = block_sum(sF, sh);
    sH = block_sum(sH, sh);
    sW = block_sum(sW, sh);
    if (threadIdx.x == 0) {
        sum_part[0 * gridDim.x + blockIdx.x] = sD;
        sum_part[1 * gridDim.x + blockIdx.x] = sF;
        sum_part[2 * gridDim.x + blockIdx.x] = sH;
        sum_part[3 * gridDim.x + blockIdx.x] = sW;
    }
}
struct PlanArgs {  // arguments of plan_body, for the workgroup that k_softmax_b lends to the planning
    ExpmPlan* plan = nullptr;
    const double* part = nullptr;
    int* viol = nullptr;
    double tol = 0.0;
    int K = 0, method = 0, max_order = 0, np = 0, m_launch = 0, apost = 0, iter_seen = 0;
};
// ---- softmax pass B: Y /= total; yavg += Y; scalars for the loss {sumYD, sumYF, sum cH YH/normH}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_softmax_b(int C, T* __restrict__ Y, T* __restrict__ yavg, int accumulate,
                                                     const double* __restrict__ sum_part, int npart,
                                                     double* __restrict__ scal /* [4] */, int baseH, const T* __restrict__ inv_norm_H,
                                                     T* __restrict__ wH /* [K]: Y_H / norm_H, the weight the LOSS gathers */,
                                                     PlanArgs pa = PlanArgs{}, const double* __restrict__ max_part = nullptr, int nmax = 0) {
    __shared__ double sh[WAVES_PER_BLOCK];
    if (pa.plan && blockIdx.x == gridDim.x - 1) {
        // one extra workgroup makes the exponential's plan from the row sums k_dual_h left two launches ago (lagged planning)
        plan_body(pa.K, pa.method, pa.max_order, pa.tol, pa.part, pa.np, pa.plan, pa.m_launch, pa.viol, pa.apost, pa.iter_seen, 1, sh);
        return;
    }
    double s[4];
    for (int q = 0; q < 4; ++q) {
        double t = 0.0;
        for (int i = threadIdx.x; i < npart; i += BLOCK) t += sum_part[q * npart + i];
        s[q] = block_sum(t, sh);
    }
    const double total = s[0] + s[1] + s[2];
    const int nwork = (int)gridDim.x - (pa.plan ? 1 : 0);  // the last workgroup went planning
    for (int c = blockIdx.x * BLOCK + threadIdx.x; c < C; c += nwork * BLOCK) {
        const T y = (T)((double)Y[c] / total);
        Y[c] = y;
        if (accumulate) yavg[c] += y;
        if (c >= baseH) wH[c - baseH] = (T)((double)y * (double)inv_norm_H[c - baseH]);
    }
    if (blockIdx.x == 0) {
        double m = -1e300;
        for (int i = threadIdx.x; i < nmax; i += BLOCK) m = max_part[i] > m ? max_part[i] : m;
        m = block_max(m, sh);
        if (threadIdx.x == 0) {
            scal[0] = s[0] / total;
            scal[1] = s[1] / total;
            scal[2] = s[3] / total;
            scal[3] = total;
            if (max_part) scal[4] = m;  // the shift the next iteration's fused pass may use (k_dual_h, mref)
        }
    }
}

// ---- the fused pass's scalars: one workgroup folds the per-block sums and maxima k_dual_h (mref form) left into what softmax pass
// B leaves in `scal`, and flags a maximum that ran away from the shift (the exponentials may have overflowed: the chunk is
// replayed on the two-pass kernels).
constexpr int DSCAL_THREADS = 1024;
__global__ __launch_bounds__(DSCAL_THREADS) void k_dual_scal(const double* __restrict__ sum_part, const double* __restrict__ max_part, int npart,
                                                            double* __restrict__ scal /* [5] */, double overflow_gap, int* __restrict__ viol,
                                                            FirstVerify V = FirstVerify{}) {
    // workgroups past the first certify the first-order exponential of the iteration before (kernels_mfma.h, first_verify): independent
    // work beside a one-workgroup launch
    if (blockIdx.x > 0) {
        first_verify(V, (int)blockIdx.x - 1);
        return;
    }
    __shared__ double sh[5][DSCAL_THREADS / WAVE];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, m = -1e300;
    // the launch is one memory round trip and a fold: everything it reads is requested at once (two slabs per thread cover the DUAL
    // pass's grid; the shift of the overflow check used to be read after the fold, a round trip of its own)
    const double m0 = threadIdx.x == 0 ? scal[4] : 0.0;
    {
        const int i0 = threadIdx.x, i1 = threadIdx.x + DSCAL_THREADS;
        const bool h0 = i0 < npart, h1 = i1 < npart;
        const int j0 = h0 ? i0 : 0, j1 = h1 ? i1 : 0;
        const double a0 = sum_part[j0], a1 = sum_part[npart + j0], a2 = sum_part[2 * npart + j0], a3 = sum_part[3 * npart + j0], b0 = max_part[j0];
        const double c0 = sum_part[j1], c1 = sum_part[npart + j1], c2 = sum_part[2 * npart + j1], c3 = sum_part[3 * npart + j1], b1 = max_part[j1];
        if (h0) { s0 = a0; s1 = a1; s2 = a2; s3 = a3; m = b0; }
        if (h1) { s0 += c0; s1 += c1; s2 += c2; s3 += c3; m = b1 > m ? b1 : m; }
    }
    for (int i = threadIdx.x + 2 * DSCAL_THREADS; i < npart; i += DSCAL_THREADS) {
        const double a0 = sum_part[i], a1 = sum_part[npart + i], a2 = sum_part[2 * npart + i], a3 = sum_part[3 * npart + i], b = max_part[i];
        s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        m = b > m ? b : m;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); m = wave_max(m);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; sh[3][w] = s3; sh[4][w] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s0 = s1 = s2 = s3 = 0.0; m = -1e300;
        for (int i = 0; i < DSCAL_THREADS / WAVE; ++i) {
            s0 += sh[0][i]; s1 += sh[1][i]; s2 += sh[2][i]; s3 += sh[3][i];
            m = sh[4][i] > m ? sh[4][i] : m;
        }
        const double total = s0 + s1 + s2;
        scal[0] = s0 / total;
        scal[1] = s1 / total;
        scal[2] = s3 / total;
        scal[3] = total;
        scal[4] = m;
        if (!(m - m0 <= overflow_gap) || !(total > 0.0) || !(total < 1e300)) atomicOr(viol, VIOL_SOFTMAX);
    }
}

// wH = Y_H / norm_H (the LOSS gathers it once per side of a gain edge); the loop gets it from softmax pass B
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_hweights(int K, const T* __restrict__ yH, const T* __restrict__ inv_norm_H, T* __restrict__ wH) {
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < K; k += gridDim.x * BLOCK) wH[k] = (T)((double)yH[k] * (double)inv_norm_H[k]);
}

// ---- LOSS: lval -= eta * (LD + LF + LH) on the pattern ----------------------------------------------
// One thread per stored entry (row ids from `lrow`): every array is read fully coalesced, the only gathers
// are the two dual weights of a gain edge.
template <typename T>
__global__ __launch_bounds__(BLOCK, (sizeof(T) == 4 ? 7 : 4)) void k_loss(PatternDev<T> P, const int* __restrict__ lrow, const T* __restrict__ Y,
                                                const T* __restrict__ wH, const double* __restrict__ scal, T* __restrict__ lval,
                                                double eta, const int* __restrict__ bpos, T* __restrict__ lval_blk,
                                                const T* __restrict__ xval = nullptr, T* __restrict__ xavg = nullptr,
                                                SketchArgs<T> sk = SketchArgs<T>{}, int Dpad = 0, const int* __restrict__ fpos = nullptr,
                                                unsigned* __restrict__ afrag = nullptr, T* __restrict__ Ynorm = nullptr,
                                                T* __restrict__ yavg = nullptr, int accumulate = 0, PlanArgs pa = PlanArgs{},
                                                long long* __restrict__ rs_zero = nullptr /* fixed-point totals the coming product / SDDMM add to */,
                                                int rs_zero_n = 0, int afrag_f16 = 0 /* 1: the image feeds the first-order product (split_f16_scaled) */,
                                                unsigned short* __restrict__ afrag16 = nullptr /* ... as ONE fp16 half, in an image of its own (SPMM_FIRST16) */) {
    // Ynorm != nullptr: Y and wH hold the unnormalised exponentials of the fused DUAL pass (k_dual_h, mref form) and scal[3]
    // their total; this pass divides where it uses them and writes the normalised Y (and its running sum) on the side.
    if ((int)blockIdx.x < sk.nblocks) {  // leading workgroups draw this iteration's sketch (same shape as k_sketch_rng: same bits)
        extern __shared__ __attribute__((aligned(16))) char smem_raw[];
        sketch_rows<T, WAVES_PER_BLOCK>(sk.K, sk.D, Dpad, sk.seed, sk.iter, sk.R, sk.colsq_part, (int)blockIdx.x, sk.nblocks,
                                        reinterpret_cast<double*>(smem_raw), sk.planes, sk.planes_f16, sk.dusq_part);
        return;
    }
    const int lead = sk.nblocks + (pa.plan ? 1 : 0);
    if (pa.plan && (int)blockIdx.x == sk.nblocks) {
        // one more workgroup plans the exponential this matrix update is followed by, from the row sums k_dual_h took (lagged planning)
        __shared__ double shp[WAVES_PER_BLOCK];
        plan_body(pa.K, pa.method, pa.max_order, pa.tol, pa.part, pa.np, pa.plan, pa.m_launch, pa.viol, pa.apost, pa.iter_seen, 1, shp);
        return;
    }
    const int bid = (int)blockIdx.x - lead, nb = (int)gridDim.x - lead;
    const int K = P.K, Z = P.Z, baseF = K;
    const double invK = 1.0 / (double)K, Zm1 = (double)(Z - 1);
    const double cF = 0.5 + 1.0 / ((double)K * Zm1);
    // Every trip's operands are requested together, one trip ahead: the pair id used to be read only after the row / column comparison, a
    // round trip of its own in a pass that is a chain of them (measured: LOSS 24.5 -> 23.1 us per step at the benchmark)
    const int e_first = bid * BLOCK + (int)threadIdx.x;
    const int e_pre = e_first < P.nnzL ? e_first : 0;
    int row_n = lrow[e_pre], c_n = P.col[e_pre], pid_n = P.pid[e_pre];
    T lv_n = lval[e_pre];
    int fp_n = fpos ? fpos[e_pre] : 0, bp_n = bpos ? bpos[e_pre] : 0;  // where the images take the new value: known before it is
    const double sumYD = scal[0], sumYF = scal[1], sumW = scal[2];
    const double total = Ynorm ? scal[3] : 1.0;
    const double dconst = -(sumYD * invK) / (1.0 - invK) + (sumYF / ((double)K * Zm1)) / cF - sumW;
    const double inv_total = 1.0 / total;
    const double gscale = (Zm1 / (double)(2 * Z)) * inv_total;
    if (rs_zero)
        for (int k = bid * BLOCK + threadIdx.x; k < rs_zero_n; k += nb * BLOCK) rs_zero[k] = 0;
    if (Ynorm)
        for (int c = bid * BLOCK + threadIdx.x; c < P.C; c += nb * BLOCK) {
            const T y = (T)((double)Y[c] / total);
            Ynorm[c] = y;
            if (accumulate) yavg[c] += y;
        }
    for (int e = e_first; e < P.nnzL; e += nb * BLOCK) {
        const int row = row_n, c = c_n, pidv = pid_n;
        const T lv = lv_n;
        const int fp = fp_n, bp = bp_n;
        if (e + nb * BLOCK < P.nnzL) {  // the next trip's operands
            const int en = e + nb * BLOCK;
            row_n = lrow[en]; c_n = P.col[en]; pid_n = P.pid[en]; lv_n = lval[en];
            if (fpos) fp_n = fpos[en];
            if (bpos) bp_n = bpos[en];
        }
        double add;
        if (c == row) {
            const double y = Ynorm ? (double)(T)((double)Y[row] / total) : (double)Y[row];
            add = y / (1.0 - invK) + dconst;
        } else if (pidv >= 0) {
            const double y = Ynorm ? (double)(T)((double)Y[baseF + pidv] / total) : (double)Y[baseF + pidv];
            add = (y * 0.5) / cF;
        } else {
            const double w_row = (double)wH[row], w_col = (double)wH[c];  // one gather per side (Y_H / norm_H from the DUAL phase)
            add = ((double)P.sab[e] * w_col + (double)P.sba[e] * w_row) * gscale;  // column-scaled S_T' symmetrised
        }
        const T nv = (T)((double)lv - eta * add);
        lval[e] = nv;
        if (bpos) lval_blk[bp] = nv;  // the same value in the LDS-staged kernel's traversal order
        if (fpos) {  // and in the matrix-core kernel's fragment order: two 16-bit halves, or (the first-order product early in a run) one fp16 half
            if (afrag16) afrag16[mf_pos16(fp)] = f16_rn((float)nv * MF_F16_SCALE);
            else afrag[fp] = afrag_f16 ? split_f16_scaled((float)nv) : split_bf16((float)nv);
        }
        if (xavg) xavg[e] += xval[e];  // the previous iteration's X joins the running sum here (same index space, one pass fewer)
    }
}

// ---- row norms of X_half: d[row] = ||y_row||^2, per-block partial sums of d -----------------------
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_rownorm2(int K, int Dpad, const T* __restrict__ Yb, T* __restrict__ d,
                                                    double* __restrict__ part) {
    __shared__ double sh[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    double tot = 0.0;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < K; row += gridDim.x * WAVES_PER_BLOCK) {
        double s = 0.0;
        for (int c = lane; c < Dpad; c += WAVE) {
            const double x = (double)Yb[(size_t)row * Dpad + c];
            s += x * x;
        }
        s = wave_sum(s);
        if (lane == 0) {
            d[row] = (T)s;
            tot += (double)(T)s;
        }
    }
    tot = block_sum(tot, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// ---- SDDMM on the edges: X[a,b] = <y_a, y_b> / tr for b > a, mirrored; diagonal d/tr; xavg += X ----
template <typename T, int NCH>
__global__ __launch_bounds__(BLOCK) void k_sddmm(PatternDev<T> P, int Dpad, int LPR, int G, const T* __restrict__ Yb,
                                                 const T* __restrict__ d, const double* __restrict__ tr_part, int ntr,
                                                 T* __restrict__ xval, T* __restrict__ xavg, int accumulate) {
    constexpr int VEC = V16<T>::N;
    __shared__ double sh[WAVES_PER_BLOCK];
    double t = 0.0;
    for (int i = threadIdx.x; i < ntr; i += BLOCK) t += tr_part[i];
    t = block_sum(t, sh);
    const double tr = t / (double)P.K;  // mmw.py:184
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int g = (NCH == 1) ? lane / LPR : 0;
    const int lig = (NCH == 1) ? lane - g * LPR : lane;
    const bool active = g < G;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < P.K; row += gridDim.x * WAVES_PER_BLOCK) {
        T ya[NCH][VEC];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) ya[c][v] = T(0);
            if (active && lig + 64 * c < LPR) load16(Yb + (size_t)row * Dpad + (size_t)(lig + 64 * c) * VEC, ya[c]);
        }
        const int dp = P.diag_pos[row];
        const int end = P.indptr[row + 1];
        if (lane == 0) {
            const T xd = (T)((double)d[row] / tr);
            xval[dp] = xd;
        }
        // entries right of the diagonal are the upper-triangular edges of this row; UNR edges per lane group are in
        // flight at once (the loop is latency-bound on the gathers otherwise)
        constexpr int UNR = 4;
        for (int e0 = dp + 1; e0 < end; e0 += G * UNR) {
            int ee[UNR];
            double s[UNR];
            T yb[UNR][NCH][VEC];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                ee[u] = e0 + g + u * G;
                s[u] = 0.0;
                const bool ok = active && ee[u] < end;
                const int b = ok ? P.col[ee[u]] : row;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) yb[u][c][v] = T(0);
                    if (ok && lig + 64 * c < LPR) load16(Yb + (size_t)b * Dpad + (size_t)(lig + 64 * c) * VEC, yb[u][c]);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) s[u] += (double)ya[c][v] * (double)yb[u][c][v];
            if (NCH == 1 && LPR >= 4) {
                // transposed reduction of the 4 partial sums over the LPR lanes of a group: two halving steps leave
                // one edge per lane quarter (5 shuffles instead of 16), then the quarter is summed; the lane quarter
                // (lig / (LPR/4)) of edge u ends up holding its total, so one store serves 4 edges per group
                const int h = LPR >> 1, q4 = LPR >> 2;
                const bool up = (lig & h) != 0;
                T a0 = (T)(up ? s[2] : s[0]), a1 = (T)(up ? s[3] : s[1]);        // kept pair
                const T b0 = (T)(up ? s[0] : s[2]), b1 = (T)(up ? s[1] : s[3]);  // sent pair
                a0 += lane_xor(b0, h);
                a1 += lane_xor(b1, h);
                const bool up2 = (lig & q4) != 0;
                T r = up2 ? a1 : a0;
                const T snd = up2 ? a0 : a1;
                r += lane_xor(snd, q4);
                r = group_sum(r, q4);
                const int u = (up ? 2 : 0) + (up2 ? 1 : 0);  // the edge this lane quarter holds
                const int e = e0 + g + u * G;
                if (active && e < end && (lig & (q4 - 1)) == 0) {
                    const T x = (T)((double)r / tr);
                    const int me = P.mirror[e];
                    xval[e] = x;
                    xval[me] = x;
                }
            } else {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (NCH == 1 && G > 1) s[u] = group_sum(s[u], LPR);  // LPR is a power of two when several groups share a wave
                    else s[u] = wave_sum(s[u]);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (active && ee[u] < end && lig == 0) {
                        const T x = (T)(s[u] / tr);
                        const int me = P.mirror[ee[u]];
                        xval[ee[u]] = x;
                        xval[me] = x;
                    }
            }
        }
    }
}

// ---- LDS-staged SDDMM over locality blocks (blocking.h) -------------------------------------------
// One workgroup per row block.  Per 256-byte column tile the block's union of X_half rows is staged in
// LDS exactly like k_spmm_blk does; every thread owns up to SD_ROUNDS upper-triangular entries (a,b) of
// the block and accumulates <y_a, y_b> over the tile in registers, walking the 16 column chunks in a
// lane-skewed order ((s + lane) mod 16) so the 16 lanes of an LDS service group touch 16 different
// chunks = all 64 banks (conflict-free whatever rows they read).  No cross-lane reduction at all.
// running elementwise product sum of 16-byte slices; float: two v_pk_fma_f32 per slice pair
template <typename T> struct Dot16;
template <> struct Dot16<float> {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
    __device__ __forceinline__ void add(const float (&a)[4], const float (&b)[4]) {
        lo = __builtin_elementwise_fma((f2){a[0], a[1]}, (f2){b[0], b[1]}, lo);
        hi = __builtin_elementwise_fma((f2){a[2], a[3]}, (f2){b[2], b[3]}, hi);
    }
    __device__ __forceinline__ float total() const { return (lo.x + lo.y) + (hi.x + hi.y); }
};
template <> struct Dot16<double> {
    double s0 = 0.0, s1 = 0.0;
    __device__ __forceinline__ void add(const double (&a)[2], const double (&b)[2]) {
        s0 = __builtin_fma(a[0], b[0], s0);
        s1 = __builtin_fma(a[1], b[1], s1);
    }
    __device__ __forceinline__ double total() const { return s0 + s1; }
};
struct SdDev {
    const int* ptr;               // [nb+1]
    const unsigned short* la;
    const unsigned short* lb;
    const int* epos;
};
constexpr int SD_ROUNDS = 3;
template <typename T>
__global__ __launch_bounds__(BLK_THREADS) void k_sddmm_blk(BlkDev B, SdDev S, PatternDev<T> P, int Dpad, int ntiles,
                                                           const T* __restrict__ Yb, const T* __restrict__ d,
                                                           const double* __restrict__ tr_part, int ntr, T* __restrict__ xval,
                                                           T* __restrict__ xavg, int accumulate) {
    constexpr int VEC = V16<T>::N;
    constexpr int CT = BLK_TILE_BYTES / (int)sizeof(T);
    constexpr int RPP = BLK_THREADS / 16;
    constexpr int NG = (BLK_UNION_ROWS + RPP - 1) / RPP;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);
    __shared__ double sh[BLK_WAVES];
    double tsum = 0.0;
    for (int i = threadIdx.x; i < ntr; i += BLK_THREADS) tsum += tr_part[i];
    tsum = block_sum(tsum, sh);
    const double tr = tsum / (double)P.K;
    const int per = (B.nb + 7) / 8;
    const int rb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (rb >= B.nb) return;
    const int l16 = threadIdx.x & 15;
    const int u0 = threadIdx.x >> 4;
    const int nun = B.desc[(size_t)rb * 8 + 5];
    unsigned gbase[NG];  // element offsets of the union rows (K * Dpad < 2^32)
#pragma unroll
    for (int j = 0; j < NG; ++j) gbase[j] = (unsigned)B.un_fixed[(size_t)rb * BLK_UNION_ROWS + min(u0 + j * RPP, BLK_UNION_ROWS - 1)] * (unsigned)Dpad;
    T x[NG][VEC];
    auto gather = [&](int t) {
        const int c = t * CT + l16 * VEC;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[j][v] = T(0);
            if (u0 + j * RPP < nun && c < Dpad) load16(Yb + (size_t)gbase[j] + c, x[j]);
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int j = 0; j < NG; ++j)
            if (u0 + j * RPP < nun) store16(tile + (size_t)(u0 + j * RPP) * CT + l16 * VEC, x[j]);
    };
    gather(0);
    // this thread's entries
    const int e0 = S.ptr[rb], ne = S.ptr[rb + 1] - e0;
    unsigned offa[SD_ROUNDS], offb[SD_ROUNDS];
    int ep[SD_ROUNDS];
    double acc[SD_ROUNDS];
#pragma unroll
    for (int k = 0; k < SD_ROUNDS; ++k) {
        const int i = threadIdx.x + k * BLK_THREADS;
        acc[k] = 0.0;
        ep[k] = -1;
        offa[k] = offb[k] = 0;
        if (i < ne) {
            offa[k] = (unsigned)S.la[e0 + i] * BLK_TILE_BYTES;
            offb[k] = (unsigned)S.lb[e0 + i] * BLK_TILE_BYTES;
            ep[k] = S.epos[e0 + i];
        }
    }
    deposit();
    __syncthreads();
    const char* tb = reinterpret_cast<const char*>(tile);
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) gather(t + 1);
#pragma unroll
        for (int k = 0; k < SD_ROUNDS; ++k) {
            if (ep[k] >= 0) {
                Dot16<T> s;
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    const int ch = ((st + l16) & 15) * 16;
                    T xa[VEC], xb[VEC];
                    load16(reinterpret_cast<const T*>(tb + offa[k] + ch), xa);
                    load16(reinterpret_cast<const T*>(tb + offb[k] + ch), xb);
                    s.add(xa, xb);
                }
                acc[k] += (double)s.total();  // one widening per entry and tile
            }
        }
        __syncthreads();
        if (t + 1 < ntiles) {
            deposit();
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < SD_ROUNDS; ++k)
        if (ep[k] >= 0) {
            const T xv = (T)(acc[k] / tr);
            const int e = ep[k], me = P.mirror[e];
            xval[e] = xv;
            xval[me] = xv;
        }
    // diagonal of the block's rows
    const int q0 = B.rowptr[rb], q1 = B.rowptr[rb + 1];
    for (int q = q0 + threadIdx.x; q < q1; q += BLK_THREADS) {
        const int row = B.order[q];
        const int dp = P.diag_pos[row];
        const T xd = (T)((double)d[row] / tr);
        xval[dp] = xd;
    }
}

// ---- the same on 128-byte half tiles, up to three workgroups per CU ---------------------------------
// 512 threads, LDS = the union's rows at 128 B (<= 56 KiB), so two or three workgroups share a CU and one computes
// while another waits on its gathers or barriers.  Thread-per-entry as above, 8 column chunks walked in the
// lane-skewed order (s + lane) mod 8.  With 128-byte rows a chunk of an even and of an odd staged row lie in different
// bank halves, and the host deals the entries so that the two lanes of a ds_read_b128 service group that read the same
// chunk hold rows of opposite parity (blocking.h, sd2_*): conflict-free for both operands.
struct Sd2Dev {
    const int* ptr;            // [nb+1] slot ranges
    const unsigned* ab;        // la | lb << 16
    const int* epos;           // -1 idle
    const int* items;          // [nitems][3] {row block, first round, end round}
    int nitems;
};
template <typename T> constexpr int sd2_rounds() { return 4; }  // rounds per work item: the host cuts longer blocks into several items (64 VGPRs)
template <typename T>
// float: <= 64 VGPRs, two 16-wave workgroups per CU; double: 128 VGPRs, one workgroup per CU (it would spill at 64)
__global__ __launch_bounds__(SD2_THREADS, sizeof(T) == 4 ? 8 : 4)
void k_sddmm_blk2(BlkDev B, Sd2Dev S, PatternDev<T> P, int Dpad, int ntiles,
                                                            const T* __restrict__ Yb, const T* __restrict__ d,
                                                            const double* __restrict__ tr_part, int ntr, T* __restrict__ xval,
                                                            T* __restrict__ xavg, int accumulate, SketchArgs<T> sk, unsigned long long* __restrict__ stamps) {
    constexpr int VEC = V16<T>::N;
    constexpr int CT = B2_ROW_BYTES / (int)sizeof(T);
    constexpr int SD2_ROUNDS = sd2_rounds<T>();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    {
        const int grid_sd = (S.nitems + 7) / 8 * 8;
        if ((int)blockIdx.x >= grid_sd) {  // appended sketch workgroups (dispatched last: they fill the slots the items leave)
            sketch_rows<T, SD2_THREADS / WAVE>(sk.K, sk.D, Dpad, sk.seed, sk.iter, sk.R, sk.colsq_part, (int)blockIdx.x - grid_sd, sk.nblocks,
                                               reinterpret_cast<double*>(smem_raw), sk.planes);
            return;
        }
    }
#define MMW_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    MMW_STAMP(0);
    if (stamps && threadIdx.x == 0) {
        stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg(63492);
        stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_getreg(63508);
    }
    constexpr int RPP = SD2_THREADS / 8;
    constexpr int NG = (BLK_UNION_ROWS + RPP - 1) / RPP;
    constexpr int NW = SD2_THREADS / WAVE;
    char* tile = smem_raw;
    __shared__ double sh[NW];
    double tsum = 0.0;
    for (int i = threadIdx.x; i < ntr; i += SD2_THREADS) tsum += tr_part[i];
    tsum = block_sum(tsum, sh);
    const double tr = tsum / (double)P.K;
    // work items (row block, first round, end round), longest first; XCD-aware: consecutive items share an XCD's L2
    const int per = (S.nitems + 7) / 8;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= S.nitems) return;
    const int rb = S.items[3 * item], k0 = S.items[3 * item + 1], k1 = S.items[3 * item + 2];
    const int l8 = threadIdx.x & 7, u0 = threadIdx.x >> 3;
    const int nun8 = (B.desc[(size_t)rb * 8 + 5] + 7) & ~7;
    unsigned gbase[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j)
        gbase[j] = (unsigned)B.un_fixed[(size_t)rb * BLK_UNION_ROWS + min(u0 + j * RPP, BLK_UNION_ROWS - 1)] * (unsigned)(Dpad * (int)sizeof(T)) + (unsigned)(l8 * 16);
    const int wrow = (threadIdx.x >> 6) * 8;
    const char* Ub = reinterpret_cast<const char*>(Yb);
    T x[NG][VEC];
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) x[j][v] = T(0);
    auto gather_one = [&](int t, int j) {  // one 1-KiB piece (8 rows x 128 B) per wave
        const unsigned cb = (unsigned)(t * B2_ROW_BYTES);
        const bool ok = t * CT + l8 * VEC < Dpad;  // only a last, partial tile tests lanes
        if (wrow + j * RPP < nun8 && ok) load16(reinterpret_cast<const T*>(Ub + (gbase[j] + cb)), x[j]);
    };
    auto gather = [&](int t) {
#pragma unroll
        for (int j = 0; j < NG; ++j) gather_one(t, j);
    };
    auto deposit = [&](bool full) {  // columns past Dpad are staged as zeros: every lane reads every chunk of its rows
#pragma unroll
        for (int j = 0; j < NG; ++j)
            if (wrow + j * RPP < nun8) {
                if (!full) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) x[j][v] = T(0);
                }
                store16(reinterpret_cast<T*>(tile + (size_t)(u0 + j * RPP) * B2_ROW_BYTES) + l8 * VEC, x[j]);
            }
    };
    MMW_STAMP(1);
    gather(0);
    const int s0 = S.ptr[rb] + k0 * SD2_THREADS, rounds = k1 - k0;
    unsigned ab[SD2_ROUNDS];  // staged row indices of the entry's two rows, packed (registers are the budget here)
    int ep[SD2_ROUNDS];
    T acc[SD2_ROUNDS];
#pragma unroll
    for (int k = 0; k < SD2_ROUNDS; ++k) {
        acc[k] = T(0);
        ep[k] = -1;
        ab[k] = 0;
        if (k < rounds) {
            ab[k] = S.ab[s0 + k * SD2_THREADS + threadIdx.x];
            ep[k] = S.epos[s0 + k * SD2_THREADS + threadIdx.x];
        }
    }
    MMW_STAMP(2);
    deposit(l8 * VEC < Dpad);
    MMW_STAMP(3);
    __syncthreads();
    MMW_STAMP(4);
    const int skew = threadIdx.x & 7;
    for (int t = 0; t < ntiles; ++t) {
        // the next tile's gathers are fed to the memory pipe a few at a time between the rounds: issued all at once they
        // fill its queue and every wave stalls on issue (1.6 us per tile) before it can start on the LDS
        constexpr int GPR = (NG + SD2_ROUNDS - 1) / SD2_ROUNDS;
        if (t == 0) MMW_STAMP(12);
#pragma unroll
        for (int k = 0; k < SD2_ROUNDS; ++k) {
            if (t + 1 < ntiles) {
#pragma unroll
                for (int j = k * GPR; j < (k + 1) * GPR && j < NG; ++j) gather_one(t + 1, j);
            }
            if (k < rounds && ep[k] >= 0) {
                Dot16<T> s;
                const char* ra = tile + (ab[k] & 0xFFFFu) * B2_ROW_BYTES;
                const char* rb_ = tile + (ab[k] >> 16) * B2_ROW_BYTES;
#pragma unroll 4
                for (int st = 0; st < 8; ++st) {  // 8 staged-row reads in flight (the register budget allows no more)
                    const int ch = ((st + skew) & 7) * 16;
                    T xa[VEC], xb[VEC];
                    load16(reinterpret_cast<const T*>(ra + ch), xa);
                    load16(reinterpret_cast<const T*>(rb_ + ch), xb);
                    s.add(xa, xb);
                }
                acc[k] += s.total();
            }
        }
        if (t == 0) MMW_STAMP(5);
        __syncthreads();
        if (t == 0) MMW_STAMP(6);
        if (t + 1 < ntiles) {
            deposit((t + 1) * CT + l8 * VEC < Dpad);
            if (t == 0) MMW_STAMP(7);
            __syncthreads();
            if (t == 0) MMW_STAMP(8);
        }
    }
#pragma unroll
    for (int k = 0; k < SD2_ROUNDS; ++k)
        if (ep[k] >= 0) {
            const T xv = (T)((double)acc[k] / tr);
            const int e = ep[k], me = P.mirror[e];
            xval[e] = xv;
            xval[me] = xv;  // the running sum xavg += xval is a coalesced pass of its own (k_accumulate), not 2 more scattered RMWs here
        }
    const int q0 = B.rowptr[rb], q1 = k0 == 0 ? B.rowptr[rb + 1] : B.rowptr[rb];  // the item with the first round writes the diagonal
    for (int q = q0 + threadIdx.x; q < q1; q += SD2_THREADS) {
        const int row = B.order[q];
        const int dp = P.diag_pos[row];
        const T xd = (T)((double)d[row] / tr);
        xval[dp] = xd;
    }
    if (stamps) {  // diagnostic runs: the end stamp waits for every wave's stores to be issued
        MMW_STAMP(13);
        __syncthreads();
    }
    MMW_STAMP(9);
#undef MMW_STAMP
}

// ---- on-device Gaussian sketch: rows of unit 2-norm (mmw.py:226-227) -----------------------------
// One wavefront per row, one pass: every lane draws its 16 bytes of the row (4 floats / 2 doubles) per
// step from Philox4x32-10 keyed by (seed; row, column group, iteration), keeps them in registers, the
// wave reduces the squared norm and the normalised row is written with one 16-B store per lane.
__device__ __forceinline__ void normals16(const uint32_t (&w)[4], float (&n)[4]) {
    const float u1 = (float)((w[0] >> 8) + 1u) * 5.9604644775390625e-8f;  // (0, 1]
    const float u2 = (float)(w[1] >> 8) * 5.9604644775390625e-8f;
    const float u3 = (float)((w[2] >> 8) + 1u) * 5.9604644775390625e-8f;
    const float u4 = (float)(w[3] >> 8) * 5.9604644775390625e-8f;
    // hardware transcendentals: v_log_f32, v_sqrt_f32 and v_sin/cos_f32, whose argument is in revolutions (u in [0,1))
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u) = sqrt(-2 ln2 log2 u)
    const float r2 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u3));
    const float s1 = __builtin_amdgcn_sinf(u2), c1 = __builtin_amdgcn_cosf(u2);
    const float s2 = __builtin_amdgcn_sinf(u4), c2 = __builtin_amdgcn_cosf(u4);
    n[0] = r1 * c1; n[1] = r1 * s1; n[2] = r2 * c2; n[3] = r2 * s2;
}
__device__ __forceinline__ void normals16(const uint32_t (&w)[4], double (&n)[2]) { box_muller(w, n[0], n[1]); }

// body shared by k_sketch_rng and by the sketch workgroups appended to the SDDMM launch (NWAVES waves per workgroup,
// `bid` of `nblocks` workgroups, `shc` = NWAVES x Dpad doubles of LDS when colsq_part)
template <typename T, int NWAVES>
__device__ __forceinline__ void sketch_rows(int K, int D, int Dpad, uint64_t seed, uint32_t iter, T* __restrict__ R,
                                            double* __restrict__ colsq_part, int bid, int nblocks, double* shc,
                                            unsigned short* __restrict__ planes, int planes_f16, double* __restrict__ dusq_part) {
    constexpr int VEC = V16<T>::N;
    constexpr int NS = 4;  // 64 lanes x 4 steps x 16 B covers Dpad <= 1024 floats / 512 doubles
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int ngroups = Dpad / VEC;
    T csq[NS][VEC];  // a wave sums a handful of unit-norm rows: T is enough, widened once at the end
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) csq[i][v] = T(0);
    // What the fp16 plane of the first-order product loses, column by column: sum_rows (u - fp16(u))^2.  The certificate of that
    // product (first_verify) uses this measured rounding instead of the format's worst case 2^-11 |u|.  The sums are kept in LDS
    // (Dpad 64-bit words behind the column-square slabs, one LDS add per element): sixteen more accumulators per lane took the LOSS pass
    // this body rides in from 7 to 5 waves per SIMD (72 -> 87 registers, 22.9 -> 25.7 us).  The adds are 2^-60 fixed-point INTEGERS,
    // each term rounded up: the sum does not depend on the order the waves arrive in, and it is an upper bound.
    unsigned long long* shd = reinterpret_cast<unsigned long long*>(shc + (size_t)NWAVES * Dpad);
    const bool measure = sizeof(T) == 4 && dusq_part != nullptr && planes != nullptr && planes_f16 != 0;
    if (measure) {
        for (int i = threadIdx.x; i < (NWAVES / WAVES_PER_BLOCK) * Dpad; i += NWAVES * WAVE) shd[i] = 0ull;
        __syncthreads();
    }
    // a workgroup of NWAVES waves stands for NWAVES/4 workgroups of the stand-alone kernel (same rows per wave, same slabs),
    // so the start norms do not depend on which launch drew the sketch
    constexpr int VB = NWAVES / WAVES_PER_BLOCK;
    const int vb = bid * VB + wib / WAVES_PER_BLOCK, vw = wib % WAVES_PER_BLOCK;
    for (int row = vb * WAVES_PER_BLOCK + vw; row < K; row += nblocks * VB * WAVES_PER_BLOCK) {
        T n[NS][VEC];
        T ssl = T(0);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int p = lane + WAVE * i;
#pragma unroll
            for (int v = 0; v < VEC; ++v) n[i][v] = T(0);
            if (p < ngroups) {
                uint32_t w[4];
                philox4x32_10((uint32_t)row, (uint32_t)p, iter, 0x4d4d5753u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
                normals16(w, n[i]);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if (p * VEC + v >= D) n[i][v] = T(0);
                    ssl += n[i][v] * n[i][v];
                }
            }
        }
        const double ss = wave_sum((double)ssl);
        const T inv = (T)(ss > 0.0 ? 1.0 / sqrt(ss) : 0.0);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int p = lane + WAVE * i;
            if (p < ngroups) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    n[i][v] *= inv;
                    csq[i][v] += n[i][v] * n[i][v];
                }
                store16(R + (size_t)row * Dpad + (size_t)p * VEC, n[i]);
                if constexpr (sizeof(T) == 4) {
                    if (planes && planes_f16) {  // the first-order product reads ONE fp16 plane
                        const size_t o = ((size_t)row * Dpad + (size_t)p * VEC) >> 2;
                        const unsigned short h0 = f16_rn(n[i][0]), h1 = f16_rn(n[i][1]), h2 = f16_rn(n[i][2]), h3 = f16_rn(n[i][3]);
                        reinterpret_cast<uint2*>(planes)[o] = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
                        if (measure) {  // the differences are exact in fp32: u and fp16(u) agree in their leading bits
                            const float e0 = (float)n[i][0] - f16_f32(h0), e1 = (float)n[i][1] - f16_f32(h1), e2 = (float)n[i][2] - f16_f32(h2), e3 = (float)n[i][3] - f16_f32(h3);
                            unsigned long long* dst = shd + (wib / WAVES_PER_BLOCK) * Dpad + p * VEC;
                            constexpr float FX = 1152921504606846976.0f;  // 2^60; |e| <= 2^-11, so a term stays below 2^38
                            atomicAdd(dst, (unsigned long long)ceilf(e0 * e0 * FX)); atomicAdd(dst + 1, (unsigned long long)ceilf(e1 * e1 * FX));
                            atomicAdd(dst + 2, (unsigned long long)ceilf(e2 * e2 * FX)); atomicAdd(dst + 3, (unsigned long long)ceilf(e3 * e3 * FX));
                        }
                    } else if (planes) {  // the matrix-core SpMM reads the block as two bf16 halves: made here, while the values are in registers
                        const unsigned a = split_bf16(n[i][0]), b = split_bf16(n[i][1]), c = split_bf16(n[i][2]), d = split_bf16(n[i][3]);
                        const size_t o = ((size_t)row * Dpad + (size_t)p * VEC) >> 2;
                        reinterpret_cast<uint2*>(planes)[o] = make_uint2((a >> 16) | (b & 0xFFFF0000u), (c >> 16) | (d & 0xFFFF0000u));
                        reinterpret_cast<uint2*>(planes + (size_t)K * Dpad)[o] = make_uint2((a & 0xFFFFu) | (b << 16), (c & 0xFFFFu) | (d << 16));
                    }
                }
            }
        }
    }
    if (colsq_part) {  // column sums of squares of this block's rows: the Lanczos start norms come for free
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int p = lane + WAVE * i;
            if (p < ngroups)
#pragma unroll
                for (int v = 0; v < VEC; ++v) shc[wib * Dpad + p * VEC + v] = (double)csq[i][v];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < VB * Dpad; i += NWAVES * WAVE) {
            const int h = i / Dpad, c = i - h * Dpad;
            double t = 0.0;
            for (int w = 0; w < WAVES_PER_BLOCK; ++w) t += shc[(h * WAVES_PER_BLOCK + w) * Dpad + c];
            colsq_part[(size_t)(bid * VB + h) * Dpad + c] = t;
        }
    }
    if (measure) {
        __syncthreads();
        for (int i = threadIdx.x; i < VB * Dpad; i += NWAVES * WAVE)
            dusq_part[(size_t)(bid * VB) * Dpad + i] = (double)shd[i] * 8.673617379884035e-19 * (1.0 + 2e-7);  // 2^-60; the fp32 squares were rounded
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_sketch_rng(int K, int D, int Dpad, uint64_t seed, uint32_t iter, T* __restrict__ R,
                                                      double* __restrict__ colsq_part, unsigned short* __restrict__ planes = nullptr, int planes_f16 = 0,
                                                      double* __restrict__ dusq_part = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    sketch_rows<T, WAVES_PER_BLOCK>(K, D, Dpad, seed, iter, R, colsq_part, blockIdx.x, gridDim.x, reinterpret_cast<double*>(smem_raw), planes, planes_f16, dusq_part);
}

// fragment image of the matrix-core SpMM rebuilt from the CSR values (after a snapshot restore)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_refrag(size_t n, const T* __restrict__ lval, const int* __restrict__ fpos, unsigned* __restrict__ afrag) {
    for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < n; e += (size_t)gridDim.x * BLOCK) afrag[fpos[e]] = split_bf16((float)lval[e]);
}
// xavg += xval over the whole pattern (the running sum of X; coalesced, 12 bytes per stored entry)
// snapshot / restore of the iterate in one launch: six value arrays (blockIdx.y) and the plan
template <typename T> struct CopySet {
    T* dst[6];
    const T* src[6];
    size_t n[6];
    ExpmPlan* plan_dst;
    const ExpmPlan* plan_src;
};
template <typename T> __global__ __launch_bounds__(BLOCK) void k_copy_state(CopySet<T> c) {
    const int seg = blockIdx.y;
    const T* __restrict__ s = c.src[seg];
    T* __restrict__ d = c.dst[seg];
    const size_t n = c.n[seg];
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) d[o] = s[o];
    if (seg == 0 && blockIdx.x == 0) {
        static_assert(sizeof(ExpmPlan) % 4 == 0, "plan copied as words");
        const unsigned* ps = reinterpret_cast<const unsigned*>(c.plan_src);
        unsigned* pd = reinterpret_cast<unsigned*>(c.plan_dst);
        for (int i = threadIdx.x; i < (int)(sizeof(ExpmPlan) / 4); i += BLOCK) pd[i] = ps[i];
    }
}
// X between the pattern's CSR order and the matrix-core SDDMM's tile order (kernels_mfma.h; e2w = slot of every CSR entry).  Both
// directions are gathers / scatters over the CSR entries: an edge's two entries hold the same value and share a slot.
template <typename T> __global__ __launch_bounds__(BLOCK) void k_x_tiles_to_csr(size_t nnz, const int* __restrict__ e2w, const T* __restrict__ t0, T* __restrict__ c0,
                                                                                 const T* __restrict__ t1, T* __restrict__ c1) {
    for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < nnz; e += (size_t)gridDim.x * BLOCK) {
        const int w = e2w[e];
        c0[e] = t0[w];
        c1[e] = t1[w];
    }
}
template <typename T> __global__ __launch_bounds__(BLOCK) void k_x_csr_to_tiles(size_t nnz, const int* __restrict__ e2w, const T* __restrict__ c0, T* __restrict__ t0,
                                                                                 const T* __restrict__ c1, T* __restrict__ t1) {
    for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < nnz; e += (size_t)gridDim.x * BLOCK) {
        const int w = e2w[e];
        t0[w] = c0[e];  // (both entries of an edge write the same value)
        t1[w] = c1[e];
    }
}
__global__ __launch_bounds__(BLOCK) void k_gather_idx(size_t n, const int* __restrict__ idx, const int* __restrict__ src, int* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) dst[i] = src[idx[i]];
}
template <typename T> __global__ __launch_bounds__(BLOCK) void k_accumulate(size_t n, const T* __restrict__ x, T* __restrict__ sum) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) sum[o] += x[o];
}
template <typename T> __global__ __launch_bounds__(BLOCK) void k_fill(size_t n, T* __restrict__ a, T v) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) a[o] = v;
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_set_identity(int K, const int* __restrict__ diag_pos, T* __restrict__ xval, T* __restrict__ xavg) {
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < K; k += gridDim.x * BLOCK) {
        xval[diag_pos[k]] = T(1);
        xavg[diag_pos[k]] = T(1);
    }
}
template <typename T> __global__ __launch_bounds__(BLOCK) void k_to_f64(size_t n, const T* __restrict__ a, double* __restrict__ b) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) b[o] = (double)a[o];
}

}  // namespace mmw
