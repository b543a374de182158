// Small dense building blocks for the epilogue factor (top-|lambda| invariant subspace of the averaged X,
// replacing scipy.sparse.linalg.svds at sim_src/alg/mmw.py:215) and the duality-gap Lanczos
// (eigsh at mmw.py:115):
//   gram      G = V^T W            (b x b, reduction over the K rows)     -- fp64 matrix cores
//   gemm_tall C = V Q              (K x b times b x n)                    -- fp64 matrix cores, LDS tiles
//   jacobi    H = Q Theta Q^T      (parallel cyclic Jacobi, two launches per round)
// plus the element-wise helpers around them.  These are the dense contractions of the path; everything
// here accumulates in float64 whatever the block dtype T is.
#pragma once
#include "device_utils.h"

namespace mmw {

typedef double d4 __attribute__((ext_vector_type(4)));

// ---- G_part[slice] (b x b) += V[rows of slice]^T W[rows of slice] ---------------------------------
// grid = (ceil(b/64) j-tiles, ceil(b/16) i-tiles, nslice); one wavefront per workgroup tile 16(i) x 64(j).
// v_mfma_f64_16x16x4_f64: lane l feeds A[i = l&15][k = l>>4] = V[k][i] and B[k][j = l&15] = W[k][j]:
// both are 16 consecutive elements of a block row -> coalesced straight from global memory.
template <typename T>
__global__ __launch_bounds__(WAVE) void k_gram(int K, int b, int ld, const T* __restrict__ V, const T* __restrict__ W,
                                               int rows_per_slice, double* __restrict__ Gpart) {
    const int lane = threadIdx.x;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 16, sl = blockIdx.z;
    const int r_beg = sl * rows_per_slice;
    const int r_end = min(K, r_beg + rows_per_slice);
    const int li = lane & 15, lk = lane >> 4;
    d4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (d4){0.0, 0.0, 0.0, 0.0};
    const bool iok = i0 + li < b;
    bool jok[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) jok[n] = j0 + n * 16 + li < b;
    for (int r = r_beg; r < r_end; r += 4) {
        const int row = r + lk;
        const bool rok = row < r_end;
        const double a = (rok && iok) ? (double)V[(size_t)row * ld + i0 + li] : 0.0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const double bb = (rok && jok[n]) ? (double)W[(size_t)row * ld + j0 + n * 16 + li] : 0.0;
            acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[n], 0, 0, 0);
        }
    }
    double* G = Gpart + (size_t)sl * b * b;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + lk + 4 * q, j = j0 + n * 16 + li;
            if (i < b && j < b) G[(size_t)i * b + j] = acc[n][q];
        }
}
// The same product from LDS tiles: a workgroup of four waves owns a 64 x 64 tile of G for one slice of rows and walks the rows 16 at a
// time -- every thread brings 16 bytes of V and of W per stage (a row's 64 columns are one 256-byte run), double-buffered, and wave w
// multiplies its 16 columns of V with the tile's 64 columns of W exactly as k_gram does.  k_gram's one wave per 16 x 64 tile fetched
// its operands in 64-byte runs, one row per lane group, W once per 16 rows of G: 627 MB of 64-byte requests per call at b = 444
// (418 us, 9 TFLOP/s); here 250 MB in 256-byte runs.  Row stride 80 floats in LDS: the four k-rows of a fragment read fall on four
// different 16-bank groups.
constexpr int GR_ROWS = 16, GR_LD = 80;
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gram_tiles(int K, int b, int ld, const T* __restrict__ V, const T* __restrict__ W,
                                                     int rows_per_slice, double* __restrict__ Gpart) {
    __shared__ T sV[2][GR_ROWS][GR_LD], sW[2][GR_ROWS][GR_LD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64, sl = blockIdx.z;
    const int r_beg = sl * rows_per_slice, r_end = min(K, r_beg + rows_per_slice);
    const int li = lane & 15, lk = lane >> 4;
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 4;  // this thread's row of a stage and its four columns
    d4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (d4){0.0, 0.0, 0.0, 0.0};
    T rv[4], rw[4];
    auto fetch = [&](int r0) {
        const int row = r0 + tr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rv[q] = (row < r_end && i0 + tc + q < b) ? V[(size_t)row * ld + i0 + tc + q] : T(0);
            rw[q] = (row < r_end && j0 + tc + q < b) ? W[(size_t)row * ld + j0 + tc + q] : T(0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sV[buf][tr][tc + q] = rv[q]; sW[buf][tr][tc + q] = rw[q]; }
    };
    fetch(r_beg);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int r = r_beg; r < r_end; r += GR_ROWS) {
        const bool more = r + GR_ROWS < r_end;
        if (more) fetch(r + GR_ROWS);  // in flight under this stage's products
#pragma unroll
        for (int kk = 0; kk < GR_ROWS; kk += 4) {
            const double a = (double)sV[buf][kk + lk][16 * wv + li];
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (double)sW[buf][kk + lk][16 * n + li], acc[n], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    double* G = Gpart + (size_t)sl * b * b;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + 16 * wv + lk + 4 * q, j = j0 + n * 16 + li;
            if (i < b && j < b) G[(size_t)i * b + j] = acc[n][q];
        }
}
// G = sum over slices, optionally symmetrised
__global__ __launch_bounds__(BLOCK) void k_gram_reduce(int b, int nslice, const double* __restrict__ Gpart, double* __restrict__ G,
                                                       int symmetrise) {
    const int n = b * b;
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < n; o += gridDim.x * BLOCK) {
        const int i = o / b, j = o % b;
        double s = 0.0;
        for (int t = 0; t < nslice; ++t) s += Gpart[(size_t)t * n + o];
        if (symmetrise) {
            double u = 0.0;
            for (int t = 0; t < nslice; ++t) u += Gpart[(size_t)t * n + (size_t)j * b + i];
            s = 0.5 * (s + u);
        }
        G[o] = s;
    }
}

// ---- C[K x n] = V[K x b] * Q[b x n] (Q float64 row-major, leading dim ldq) -------------------------
// workgroup = 4 waves = 64 rows x 64 columns; the k dimension is walked in LDS tiles of 16.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gemm_tall(int K, int b, int n, int ldv, const T* __restrict__ V, int ldq,
                                                     const double* __restrict__ Q, int ldc, T* __restrict__ C) {
    __shared__ double sA[64][17];
    __shared__ double sB[16][65];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int li = lane & 15, lk = lane >> 4;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < b; k0 += 16) {
        for (int t = threadIdx.x; t < 64 * 16; t += BLOCK) {
            const int r = t >> 4, c = t & 15;
            sA[r][c] = (r0 + r < K && k0 + c < b) ? (double)V[(size_t)(r0 + r) * ldv + k0 + c] : 0.0;
        }
        for (int t = threadIdx.x; t < 16 * 64; t += BLOCK) {
            const int r = t >> 6, c = t & 63;
            sB[r][c] = (k0 + r < b && c0 + c < n) ? Q[(size_t)(k0 + r) * ldq + c0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            const double a = sA[wib * 16 + li][kk + lk];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sB[kk + lk][t * 16 + li], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + wib * 16 + lk + 4 * q, c = c0 + t * 16 + li;
            if (r < K && c < n) C[(size_t)r * ldc + c] = (T)acc[t][q];
        }
}

// ---- parallel cyclic Jacobi on a symmetric b x b matrix H (float64), eigenvectors accumulated in Q --
// Round-robin ("tournament") ordering: n = even padded size, round r pairs up all indices disjointly.
__device__ __forceinline__ void jacobi_pair(int n, int r, int i, int& p, int& q) {
    // positions 0..n-1 around a table, index n-1 fixed; standard circle method
    const int m = n - 1;
    if (i == 0) {
        p = m;
        q = r % m;
    } else {
        p = (r + i) % m;
        q = (r - i + m) % m;
    }
    if (p > q) {
        const int t = p;
        p = q;
        q = t;
    }
}
// rotation (c, s) that annihilates H[p][q]
__device__ __forceinline__ void jacobi_cs(const double* __restrict__ H, int b, int p, int q, double& c, double& s) {
    c = 1.0;
    s = 0.0;
    if (q < b) {
        const double apq = H[(size_t)p * b + q];
        if (apq != 0.0) {
            const double app = H[(size_t)p * b + p], aqq = H[(size_t)q * b + q];
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            s = t * c;
        }
    }
}
// one Jacobi round in ONE launch, ping-pong buffers: Hout = J^T Hin J on 2x2 blocks (pair i, pair j), Qout = Qin J.
// Every thread rebuilds the two rotations it needs from the untouched input matrix, so there is no
// separate parameter pass and no race.
__global__ __launch_bounds__(BLOCK) void k_jacobi_round(int b, int n, int r, const double* __restrict__ Hin, double* __restrict__ Hout,
                                                        const double* __restrict__ Qin, double* __restrict__ Qout) {
    const int half = n / 2;
    const int total = half * half;
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < total; o += gridDim.x * BLOCK) {
        const int i = o / half, j = o % half;
        int pi, qi, pj, qj;
        jacobi_pair(n, r, i, pi, qi);
        jacobi_pair(n, r, j, pj, qj);
        double ci, si, cj, sj;
        jacobi_cs(Hin, b, pi, qi, ci, si);
        jacobi_cs(Hin, b, pj, qj, cj, sj);
        const bool qi_ok = qi < b, qj_ok = qj < b;
        double h00 = Hin[(size_t)pi * b + pj];
        double h01 = qj_ok ? Hin[(size_t)pi * b + qj] : 0.0;
        double h10 = qi_ok ? Hin[(size_t)qi * b + pj] : 0.0;
        double h11 = (qi_ok && qj_ok) ? Hin[(size_t)qi * b + qj] : 0.0;
        const double t00 = ci * h00 - si * h10, t01 = ci * h01 - si * h11;
        const double t10 = si * h00 + ci * h10, t11 = si * h01 + ci * h11;
        h00 = t00 * cj - t01 * sj;
        h01 = t00 * sj + t01 * cj;
        h10 = t10 * cj - t11 * sj;
        h11 = t10 * sj + t11 * cj;
        Hout[(size_t)pi * b + pj] = h00;
        if (qj_ok) Hout[(size_t)pi * b + qj] = h01;
        if (qi_ok) Hout[(size_t)qi * b + pj] = h10;
        if (qi_ok && qj_ok) Hout[(size_t)qi * b + qj] = h11;
    }
    const int totq = b * half;
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < totq; o += gridDim.x * BLOCK) {
        const int x = o / half, j = o % half;
        int pj, qj;
        jacobi_pair(n, r, j, pj, qj);
        const double a = Qin[(size_t)x * b + pj];
        if (qj >= b) {
            Qout[(size_t)x * b + pj] = a;
            continue;
        }
        double cj, sj;
        jacobi_cs(Hin, b, pj, qj, cj, sj);
        const double d = Qin[(size_t)x * b + qj];
        Qout[(size_t)x * b + pj] = a * cj - d * sj;
        Qout[(size_t)x * b + qj] = a * sj + d * cj;
    }
}
// ---- whole Jacobi eigensolve of a small matrix (b <= JAC_LDS_MAX) in ONE launch: H and Q live in LDS, rounds are
// separated by workgroup barriers instead of kernel boundaries.  Same rotations and ordering as k_jacobi_round.
constexpr int JAC_LDS_MAX = 96;
__global__ __launch_bounds__(1024) void k_jacobi_lds(int b, const double* __restrict__ Hin, double* __restrict__ diag, double* __restrict__ Qout,
                                                     double rel_tol, int max_sweeps, int* __restrict__ sweeps_out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* H = reinterpret_cast<double*>(smem_raw);  // [b][b]
    double* Q = H + (size_t)b * b;                    // [b][b]
    double* cs = Q + (size_t)b * b;                   // [n]
    __shared__ double red[32];
    __shared__ int stop;
    const int n = (b % 2 == 0) ? b : b + 1, half = n / 2;
    for (int o = threadIdx.x; o < b * b; o += blockDim.x) {
        H[o] = Hin[o];
        Q[o] = (o / b == o % b) ? 1.0 : 0.0;
    }
    __syncthreads();
    int sw = 0;
    for (; sw < max_sweeps; ++sw) {
        // off-diagonal norm and diagonal scale
        double s = 0.0, d = 0.0;
        for (int o = threadIdx.x; o < b * b; o += blockDim.x) {
            const int i = o / b, j = o % b;
            const double h = H[o];
            if (i != j) s += h * h;
            else d = fabs(h) > d ? fabs(h) : d;
        }
        s = wave_sum(s);
        d = wave_max(d);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[16 + (threadIdx.x >> 6)] = d; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double ts = 0.0, td = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += red[w]; td = red[16 + w] > td ? red[16 + w] : td; }
            stop = !(sqrt(ts) > rel_tol * td * sqrt((double)b)) || b < 2;
        }
        __syncthreads();
        if (stop) break;
        for (int r = 0; r < n - 1; ++r) {
            for (int i = threadIdx.x; i < half; i += blockDim.x) {
                int p, q;
                jacobi_pair(n, r, i, p, q);
                double c, sn;
                jacobi_cs(H, b, p, q, c, sn);
                cs[2 * i] = c;
                cs[2 * i + 1] = sn;
            }
            __syncthreads();
            for (int o = threadIdx.x; o < half * half; o += blockDim.x) {
                const int i = o / half, j = o % half;
                int pi, qi, pj, qj;
                jacobi_pair(n, r, i, pi, qi);
                jacobi_pair(n, r, j, pj, qj);
                const double ci = cs[2 * i], si = cs[2 * i + 1], cj = cs[2 * j], sj = cs[2 * j + 1];
                const bool qi_ok = qi < b, qj_ok = qj < b;
                double h00 = H[pi * b + pj];
                double h01 = qj_ok ? H[pi * b + qj] : 0.0;
                double h10 = qi_ok ? H[qi * b + pj] : 0.0;
                double h11 = (qi_ok && qj_ok) ? H[qi * b + qj] : 0.0;
                const double t00 = ci * h00 - si * h10, t01 = ci * h01 - si * h11;
                const double t10 = si * h00 + ci * h10, t11 = si * h01 + ci * h11;
                H[pi * b + pj] = t00 * cj - t01 * sj;
                if (qj_ok) H[pi * b + qj] = t00 * sj + t01 * cj;
                if (qi_ok) H[qi * b + pj] = t10 * cj - t11 * sj;
                if (qi_ok && qj_ok) H[qi * b + qj] = t10 * sj + t11 * cj;
            }
            for (int o = threadIdx.x; o < b * half; o += blockDim.x) {
                const int x = o / half, j = o % half;
                int pj, qj;
                jacobi_pair(n, r, j, pj, qj);
                if (qj >= b) continue;
                const double cj = cs[2 * j], sj = cs[2 * j + 1];
                const double a = Q[x * b + pj], dd = Q[x * b + qj];
                Q[x * b + pj] = a * cj - dd * sj;
                Q[x * b + qj] = a * sj + dd * cj;
            }
            __syncthreads();
        }
    }
    for (int o = threadIdx.x; o < b * b; o += blockDim.x) Qout[o] = Q[o];
    for (int i = threadIdx.x; i < b; i += blockDim.x) diag[i] = H[i * b + i];
    if (threadIdx.x == 0 && sweeps_out) *sweeps_out = sw;
}
// ---- block Jacobi: the eigensolve above one round per launch (b - 1 launches a sweep, each a few microseconds of work) is
// launch-bound.  Here the matrix is cut into 32-wide block columns (M of them, M even, zero-padded to P = 32 M); a block round
// pairs them up disjointly (the same round-robin order, on blocks), and per round two launches do the work of ~63 element
// rounds: k_bj_solve runs a cyclic Jacobi sweep on every pair's 64 x 64 diagonal problem in LDS (rotations as in
// k_jacobi_lds) and leaves the accumulated 64 x 64 orthogonal factor R_a per pair; k_bj_apply forms H <- R^T H R and Q <- Q R
// from the untouched input (ping-pong buffers), one workgroup per pair of pairs / per 64 rows of Q.  M - 1 block rounds make a
// block sweep: every element pair has then been rotated at least once.  Padding rows are zero, so no rotation touches them.
constexpr int BJ_NB = 32, BJ_N2 = 64;
__global__ __launch_bounds__(BLOCK) void k_bj_pad(int b, int P, const double* __restrict__ G, double* __restrict__ Hp, double* __restrict__ Qp) {
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < P * P; o += gridDim.x * BLOCK) {
        const int i = o / P, j = o % P;
        Hp[o] = (i < b && j < b) ? G[(size_t)i * b + j] : 0.0;
        Qp[o] = i == j ? 1.0 : 0.0;
    }
}
__global__ __launch_bounds__(BLOCK) void k_bj_unpad(int b, int P, const double* __restrict__ Hp, const double* __restrict__ Qp,
                                                    double* __restrict__ diag, double* __restrict__ Q) {
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < b * b; o += gridDim.x * BLOCK) {
        const int i = o / b, j = o % b;
        Q[o] = Qp[(size_t)i * P + j];
        if (i == j) diag[i] = Hp[(size_t)i * P + i];
    }
}
__device__ __forceinline__ int bj_index(int I, int J, int k) { return k < BJ_NB ? BJ_NB * I + k : BJ_NB * J + (k - BJ_NB); }
// 1024 threads: thread (i, j) owns the 2 x 2 quad of S that rotation pairs i and j meet in, and two rows of Qm for pair j.
// `full` (the sweep's first block round): all 63 rounds of the 64-index tournament, so the pairs inside a block column are
// rotated once per sweep too; otherwise only the 32 rounds that pair an index of block I with one of block J -- the pairs this
// meeting exists for.  The rotation angle comes from a single-precision tangent (its exact value only decides how much of S_pq
// is left for the next meeting); c and s are then formed in double, so every rotation is orthogonal to rounding.
__device__ __forceinline__ void bj_pair(bool full, int rr, int i, int& p, int& q) {
    if (full) jacobi_pair(BJ_N2, rr, i, p, q);
    else { p = i; q = BJ_NB + ((i + rr) & (BJ_NB - 1)); }
}
__global__ __launch_bounds__(1024) void k_bj_solve(int M, int P, int r, int full, const double* __restrict__ Hin, double* __restrict__ R /* [M/2][64][64] */,
                                                   double skip_tol) {
    __shared__ double S[BJ_N2][BJ_N2 + 1], Qm[BJ_N2][BJ_N2 + 1], cs[BJ_N2];
    __shared__ double red[2][16];
    int I, J;
    jacobi_pair(M, r, (int)blockIdx.x, I, J);
    for (int o = threadIdx.x; o < BJ_N2 * BJ_N2; o += 1024) {
        const int k = o >> 6, l = o & 63;
        S[k][l] = Hin[(size_t)bj_index(I, J, k) * P + bj_index(I, J, l)];
        Qm[k][l] = k == l ? 1.0 : 0.0;
    }
    __syncthreads();
    {   // A pair whose off-diagonal part (the part this meeting rotates) is already below the solver's tolerance, relative to its own
        // diagonal, is left alone: in the last sweeps that is most pairs, and its rounds are the cost of a block round.
        double so = 0.0, dm = 0.0;
        for (int o = threadIdx.x; o < BJ_N2 * BJ_N2; o += 1024) {
            const int k = o >> 6, l = o & 63;
            const double v = S[k][l];
            if (k == l) dm = fabs(v) > dm ? fabs(v) : dm;
            else if (full || ((k < BJ_NB) != (l < BJ_NB))) so += v * v;
        }
        so = wave_sum(so);
        dm = wave_max(dm);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = so; red[1][threadIdx.x >> 6] = dm; }
        __syncthreads();
        so = 0.0; dm = 0.0;
        for (int w = 0; w < 16; ++w) { so += red[0][w]; dm = red[1][w] > dm ? red[1][w] : dm; }
        if (!(sqrt(so) > skip_tol * dm)) {
            double* out = R + (size_t)blockIdx.x * BJ_N2 * BJ_N2;
            for (int o = threadIdx.x; o < BJ_N2 * BJ_N2; o += 1024) out[o] = (o >> 6) == (o & 63) ? 1.0 : 0.0;
            return;
        }
    }
    const int i = threadIdx.x >> 5, j = threadIdx.x & 31;
    const int nrounds = full ? BJ_N2 - 1 : BJ_NB;
    for (int rr = 0; rr < nrounds; ++rr) {
        if (threadIdx.x < BJ_N2 / 2) {
            int p, q;
            bj_pair(full != 0, rr, (int)threadIdx.x, p, q);
            double c = 1.0, sn = 0.0;
            const float den = 2.f * (float)S[p][q];
            if (den != 0.f) {
                const float tau = (float)(S[q][q] - S[p][p]) / den;  // +-inf when S_pq is negligible: the tangent is then 0
                const float tf = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
                const double t = (double)tf;
                c = 1.0 / sqrt(1.0 + t * t);
                sn = t * c;
            }
            cs[2 * threadIdx.x] = c;
            cs[2 * threadIdx.x + 1] = sn;
        }
        __syncthreads();
        int pi, qi, pj, qj;
        bj_pair(full != 0, rr, i, pi, qi);
        bj_pair(full != 0, rr, j, pj, qj);
        const double ci = cs[2 * i], si = cs[2 * i + 1], cj = cs[2 * j], sj = cs[2 * j + 1];
        {
            const double h00 = S[pi][pj], h01 = S[pi][qj], h10 = S[qi][pj], h11 = S[qi][qj];
            const double t00 = ci * h00 - si * h10, t01 = ci * h01 - si * h11;
            const double t10 = si * h00 + ci * h10, t11 = si * h01 + ci * h11;
            S[pi][pj] = t00 * cj - t01 * sj;
            S[pi][qj] = t00 * sj + t01 * cj;
            S[qi][pj] = t10 * cj - t11 * sj;
            S[qi][qj] = t10 * sj + t11 * cj;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // Q <- Q J: row x = i + 32 h, column pair j
            const int x = i + 32 * h;
            const double a = Qm[x][pj], d = Qm[x][qj];
            Qm[x][pj] = a * cj - d * sj;
            Qm[x][qj] = a * sj + d * cj;
        }
        __syncthreads();
    }
    double* out = R + (size_t)blockIdx.x * BJ_N2 * BJ_N2;
    for (int o = threadIdx.x; o < BJ_N2 * BJ_N2; o += 1024) out[o] = Qm[o >> 6][o & 63];
}
// H part: workgroup (a, c) of the (M/2)^2 pairs of pairs: Hout[A, C] = R_a^T Hin[A, C] R_c.  Q part: workgroup (x, c):
// Qout[64 x .. 64 x + 63, C] = Qin[same rows, C] R_c.  256 threads, a 4 x 4 micro-tile each.
__global__ __launch_bounds__(BLOCK) void k_bj_apply(int M, int P, int r, const double* __restrict__ Hin, double* __restrict__ Hout,
                                                    const double* __restrict__ Qin, double* __restrict__ Qout, const double* __restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef double Tile[BJ_N2][BJ_N2 + 1];
    Tile& T = *reinterpret_cast<Tile*>(smem_raw);
    Tile& Ra = *reinterpret_cast<Tile*>(smem_raw + sizeof(Tile));
    Tile& Rc = *reinterpret_cast<Tile*>(smem_raw + 2 * sizeof(Tile));
    const int half = M / 2;
    const int nh = half * half;
    const bool qpart = (int)blockIdx.x >= nh;
    int a = 0, c, x = 0;
    if (!qpart) { a = (int)blockIdx.x / half; c = (int)blockIdx.x % half; }
    else { x = ((int)blockIdx.x - nh) / half; c = ((int)blockIdx.x - nh) % half; }
    int Ia = 0, Ja = 0, Ic, Jc;
    jacobi_pair(M, r, c, Ic, Jc);
    if (!qpart) jacobi_pair(M, r, a, Ia, Ja);
    const double* src = qpart ? Qin : Hin;
    for (int o = threadIdx.x; o < BJ_N2 * BJ_N2; o += BLOCK) {
        const int k = o >> 6, l = o & 63;
        const int gr = qpart ? BJ_N2 * x + k : bj_index(Ia, Ja, k);
        T[k][l] = src[(size_t)gr * P + bj_index(Ic, Jc, l)];
        Rc[k][l] = R[(size_t)c * BJ_N2 * BJ_N2 + o];
        if (!qpart) Ra[k][l] = R[(size_t)a * BJ_N2 * BJ_N2 + o];
    }
    __syncthreads();
    // 64 x 64 x 64 products on the fp64 matrix cores: wave w owns output rows 16 w .. 16 w + 15, four 16 x 16 column tiles
    // (operand / accumulator layout of v_mfma_f64_16x16x4_f64 as in k_gram)
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    d4 acc[4];
    if (!qpart) {  // tmp = Ra^T T, written back over T
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int kk = 0; kk < BJ_N2; kk += 4) {
            const double av = Ra[kk + lk][wib * 16 + li];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, T[kk + lk][t * 16 + li], acc[t], 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) T[wib * 16 + lk + 4 * q][t * 16 + li] = acc[t][q];
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < BJ_N2; kk += 4) {
        const double av = T[wib * 16 + li][kk + lk];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Rc[kk + lk][t * 16 + li], acc[t], 0, 0, 0);
    }
    double* dst = qpart ? Qout : Hout;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = wib * 16 + lk + 4 * q;
        const int gr = qpart ? BJ_N2 * x + k : bj_index(Ia, Ja, k);
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[(size_t)gr * P + bj_index(Ic, Jc, t * 16 + li)] = acc[t][q];
    }
}
constexpr int BJ_APPLY_LDS = 3 * BJ_N2 * (BJ_N2 + 1) * (int)sizeof(double);

// ---- Cholesky G = L L^T in place (lower triangle), right-looking in panels of 32 columns; *flag = 1 when a pivot is not
// positive.  Per panel two launches: k_chol_panel factors the 32 x 32 diagonal block in LDS (every workgroup redundantly --
// it is 3 us of work) and solves its slice of the rows below against it, one thread per row; k_chol_update subtracts the
// panel's outer product from the trailing lower triangle in 32 x 32 tiles.  (The unblocked one-workgroup version took
// 5 ms at b = 444; this takes 0.3 ms.)
constexpr int CH_NB = 32;
__device__ __forceinline__ double readlane_f64(double v, int l) {  // l: wave-uniform (a constant after unrolling)
    const long long bits = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xFFFFFFFFll), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ __launch_bounds__(BLOCK) void k_chol_panel(int b, int j0, double* __restrict__ G, int* __restrict__ flag,
                                                      double* __restrict__ Dfac /* [32*32] scratch */, int lds_form = 0,
                                                      double* __restrict__ Dinv = nullptr /* [panels][32*32]: the inverses of the diagonal blocks, for k_trsm_blocked */) {
    __shared__ double D[CH_NB][CH_NB + 1];
    __shared__ int bad;
    const int nb = min(CH_NB, b - j0);
    for (int o = threadIdx.x; o < CH_NB * CH_NB; o += BLOCK) {
        const int i = o / CH_NB, k = o % CH_NB;
        D[i][k] = (i < nb && k <= i) ? G[(size_t)(j0 + i) * b + j0 + k] : (i == k ? 1.0 : 0.0);
    }
    __shared__ double rdiag[CH_NB];  // 1 / D[k][k]
    __shared__ double Di[CH_NB][CH_NB + 1];  // D^-1, zero above the diagonal (padded: the matrix-core operand read walks a column)
    double* const Di_l = &Di[0][0];
    if (threadIdx.x == 0) bad = *flag;  // an earlier panel already failed: do nothing
    __syncthreads();
    if (bad) return;
    // The 32 x 32 block is factored by the first wave alone, in registers: lane i owns row i, a column step broadcasts what it needs with
    // v_readlane (the lane is a constant of the unrolled step) -- 496 broadcast + multiply-add pairs, no LDS round trip, no fence.  The
    // same operations in the same order as the LDS form below (kept for MMW_CHOL_LDS=1: three fenced LDS phases per column step, ~1 us
    // each, were 30 of the launch's 59 us).
    if (!lds_form && threadIdx.x < WAVE) {
        const int lane = threadIdx.x, row = lane & 31;
        double a[CH_NB];
#pragma unroll
        for (int k = 0; k < CH_NB; ++k) a[k] = D[row][k];
        bool fail_here = false;
        double rinv[CH_NB];  // 1 / D[j][j] (the same value in every lane)
#pragma unroll
        for (int j = 0; j < CH_NB; ++j) {
            const double g = readlane_f64(a[j], j);
            if (!(g > 0.0)) { fail_here = true; break; }  // (padding rows are the identity's: never here)
            const double dj = sqrt(g), inv = 1.0 / dj;
            rinv[j] = inv;
            if (lane == 0) rdiag[j] = inv;
            a[j] = row == j ? dj : a[j] * inv;
#pragma unroll
            for (int k = j + 1; k < CH_NB; ++k) a[k] -= a[j] * readlane_f64(a[j], k);  // (rows above k compute entries nobody reads)
        }
        if (fail_here) { if (lane == 0) bad = 1; }
        else {
            if (lane < CH_NB) {
#pragma unroll
                for (int k = 0; k < CH_NB; ++k)
                    if (k <= row) D[row][k] = a[k];
            }
            // the inverse of the factor the same way: lane c solves column c, row k of the factor comes by broadcast
            double di[CH_NB];
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) {
                double sdi = 0.0;
#pragma unroll
                for (int m = 0; m < k; ++m) sdi += readlane_f64(a[m], k) * di[m];  // (di[m] = 0 above the column's diagonal)
                di[k] = k < row ? 0.0 : (k == row ? rinv[k] : -sdi * rinv[k]);
            }
            if (lane < CH_NB) {
#pragma unroll
                for (int k = 0; k < CH_NB; ++k) Di_l[k * (CH_NB + 1) + row] = di[k];
            }
        }
    } else
    if (threadIdx.x < WAVE) {
        const int lane = threadIdx.x;
        for (int j = 0; j < nb; ++j) {
            const double g = D[j][j];
            if (!(g > 0.0)) {  // the same value in every lane
                if (lane == 0) bad = 1;
                break;
            }
            const double dj = sqrt(g), inv = 1.0 / dj;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { D[j][j] = dj; rdiag[j] = inv; }
            for (int i = j + 1 + lane; i < nb; i += WAVE) D[i][j] *= inv;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int i = j + 1 + (lane & 31);  // a row per lane of each half-wave, the halves take alternate columns
            if (i < nb) {
                const double lij = D[i][j];
                for (int k = j + 1 + (lane >> 5); k <= i; k += 2) D[i][k] -= lij * D[k][j];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        for (int k = nb + lane; k < CH_NB; k += WAVE) rdiag[k] = 1.0;  // identity padding
    }
    __syncthreads();
    if (bad) {
        if (threadIdx.x == 0) *flag = 1;
        return;
    }
    // the other workgroups of this launch read the unfactored diagonal block, so with more than one workgroup the factor
    // goes to a scratch tile and k_chol_update copies it in
    if (blockIdx.x == 0)
        for (int o = threadIdx.x; o < nb * nb; o += BLOCK) {
            const int i = o / nb, k = o % nb;
            if (k <= i) {
                if (gridDim.x == 1 && j0 + nb >= b) G[(size_t)(j0 + i) * b + j0 + k] = D[i][k];
                else Dfac[i * CH_NB + k] = D[i][k];
            }
        }
    // rows below the diagonal block: x = g D^-T, one thread per row, as a product with the explicit inverse of the 32 x 32 factor
    // (its rows read as LDS broadcasts, the row of g in registers, the output column in a rolled loop).  The substitution
    // form of the same triangle, fully unrolled, makes the compiler hoist 496 LDS reads: 512 VGPRs, 2.4 KB of scratch, 54 us.
    if (lds_form && threadIdx.x < CH_NB) {  // lane c solves column c of the inverse
        const int c = threadIdx.x;
        for (int k = 0; k < CH_NB; ++k) {
            double v = 0.0;
            if (k == c) v = rdiag[c];
            else if (k > c) {
                double s = 0.0;
                for (int m = c; m < k; ++m) s += D[k][m] * Di[m][c];
                v = -s * rdiag[k];
            }
            Di[k][c] = v;
        }
    }
    __syncthreads();
    if (Dinv && blockIdx.x == 0)
        for (int o = threadIdx.x; o < CH_NB * CH_NB; o += BLOCK) Dinv[(size_t)(j0 / CH_NB) * CH_NB * CH_NB + o] = Di[o / CH_NB][o % CH_NB];
    if (lds_form) {  // one thread per row: 32 loads and 32 stores of 8 bytes, each thread on a row of its own
        const int r = j0 + nb + blockIdx.x * BLOCK + threadIdx.x;
        if (r < b) {
            double gr[CH_NB];
            double* g = G + (size_t)r * b + j0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) gr[c] = g[c < nb ? c : nb - 1];  // the padding columns of D^-1 are the identity's: unused
#pragma unroll 1
            for (int k = 0; k < nb; ++k) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) s += gr[c] * Di[k][c];
                g[k] = s;
            }
        }
        return;
    }
    // (rows exist below the block only when the block is a full one: nb == CH_NB here.)  64 rows at a time through LDS -- a row's 32
    // values are one 256-byte run -- and X = G_rows D^-T on the fp64 matrix cores, wave w its 16 rows x 32 columns.
    __shared__ double Gt[64][CH_NB + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    for (int sub = 0; sub < BLOCK / 64; ++sub) {
        const int r0 = j0 + nb + blockIdx.x * BLOCK + sub * 64;
        if (r0 >= b) break;  // (uniform)
        for (int o = threadIdx.x; o < 64 * CH_NB; o += BLOCK) {
            const int rr = o / CH_NB, c = o % CH_NB;
            Gt[rr][c] = r0 + rr < b ? G[(size_t)(r0 + rr) * b + j0 + c] : 0.0;
        }
        __syncthreads();
        d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int kk = 0; kk < CH_NB; kk += 4) {
            const double a = Gt[16 * wv + li][kk + lk];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Di[16 * t + li][kk + lk], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 16 * wv + lk + 4 * q;
                if (r < b) G[(size_t)r * b + j0 + 16 * t + li] = acc[t][q];
            }
        __syncthreads();
    }
}
// trailing update after panel [j0, j0+nb): G[i][k] -= sum_c L[i][j0+c] L[k][j0+c] for j0+nb <= k <= i < b, 32 x 32 tiles
__global__ __launch_bounds__(BLOCK) void k_chol_update(int b, int j0, double* __restrict__ G, const int* __restrict__ flag,
                                                       const double* __restrict__ Dfac) {
    if (blockIdx.y > blockIdx.x || *flag) return;  // lower triangle of tiles only
    if (blockIdx.x == 0 && blockIdx.y == 0)  // the panel's factored diagonal block (see k_chol_panel)
        for (int o = threadIdx.x; o < CH_NB * CH_NB; o += BLOCK) {
            const int i = o / CH_NB, k = o % CH_NB;
            if (k <= i) G[(size_t)(j0 + i) * b + j0 + k] = Dfac[o];
        }
    __shared__ double A[CH_NB][CH_NB + 1], Bt[CH_NB][CH_NB + 1];
    const int j1 = j0 + CH_NB;
    const int i0 = j1 + blockIdx.x * CH_NB, k0 = j1 + blockIdx.y * CH_NB;
    for (int o = threadIdx.x; o < CH_NB * CH_NB; o += BLOCK) {
        const int r = o / CH_NB, c = o % CH_NB;
        A[r][c] = i0 + r < b ? G[(size_t)(i0 + r) * b + j0 + c] : 0.0;
        Bt[r][c] = k0 + r < b ? G[(size_t)(k0 + r) * b + j0 + c] : 0.0;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < CH_NB * CH_NB; o += BLOCK) {
        const int r = o / CH_NB, q = o % CH_NB;
        const int i = i0 + r, k = k0 + q;
        if (i < b && k <= i) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) s += A[r][c] * Bt[q][c];
            G[(size_t)i * b + k] -= s;
        }
    }
}
// ---- Vout = (V diag(dscale)) L^{-T} in column blocks of 32 on the fp64 matrix cores: X_J = (V_J D_J - X_{<J} L[J, <J]^T) Linv_JJ^T with the
// inverses of L's diagonal blocks from k_chol_panel.  A workgroup owns 64 rows and walks the blocks left to right (its rows depend on
// nobody else's); the part of X already computed comes back from Vout (as stored, in T) through LDS tiles of 64 columns.  k_trsm_rows
// below -- one wavefront per row, b dependent steps of a 64-lane sum, two broadcasts and a division -- took 0.13 ms at b = 105 and
// 0.89 ms at b = 444.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_trsm_blocked(int K, int b, int ld, const T* __restrict__ V, const double* __restrict__ L,
                                                       const double* __restrict__ dscale, const double* __restrict__ Dinv, T* __restrict__ Vout) {
    __shared__ double Xt[64][65], Lt[CH_NB][65];
    // T and the block's inverse take the tiles' places once the products over the earlier columns are done
    double (*Tt)[CH_NB + 1] = reinterpret_cast<double (*)[CH_NB + 1]>(&Xt[0][0]);
    double (*Dl)[CH_NB + 1] = reinterpret_cast<double (*)[CH_NB + 1]>(&Lt[0][0]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    const int r0 = blockIdx.x * 64;
    for (int j0 = 0; j0 < b; j0 += CH_NB) {
        d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
        for (int k0 = 0; k0 < j0; k0 += 64) {
            for (int o = threadIdx.x; o < 64 * 64; o += BLOCK) {
                const int rr = o >> 6, c = o & 63;
                Xt[rr][c] = (r0 + rr < K && k0 + c < j0) ? (double)Vout[(size_t)(r0 + rr) * ld + k0 + c] : 0.0;
            }
            for (int o = threadIdx.x; o < CH_NB * 64; o += BLOCK) {
                const int jr = o >> 6, c = o & 63;
                Lt[jr][c] = (j0 + jr < b && k0 + c < j0) ? L[(size_t)(j0 + jr) * b + k0 + c] : 0.0;
            }
            __syncthreads();
#pragma unroll 4
            for (int kk = 0; kk < 64; kk += 4) {
                const double a = Xt[16 * wv + li][kk + lk];
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Lt[16 * t + li][kk + lk], acc[t], 0, 0, 0);
            }
            __syncthreads();
        }
        // T = V_J D_J - (X L^T)_J, and the block's inverse
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * wv + lk + 4 * q, col = j0 + 16 * t + li;
                const double v = (r0 + i < K && col < b) ? (double)V[(size_t)(r0 + i) * ld + col] * dscale[col] : 0.0;
                Tt[i][16 * t + li] = v - acc[t][q];
            }
        for (int o = threadIdx.x; o < CH_NB * CH_NB; o += BLOCK) Dl[o / CH_NB][o % CH_NB] = Dinv[(size_t)(j0 / CH_NB) * CH_NB * CH_NB + o];
        __syncthreads();
        d4 xo[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int kk = 0; kk < CH_NB; kk += 4) {
            const double a = Tt[16 * wv + li][kk + lk];
#pragma unroll
            for (int t = 0; t < 2; ++t) xo[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Dl[16 * t + li][kk + lk], xo[t], 0, 0, 0);  // X[i][k] = sum_c T[i][c] Linv[k][c]
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * wv + lk + 4 * q, col = j0 + 16 * t + li;
                if (r0 + i < K && col < b) Vout[(size_t)(r0 + i) * ld + col] = (T)xo[t][q];
            }
        __threadfence_block();  // the next blocks read these columns back
        __syncthreads();
    }
    for (int o = threadIdx.x; o < 64 * (ld - b); o += BLOCK) {  // padding columns
        const int rr = o / (ld - b), c = b + o % (ld - b);
        if (r0 + rr < K) Vout[(size_t)(r0 + rr) * ld + c] = T(0);
    }
}
// ---- Vout[r,:] = (V[r,:] * dscale) L^{-T}: forward substitution per block row, one wavefront per row.
// x_j = (v_j d_j - sum_{i<j} x_i L[j][i]) / L[j][j]; the running x lives in LDS, row j of L is read coalesced.
// This is the orthonormalising right factor of Cholesky-QR applied without ever forming an inverse.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_trsm_rows(int K, int b, int ld, const T* __restrict__ V, const double* __restrict__ L,
                                                     const double* __restrict__ dscale, T* __restrict__ Vout) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xs = reinterpret_cast<double*>(smem_raw);  // [WAVES_PER_BLOCK][b]
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    double* x = xs + (size_t)wib * b;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < K; row += gridDim.x * WAVES_PER_BLOCK) {
        for (int j = 0; j < b; ++j) {
            double s = 0.0;
            const double* Lj = L + (size_t)j * b;
            for (int i = lane; i < j; i += WAVE) s += x[i] * Lj[i];
            s = wave_sum(s);
            if (lane == 0) x[j] = ((double)V[(size_t)row * ld + j] * dscale[j] - s) / Lj[j];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): x[j] is in LDS before the next step reads it
        }
        for (int j = lane; j < b; j += WAVE) Vout[(size_t)row * ld + j] = (T)x[j];
        for (int j = b + lane; j < ld; j += WAVE) Vout[(size_t)row * ld + j] = T(0);
    }
}
// off-diagonal Frobenius norm^2 and diagonal scale: out[0] = sum_{i != j} H_ij^2, out[1] = max |H_ii|
__global__ __launch_bounds__(1024) void k_offdiag(int b, const double* __restrict__ H, double* __restrict__ out) {
    __shared__ double sh[2][16];
    double s = 0.0, d = 0.0;
    for (int i = threadIdx.x >> 6; i < b; i += 16) {  // a wave per row: coalesced, no division
        const double* row = H + (size_t)i * b;
        for (int j = threadIdx.x & 63; j < b; j += WAVE) {
            const double h = row[j];
            if (i != j) s += h * h;
            else d = fabs(h) > d ? fabs(h) : d;
        }
    }
    s = wave_sum(s);
    d = wave_max(d);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, td = 0.0;
        for (int w = 0; w < 16; ++w) { ts += sh[0][w]; td = sh[1][w] > td ? sh[1][w] : td; }
        out[0] = ts;
        out[1] = td;
    }
}
__global__ __launch_bounds__(BLOCK) void k_set_eye(int b, double* __restrict__ Q) {
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < b * b; o += gridDim.x * BLOCK) Q[o] = (o / b == o % b) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(BLOCK) void k_get_diag(int b, const double* __restrict__ H, double* __restrict__ d) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < b; i += gridDim.x * BLOCK) d[i] = H[(size_t)i * b + i];
}
// Qout[:, c] = Q[:, perm[c]] * scale[c]   (column selection / permutation / scaling of a b x b matrix -> b x n)
__global__ __launch_bounds__(BLOCK) void k_select_cols(int b, int n, const double* __restrict__ Q, const int* __restrict__ perm,
                                                       const double* __restrict__ scale, double* __restrict__ Qout) {
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < b * n; o += gridDim.x * BLOCK) {
        const int i = o / n, c = o % n;
        Qout[o] = Q[(size_t)i * b + perm[c]] * scale[c];
    }
}
// G'_{ij} = d_i G_ij d_j with d = 1/sqrt(diag G) (unit-diagonal scaling before the orthonormalising eigensolve)
__global__ __launch_bounds__(BLOCK) void k_scale_sym(int b, double* __restrict__ G, double* __restrict__ dscale) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < b; i += gridDim.x * BLOCK) {
        const double g = G[(size_t)i * b + i];
        dscale[i] = g > 0.0 ? 1.0 / sqrt(g) : 0.0;
    }
}
__global__ __launch_bounds__(BLOCK) void k_apply_scale_sym(int b, double* __restrict__ G, const double* __restrict__ dscale) {
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < b * b; o += gridDim.x * BLOCK) G[o] *= dscale[o / b] * dscale[o % b];
}
// Qout = diag(dscale) * Q * diag(1/sqrt(max(lambda, floor)))   -> the orthonormalising right factor
__global__ __launch_bounds__(BLOCK) void k_orth_factor(int b, const double* __restrict__ Q, const double* __restrict__ dscale,
                                                       const double* __restrict__ lam, double floor_rel, double lam_max,
                                                       double* __restrict__ Qout) {
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < b * b; o += gridDim.x * BLOCK) {
        const int i = o / b, j = o % b;
        double l = lam[j];
        const double fl = floor_rel * lam_max;
        if (!(l > fl)) l = fl;
        Qout[o] = dscale[i] * Q[o] / sqrt(l);
    }
}
// partial[block][c] = sum over the block's rows of (W[r,c] - theta[c] V[r,c])^2
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_resid_colsq(int K, int b, int ld, const T* __restrict__ W, const T* __restrict__ V,
                                                       const double* __restrict__ theta, double* __restrict__ partial) {
    for (int c = threadIdx.x; c < b; c += BLOCK) {
        double s = 0.0;
        const double th = theta[c];
        for (int r = blockIdx.x; r < K; r += gridDim.x) {
            const double d = (double)W[(size_t)r * ld + c] - th * (double)V[(size_t)r * ld + c];
            s += d * d;
        }
        partial[(size_t)blockIdx.x * b + c] = s;
    }
}
// Out = c1 * X + c2 * Y  (block axpby)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_axpby(size_t n, double c1, const T* __restrict__ X, double c2, const T* __restrict__ Y,
                                                 T* __restrict__ Out) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK)
        Out[o] = (T)(c1 * (double)X[o] + c2 * (double)Y[o]);
}

}  // namespace mmw

namespace mmw {
// out[r, c] = V[r, sel[c]] * sc[c]   (K x rank float64, row-major)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_export_factor(int K, int rank, int ld, const T* __restrict__ V, const int* __restrict__ sel,
                                                         const double* __restrict__ sc, double* __restrict__ out) {
    const size_t n = (size_t)K * rank;
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) {
        const int c = (int)(o % rank);
        const size_t r = o / rank;
        out[o] = (double)V[r * ld + sel[c]] * sc[c];
    }
}
// dst[i] = src[map[i]] (0 where map[i] < 0): CSR-ordered values into the blocked traversal order
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gather_blocked(size_t n, const int* __restrict__ map, const T* __restrict__ src, T* __restrict__ dst) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) {
        const int m = map[o];
        dst[o] = m >= 0 ? src[m] : T(0);
    }
}
// dst = src * s
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_scaled_copy(size_t n, const T* __restrict__ src, double s, T* __restrict__ dst) {
    for (size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x; o < n; o += (size_t)gridDim.x * BLOCK) dst[o] = (T)((double)src[o] * s);
}
// scal = { sum Y_D, sum Y_F, sum_k cH_k Y_H,k / norm_H,k , total } for an arbitrary weight vector Y (used at Ybar)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_ysums(int K, int E_asso, const T* __restrict__ Y, const T* __restrict__ cH,
                                                 const T* __restrict__ inv_norm_H, double* __restrict__ scal) {
    __shared__ double sh[WAVES_PER_BLOCK];
    double sD = 0, sF = 0, sW = 0;
    const int C = E_asso + 2 * K, baseH = K + E_asso;
    for (int c = threadIdx.x; c < C; c += BLOCK) {
        const double y = (double)Y[c];
        if (c < K) sD += y;
        else if (c < baseH) sF += y;
        else sW += (double)cH[c - baseH] * y * (double)inv_norm_H[c - baseH];
    }
    sD = block_sum(sD, sh);
    sF = block_sum(sF, sh);
    sW = block_sum(sW, sh);
    if (threadIdx.x == 0) {
        scal[0] = sD; scal[1] = sF; scal[2] = sW; scal[3] = 1.0;
    }
}
// single value: max over a partial slab
__global__ __launch_bounds__(BLOCK) void k_max_reduce(int n, const double* __restrict__ part, double* __restrict__ out) {
    __shared__ double sh[WAVES_PER_BLOCK];
    double m = -1e300;
    for (int i = threadIdx.x; i < n; i += BLOCK) m = part[i] > m ? part[i] : m;
    m = block_max(m, sh);
    if (threadIdx.x == 0) out[0] = m;
}
// ---- second pass of the orthonormalisation when the block is already nearly orthonormal: G = V^T V = I + E with a small E, and
// (I + E)^{-1/2} = I - E/2 + O(E^2).  out[0] = ||G - I||_F^2 (one workgroup); M = 1.5 I - 0.5 G.
__global__ __launch_bounds__(1024) void k_dev_from_identity(int b, const double* __restrict__ G, double* __restrict__ out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int i = threadIdx.x >> 6; i < b; i += 16) {
        const double* row = G + (size_t)i * b;
        for (int j = threadIdx.x & 63; j < b; j += WAVE) {
            const double e = row[j] - (i == j ? 1.0 : 0.0);
            s += e * e;
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}
__global__ __launch_bounds__(BLOCK) void k_ns_first(int b, const double* __restrict__ G, double* __restrict__ M) {
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < b * b; o += gridDim.x * BLOCK) M[o] = ((o / b == o % b) ? 1.5 : 0.0) - 0.5 * G[o];
}
// out[0] = max_i a[i] (one workgroup; the objective record of a solve)
template <typename T> __global__ __launch_bounds__(1024) void k_max_of(size_t n, const T* __restrict__ a, double* __restrict__ out) {
    __shared__ double sh[16];
    double m = -1e300;
    // sixteen independent elements per thread and round (one dependent 4-byte load per round took 27 us for 84 k entries: the read-back
    // of the objective record was 5 % of a 20-step timed region)
    constexpr int U = 16;
    const size_t nu = n - n % ((size_t)1024 * U);
    for (size_t i = threadIdx.x; i < nu; i += (size_t)1024 * U) {
        T x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) x[k] = a[i + (size_t)k * 1024];
#pragma unroll
        for (int k = 0; k < U; ++k) m = (double)x[k] > m ? (double)x[k] : m;
    }
    for (size_t i = nu + threadIdx.x; i < n; i += 1024) m = (double)a[i] > m ? (double)a[i] : m;
    m = block_max(m, sh);
    if (threadIdx.x == 0) out[0] = m;
}
}  // namespace mmw
