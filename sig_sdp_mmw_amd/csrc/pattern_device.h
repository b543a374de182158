// State processing ON THE DEVICE for a handle created straight from the device problem generator (mmw_create_from_env,
// SURVEY.md §8 f2): mmw._process_state and the prologue of mmw._run (sim_src/alg/mmw.py:26-41, 46-74) without the host round trip.
//
// The generator (env_device.h) holds the K x A receive powers rx, the association and the members of every AP.  With
//     S_gain[k][j] = thr0(rx[k][asso[j]])                       (env.py:190-192; thr0 drops what is below the threshold)
// everything the host build of pattern.h derives from the CSR arrays follows row by row from rx itself:
//     S_T'[a][j] = thr0(rx[j][asso[a]])   for j != a, asso[j] != asso[a]      (transpose, association mask, diagonal: mmw.py:28-33)
//     Q[a][j]    = 1                      for j != a, asso[j] == asso[a]      (env.py:181-189)
//     L / X row a = {a} + {j : S_T'[a][j] != 0 or S_T'[j][a] != 0} + Q row a  (mmw.py:52-57, 144-194)
// One wavefront per row scans the users in ascending order, 64 at a time, and compacts the members of each list with ballots:
// rows come out sorted, there is no sort, no transpose of a sparse matrix and no atomic.  rx is read once by rows (rx[a][asso[j]],
// a gather inside one 6 KB row) and once by columns through a dense transpose rxT made at the start (coalesced along j).
// Pair ids (triu-CSR order of Q, mmw.py:57) come from a user's position in its AP's sorted member list; mirrors by a binary search
// in the (sorted) row of the column; S_sum / the squared row sums by one thread per row in ascending column order with separately
// rounded products and sums -- the order and the roundings of the host build, so the values are bit-identical to it.
#pragma once
#include "device_utils.h"
#include "runtime.h"

namespace mmw {

__device__ __forceinline__ double pat_thr0(double v, double thr) { return v < thr ? 0.0 : v; }

// rxT[a][k] = rx[k][a] (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void k_pat_transpose(int K, int A, const double* __restrict__ rx, double* __restrict__ rxT) {
    __shared__ double tile[32][33];
    const int k0 = blockIdx.x * 32, a0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, a = a0 + tx;
        tile[r][tx] = (k < K && a < A) ? rx[(size_t)k * A + a] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, k = k0 + tx;
        if (a < A && k < K) rxT[(size_t)a * K + k] = tile[tx][r];
    }
}
// position of every user in its AP's ascending member list
__global__ __launch_bounds__(BLOCK) void k_pat_appos(int A, const int* __restrict__ ap_ptr, const int* __restrict__ ap_mem, int* __restrict__ pos) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    for (int a = blockIdx.x * WAVES_PER_BLOCK + wib; a < A; a += gridDim.x * WAVES_PER_BLOCK)
        for (int i = ap_ptr[a] + lane; i < ap_ptr[a + 1]; i += WAVE) pos[ap_mem[i]] = i - ap_ptr[a];
}

// what a lane sees of (row a, user j)
struct PatCell {
    double vab, vba;  // S_T'[a][j], S_T'[j][a]
    bool same;        // same AP (j == a included)
    bool inL;
};
__device__ __forceinline__ PatCell pat_cell(int a, int aa, int j, int K, int A, double thr, const double* __restrict__ rx, const double* __restrict__ rxT,
                                            const int* __restrict__ asso) {
    PatCell c{0.0, 0.0, false, false};
    if (j >= K) return c;
    const int aj = asso[j];
    c.same = aj == aa;
    if (!c.same) {
        c.vab = pat_thr0(rxT[(size_t)aa * K + j], thr);
        c.vba = pat_thr0(rx[(size_t)a * A + aj], thr);
    }
    c.inL = c.same || c.vab != 0.0 || c.vba != 0.0;
    return c;
}

// pass 1: row lengths {L pattern, S_T', upper-triangular gain edges} and whether S_gain stores the row's diagonal; also the stored
// off-diagonal length of S + S^T and of Q, which the bisection's bounds take (binary_search_relaxation.py:13-29)
__global__ __launch_bounds__(BLOCK) void k_pat_count(int K, int A, double thr, const double* __restrict__ rx, const double* __restrict__ rxT,
                                                     const int* __restrict__ asso, const int* __restrict__ ap_cnt, const int* __restrict__ appos,
                                                     int* __restrict__ nL, int* __restrict__ nST, int* __restrict__ nGU, int* __restrict__ nQU,
                                                     int* __restrict__ sdiag, int* __restrict__ nSym) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    for (int a = blockIdx.x * WAVES_PER_BLOCK + wib; a < K; a += gridDim.x * WAVES_PER_BLOCK) {
        const int aa = asso[a];
        int cL = 0, cS = 0, cG = 0, cY = 0;
        for (int j0 = 0; j0 < K; j0 += WAVE) {
            const int j = j0 + lane;
            const PatCell c = pat_cell(a, aa, j, K, A, thr, rx, rxT, asso);
            cL += __popcll(__ballot(c.inL));
            cS += __popcll(__ballot(c.vab != 0.0));
            cG += __popcll(__ballot(j > a && (c.vab != 0.0 || c.vba != 0.0)));
            // S + S^T off the diagonal: S itself also holds the same-AP users an AP hears (rx[a][aa] is one value for all of them)
            bool sym = false;
            if (j < K && j != a) {
                if (c.same) sym = pat_thr0(rx[(size_t)a * A + aa], thr) != 0.0 || pat_thr0(rxT[(size_t)aa * K + j], thr) != 0.0;
                else sym = c.vab != 0.0 || c.vba != 0.0;
            }
            cY += __popcll(__ballot(sym));
        }
        if (lane == 0) {
            nL[a] = cL; nST[a] = cS; nGU[a] = cG; nSym[a] = cY;
            nQU[a] = ap_cnt[aa] - appos[a] - 1;  // same-AP users above a: row a of triu(Q, 1)
            sdiag[a] = pat_thr0(rx[(size_t)a * A + aa], thr) != 0.0 ? 1 : 0;
        }
    }
}

template <typename T> struct PatOut {
    const int* l_ptr; const int* st_ptr; const int* gu_ptr; const int* qu_ptr;  // exclusive prefix sums of the counts
    const int* appos;
    int* l_idx; int* lrow; T* sab; T* sba; int* pid; int* diag_pos;
    int* st_idx; double* st_val;
    int* gain_x; int* gain_y; int* asso_x; int* asso_y; int* asso_pos;
};
// pass 2: the rows themselves
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_pat_fill(int K, int A, double thr, const double* __restrict__ rx, const double* __restrict__ rxT,
                                                    const int* __restrict__ asso, PatOut<T> O) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int a = blockIdx.x * WAVES_PER_BLOCK + wib; a < K; a += gridDim.x * WAVES_PER_BLOCK) {
        const int aa = asso[a], pa = O.appos[a];
        int oL = O.l_ptr[a], oS = O.st_ptr[a], oG = O.gu_ptr[a];
        for (int j0 = 0; j0 < K; j0 += WAVE) {
            const int j = j0 + lane;
            const PatCell c = pat_cell(a, aa, j, K, A, thr, rx, rxT, asso);
            const unsigned long long mL = __ballot(c.inL), mS = __ballot(c.vab != 0.0);
            const bool up = j > a && (c.vab != 0.0 || c.vba != 0.0);
            const unsigned long long mG = __ballot(up);
            if (c.inL) {
                const int e = oL + __popcll(mL & below);
                O.l_idx[e] = j;
                O.lrow[e] = a;
                O.sab[e] = (T)c.vab;
                O.sba[e] = (T)c.vba;
                int id = -1;
                if (c.same && j != a) {  // pair (lo, hi) in the triu-CSR order of Q: row lo, its same-AP users above it in ascending order
                    const int pj = O.appos[j];
                    const int lo = j > a ? a : j, plo = j > a ? pa : pj, phi = j > a ? pj : pa;
                    id = O.qu_ptr[lo] + (phi - plo - 1);
                    if (j > a) { O.asso_pos[id] = e; O.asso_x[id] = a; O.asso_y[id] = j; }
                }
                O.pid[e] = id;
                if (j == a) O.diag_pos[a] = e;
            }
            if (c.vab != 0.0) {
                const int e = oS + __popcll(mS & below);
                O.st_idx[e] = j;
                O.st_val[e] = c.vab;
            }
            if (up) {
                const int e = oG + __popcll(mG & below);
                O.gain_x[e] = a;
                O.gain_y[e] = j;
            }
            oL += __popcll(mL); oS += __popcll(mS); oG += __popcll(mG);
        }
    }
}
// mirror[e] = position of (column, row): the pattern is symmetric and its rows are sorted
__global__ __launch_bounds__(BLOCK) void k_pat_mirror(size_t nnz, const int* __restrict__ l_ptr, const int* __restrict__ l_idx, const int* __restrict__ lrow,
                                                      int* __restrict__ mirror) {
    for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < nnz; e += (size_t)gridDim.x * BLOCK) {
        const int a = lrow[e], c = l_idx[e];
        int lo = l_ptr[c], hi = l_ptr[c + 1];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (l_idx[mid] < a) lo = mid + 1;
            else hi = mid;
        }
        mirror[e] = lo;
    }
}
// S_sum and the squared row sums of S_T' in ascending column order, products and sums rounded separately (mmw.py:34,37-39 as the host
// build of pattern.h evaluates them)
__global__ __launch_bounds__(BLOCK) void k_pat_rowstats(int K, const int* __restrict__ st_ptr, const double* __restrict__ st_val, double* __restrict__ s_sum,
                                                        double* __restrict__ sq_sum) {
#pragma clang fp contract(off)  // (HIP's __dmul_rn / __dadd_rn are plain operators on this target: a fused multiply-add would differ from the host's last bit)
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < K; k += gridDim.x * BLOCK) {
        double s = 0.0, q = 0.0;
        for (int i = st_ptr[k]; i < st_ptr[k + 1]; ++i) {
            const double v = st_val[i];
            const double vv = v * v;
            s = s + v;
            q = q + vv;
        }
        s_sum[k] = s;
        sq_sum[k] = q;
    }
}
// the rounding's view of the state (sdp_solver.py:36-41): S_gain without its diagonal, and h_max of every stored column
__global__ __launch_bounds__(BLOCK) void k_pat_so_fill(int K, const int* __restrict__ s_ptr, const int* __restrict__ s_idx, const double* __restrict__ s_val,
                                                       const int* __restrict__ so_ptr, const double* __restrict__ h_max, int* __restrict__ so_idx,
                                                       double* __restrict__ so_val, double* __restrict__ so_hmax) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k = blockIdx.x * WAVES_PER_BLOCK + wib; k < K; k += gridDim.x * WAVES_PER_BLOCK) {
        int o = so_ptr[k];
        for (int i0 = s_ptr[k]; i0 < s_ptr[k + 1]; i0 += WAVE) {
            const int i = i0 + lane;
            const bool on = i < s_ptr[k + 1] && s_idx[i] != k;
            const unsigned long long m = __ballot(on);
            if (on) {
                const int e = o + __popcll(m & below);
                so_idx[e] = s_idx[i];
                so_val[e] = s_val[i];
                so_hmax[e] = h_max[s_idx[i]];
            }
            o += __popcll(m);
        }
    }
}

}  // namespace mmw
