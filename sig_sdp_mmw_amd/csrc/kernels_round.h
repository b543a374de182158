// Device kernels for the randomized vector rounding, sdp_solver.rounding_one_attempt
// (sim_src/alg/sdp_solver.py:27-107): user visiting order, the Gaussian projection inprod = randv gX^T
// on the fp64 matrix cores (the one dense contraction of the path), per-user slot preference order and
// the greedy feasible slot assignment.  All arithmetic in float64: the outputs are integers that must
// agree exactly with the reference on identical inputs.
#pragma once
#include "device_utils.h"

namespace mmw {

// ---- ||gX_k||_2 per user (sdp_solver.py:51) ---------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_row_norms_f64(int K, int Dp, const double* __restrict__ gX, double* __restrict__ nrm) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wib; row < K; row += gridDim.x * WAVES_PER_BLOCK) {
        double s = 0.0;
        for (int c = lane; c < Dp; c += WAVE) {
            const double x = gX[(size_t)row * Dp + c];
            s += x * x;
        }
        s = wave_sum(s);
        if (lane == 0) nrm[row] = sqrt(s);
    }
}

// ---- order = argsort(-key): rank by counting, ties by lower index (stable) --------------------
// rank[k] = #{ j : key[j] > key[k]  or (key[j] == key[k] and j < k) };  order[rank[k]] = k
// From the whole chip: the K^2 comparisons are split into RANK_SPLIT groups of key tiles (grid.y), every workgroup leaves its partial
// counts, and the second launch adds them (fixed order, integers) and scatters.  (One workgroup per 256 keys walking all K keys -- 40
// workgroups at K = 10 003 -- took 0.70 ms, a tenth of a rounding call.)
constexpr int RANK_SPLIT = 16;
__global__ __launch_bounds__(BLOCK) void k_rank_count(int n, const double* __restrict__ key, int* __restrict__ part /* [RANK_SPLIT][n] */) {
    __shared__ double tile[BLOCK];
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    const double mine = k < n ? key[k] : 0.0;
    const int ntiles = (n + BLOCK - 1) / BLOCK, per = (ntiles + RANK_SPLIT - 1) / RANK_SPLIT;
    const int t_beg = (int)blockIdx.y * per, t_end = min(ntiles, t_beg + per);
    int r = 0;
    for (int tl = t_beg; tl < t_end; ++tl) {
        const int j0 = tl * BLOCK, j = j0 + (int)threadIdx.x;
        tile[threadIdx.x] = j < n ? key[j] : 0.0;
        __syncthreads();
        const int lim = n - j0 < BLOCK ? n - j0 : BLOCK;
        if (lim == BLOCK) {
#pragma unroll 16
            for (int t = 0; t < BLOCK; ++t) {
                const double o = tile[t];
                r += (o > mine) || (o == mine && (j0 + t) < k);
            }
        } else
        for (int t = 0; t < lim; ++t) {
            const double o = tile[t];
            r += (o > mine) || (o == mine && (j0 + t) < k);
        }
        __syncthreads();
    }
    if (k < n) part[(size_t)blockIdx.y * n + k] = r;
}
__global__ __launch_bounds__(BLOCK) void k_rank_scatter(int n, const int* __restrict__ part, int* __restrict__ order) {
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    int r = 0;
#pragma unroll
    for (int s = 0; s < RANK_SPLIT; ++s) r += part[(size_t)s * n + k];
    order[r] = k;
}

// ---- projection on the fp64 matrix cores -------------------------------------------------------
// P[b][k][z] = sum_d randv[b][z][d] * gX[k][d].  One wavefront owns a 16(z) x 16(users) tile and walks
// the D' dimension 4 at a time with v_mfma_f64_16x16x4_f64: lane l feeds A[i = l&15][kk = l>>4] and
// B[kk = l>>4][j = l&15]; its 4 results are rows (l>>4) + 4r, column l&15 of the tile.
// The 16 x DT slabs of both operands are staged through LDS with coalesced row reads (DT = 64 columns).
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int PROJ_DT = 64;
__global__ __launch_bounds__(BLOCK) void k_project_mfma(int K, int Z, int Dp, const double* __restrict__ gX,
                                                        const double* __restrict__ randv, double* __restrict__ P) {
    // block = 4 waves: 16 slots x 64 users; blockIdx.x -> user tile, blockIdx.y -> slot tile, blockIdx.z -> batch
    __shared__ double sA[16][PROJ_DT + 1];       // randv rows z0..z0+15
    __shared__ double sB[64][PROJ_DT + 1];       // gX rows u0..u0+63
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int u0 = blockIdx.x * 64, z0 = blockIdx.y * 16, b = blockIdx.z;
    const double* R = randv + (size_t)b * Z * Dp;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int i = lane & 15, kk = lane >> 4;
    for (int d0 = 0; d0 < Dp; d0 += PROJ_DT) {
        for (int t = threadIdx.x; t < 16 * PROJ_DT; t += BLOCK) {
            const int r = t / PROJ_DT, c = t % PROJ_DT;
            sA[r][c] = (z0 + r < Z && d0 + c < Dp) ? R[(size_t)(z0 + r) * Dp + d0 + c] : 0.0;
        }
        for (int t = threadIdx.x; t < 64 * PROJ_DT; t += BLOCK) {
            const int r = t / PROJ_DT, c = t % PROJ_DT;
            sB[r][c] = (u0 + r < K && d0 + c < Dp) ? gX[(size_t)(u0 + r) * Dp + d0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int d = 0; d < PROJ_DT; d += 4) {
            const double a = sA[i][d + kk];
            const double bb = sB[wib * 16 + i][d + kk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int user = u0 + wib * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int z = z0 + (lane >> 4) + 4 * r;
        if (user < K && z < Z) P[((size_t)b * K + user) * Z + z] = acc[r];
    }
}

// ---- per-user preference order: pref[b][k][rank of slot z] = z, descending inprod, ties by lower z
__global__ __launch_bounds__(BLOCK) void k_slot_pref(int K, int Z, const double* __restrict__ P, int* __restrict__ pref) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* row = reinterpret_cast<double*>(smem_raw);  // [Z]
    const size_t base = ((size_t)blockIdx.y * K + blockIdx.x) * Z;
    for (int z = threadIdx.x; z < Z; z += BLOCK) row[z] = P[base + z];
    __syncthreads();
    for (int z = threadIdx.x; z < Z; z += BLOCK) {
        const double mine = row[z];
        int r = 0;
        for (int j = 0; j < Z; ++j) r += (row[j] > mine) || (row[j] == mine && j < z);
        pref[base + r] = z;
    }
}

// ---- greedy feasible assignment (sdp_solver.py:70-101): one workgroup per attempt ---------------
// Users are visited in `order`; user k takes the first slot of its preference list where
//   (a) the interference already accumulated at k stays within h_max[k],
//   (b) adding k's emission keeps every current member n of the slot that k reaches within h_max[n],
//   (c) no current member shares an access point with k.
// gain_sum[n][z] accumulates S[k'][n] over the members k' of slot z in assignment order, exactly like
// the reference's dense row adds (sdp_solver.py:94) restricted to the nonzeros (user-major layout: the
// Z sums of one user are contiguous).
// The loop is sequential in the users, so its speed is the length of the dependent-load chain per user.
// Everything that does not depend on earlier assignments (the user's id, neighbour lists, gains, thresholds,
// preference row) is fetched one user ahead into a double-buffered LDS record; what remains on the chain is
// slot[n] (LDS when K fits) -> gain_sum[n][slot[n]].
struct GreedyLds {          // one prefetched user
    int k, deg, qdeg, pad;
};
// one header per processing position (shared by the attempts of a batch): everything needed to address user order[kk]'s
// static data without a chain of dependent loads
struct GreedyHdr {
    int k, sb, deg, qb;
    int qdeg, pad;
    double hk;
};
__global__ __launch_bounds__(BLOCK) void k_greedy_headers(int K, const int* __restrict__ order, const int* __restrict__ so_indptr,
                                                          const int* __restrict__ q_indptr, const double* __restrict__ h_max,
                                                          GreedyHdr* __restrict__ hdr) {
    for (int kk = blockIdx.x * BLOCK + threadIdx.x; kk < K; kk += gridDim.x * BLOCK) {
        const int k = order[kk];
        GreedyHdr h;
        h.k = k; h.sb = so_indptr[k]; h.deg = so_indptr[k + 1] - h.sb; h.qb = q_indptr[k]; h.qdeg = q_indptr[k + 1] - h.qb; h.pad = 0;
        h.hk = h_max[k];
        hdr[kk] = h;
    }
}
// workgroup barrier that orders LDS traffic only: global loads issued earlier (the prefetch of the next user) stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool SLOT_LDS>
__global__ __launch_bounds__(BLOCK) void k_greedy(int K, int Z, int maxdeg, int maxq, const GreedyHdr* __restrict__ hdr,
                                                  const int* __restrict__ pref_all, const int* __restrict__ so_indices,
                                                  const double* __restrict__ so_data, const double* __restrict__ so_hmax,
                                                  const int* __restrict__ q_indices, double* __restrict__ gain_all,
                                                  int* __restrict__ slot_all, int* __restrict__ rem) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // layout: [2] x { nid[maxdeg] int, nval[maxdeg] f64, nh[maxdeg] f64, qid[maxq] int, pref[Z] int }, bad[Z] int, slot[K] int (optional)
    char* sp = smem_raw;
    double* nval[2]; double* nh[2]; int* nid[2]; int* qid[2]; int* prf[2];
    for (int s2 = 0; s2 < 2; ++s2) { nval[s2] = reinterpret_cast<double*>(sp); sp += (size_t)maxdeg * 8; nh[s2] = reinterpret_cast<double*>(sp); sp += (size_t)maxdeg * 8; }
    for (int s2 = 0; s2 < 2; ++s2) { nid[s2] = reinterpret_cast<int*>(sp); sp += (size_t)maxdeg * 4; qid[s2] = reinterpret_cast<int*>(sp); sp += (size_t)maxq * 4;
                                     prf[s2] = reinterpret_cast<int*>(sp); sp += (size_t)Z * 4; }
    int* bad = reinterpret_cast<int*>(sp); sp += (size_t)Z * 4;
    int* slot_l = reinterpret_cast<int*>(sp);
    __shared__ GreedyLds rec[2];
    __shared__ double hk[2];
    __shared__ int best;
    __shared__ int unassigned;
    const int b = blockIdx.x;
    const int* pref = pref_all + (size_t)b * K * Z;
    double* gain = gain_all + (size_t)b * K * Z;   // [K][Z]
    int* slot_g = slot_all + (size_t)b * K;
    int* slot = SLOT_LDS ? slot_l : slot_g;
    if (SLOT_LDS)
        for (int i = threadIdx.x; i < K; i += BLOCK) slot_l[i] = -1;
    if (threadIdx.x == 0) unassigned = 0;
    // Software pipeline over the users, none of its loads depends on another load of the same step:
    //   step kk   requests the header of user kk+2 (h2) and, with the header of user kk+1 that arrived during step kk-1 (h1),
    //             that user's neighbour ids, gains, thresholds (so_hmax: h_max of the neighbour, per edge), access-point
    //             peers and preference row; they are written to the other LDS record at the end of the step (commit).
    // The barriers inside a step order LDS only (lds_barrier), so these requests stay in flight; the barrier that ends the
    // step is a full one: it publishes the step's additions to gain[] (global) before the next user's sums are read.
    constexpr int NE = 4;  // neighbour elements per thread: maxdeg <= NE * BLOCK
    GreedyHdr h1 = hdr[0], h2 = hdr[K > 1 ? 1 : 0];
    int r_n[NE], r_q[NE], r_p[NE];
    double r_v[NE], r_h[NE];
    auto issue = [&](const GreedyHdr& h) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * BLOCK;
            r_n[i] = e < h.deg ? so_indices[h.sb + e] : 0;
            r_v[i] = e < h.deg ? so_data[h.sb + e] : 0.0;
            r_h[i] = e < h.deg ? so_hmax[h.sb + e] : 0.0;
            r_q[i] = e < h.qdeg ? q_indices[h.qb + e] : 0;
            r_p[i] = e < Z ? pref[(size_t)h.k * Z + e] : 0;
        }
    };
    auto commit = [&](int s2, const GreedyHdr& h) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * BLOCK;
            if (e < h.deg) { nid[s2][e] = r_n[i]; nval[s2][e] = r_v[i]; nh[s2][e] = r_h[i]; }
            if (e < h.qdeg) qid[s2][e] = r_q[i];
            if (e < Z) prf[s2][e] = r_p[i];
        }
        if (threadIdx.x == 0) {
            rec[s2].k = h.k; rec[s2].deg = h.deg; rec[s2].qdeg = h.qdeg;
            hk[s2] = h.hk;
        }
    };
    issue(h1);
    commit(0, h1);
    h1 = h2;                                  // header of user 1
    if (K > 2) h2 = hdr[2];
    __syncthreads();
    for (int kk = 0; kk < K; ++kk) {
        const int cur = kk & 1;
        const bool more = kk + 1 < K;
        const GreedyHdr hn = h1;               // user kk+1 (arrived during the previous step)
        const int k = rec[cur].k, deg = rec[cur].deg, qdeg = rec[cur].qdeg;
        const double hmk = hk[cur];
        // the sums on the chain, requested together: the user's own row and, through slot[], each neighbour's sum in its slot
        double g_self[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int z = threadIdx.x + i * BLOCK;
            g_self[i] = z < Z ? gain[(size_t)k * Z + z] : 0.0;
        }
        int zn_e[NE];
        double g_ne[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * BLOCK;
            zn_e[i] = -1;
            g_ne[i] = 0.0;
            if (e < deg) {
                zn_e[i] = slot[nid[cur][e]];
                if (zn_e[i] >= 0) g_ne[i] = gain[(size_t)nid[cur][e] * Z + zn_e[i]];
            }
        }
        // the prefetch is requested AFTER the chain loads: loads return in order, so waiting for the chain loads leaves every
        // younger request in flight
        if (more) issue(hn);
        h1 = h2;
        if (kk + 3 < K) h2 = hdr[kk + 3];      // two steps ahead
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int z = threadIdx.x + i * BLOCK;
            if (z < Z) bad[z] = g_self[i] > hmk ? 1 : 0;
        }
        if (threadIdx.x == 0) best = Z;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * BLOCK;
            if (e < deg && zn_e[i] >= 0 && g_ne[i] + nval[cur][e] > nh[cur][e]) bad[zn_e[i]] = 1;
        }
        for (int e = threadIdx.x; e < qdeg; e += BLOCK) {
            const int zn = slot[qid[cur][e]];
            if (zn >= 0) bad[zn] = 1;
        }
        lds_barrier();
        for (int zz = threadIdx.x; zz < Z; zz += BLOCK)
            if (!bad[prf[cur][zz]]) {
                atomicMin(&best, zz);
                break;  // this thread's later candidates are worse
            }
        lds_barrier();
        const int zz = best;
        if (zz < Z) {
            const int z = prf[cur][zz];
            // fire-and-forget f64 atomic adds: one add per address and step, the steps separated by the full barrier below,
            // so the sums are formed in the reference's order (sdp_solver.py:94) without waiting for a load
            for (int e = threadIdx.x; e < deg; e += BLOCK) unsafeAtomicAdd(&gain[(size_t)nid[cur][e] * Z + z], nval[cur][e]);
            if (threadIdx.x == 0) {
                slot[k] = z;
                if (SLOT_LDS) slot_g[k] = z;
            }
        } else if (threadIdx.x == 0) {
            unassigned++;
        }
        if (more) commit(cur ^ 1, hn);
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) rem[b] = unassigned;
}


// ---- the same greedy pass, several users per step ------------------------------------------------------------------------
// The loop above is sequential in the users because a user's decision reads what earlier users wrote: slot[] of its out- and
// access-point neighbours, its own row of gain sums and the sums of its neighbours.  Two users a, b touch disjoint state -- and
// can be decided in the same step with the reference's result -- unless a is an out-neighbour of b or b of a, they share an access
// point, or they have a common out-neighbour.  The visiting order is by descending ||gX_k||, unrelated to the geometry, so on an
// interference graph consecutive users rarely interact (~4 % of pairs at the benchmark).  k_greedy_conflicts finds, for every
// position of the order, the nearest earlier position within the window that it interacts with; k_greedy_b then takes, per step,
// the longest run of positions free of such pairs (up to GB_WAVES), one wavefront per user: every check, the preference scan and
// the sum updates of a user are wave-level (no workgroup barrier inside a step), and one full barrier per step publishes the
// step's slots and sums.  The sums are still formed one add per address and step, in assignment order (sdp_solver.py:94).
constexpr int GB_WAVES = 8;
constexpr int GB_NE = 4;   // neighbour chunks (64 each) kept in registers one step ahead; longer lists are re-read in place
constexpr int GB_NP = 4;   // preference chunks kept in registers one step ahead (Z <= 256); longer rows are re-read in place

__device__ __forceinline__ bool sorted_contains(const int* __restrict__ a, int n, int x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && a[lo] == x;
}
// lag[i] = smallest d in [1, GB_WAVES) such that the users at positions i and i - d interact; 0 if none.  One thread per (i, d).
__global__ __launch_bounds__(BLOCK) void k_greedy_conflicts(int K, const GreedyHdr* __restrict__ hdr, const int* __restrict__ so_indices,
                                                            const int* __restrict__ q_indices, int* __restrict__ lag) {
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    const int i = t / (GB_WAVES - 1), d = t % (GB_WAVES - 1) + 1;
    if (i >= K || i - d < 0) return;
    const GreedyHdr a = hdr[i - d], b = hdr[i];
    const int* na = so_indices + a.sb;
    const int* nb = so_indices + b.sb;
    bool hit = sorted_contains(na, a.deg, b.k) || sorted_contains(nb, b.deg, a.k) || sorted_contains(q_indices + b.qb, b.qdeg, a.k);
    for (int x = 0, y = 0; !hit && x < a.deg && y < b.deg;) {  // common out-neighbour: merge of the two sorted lists
        const int u = na[x], v = nb[y];
        if (u == v) hit = true;
        else if (u < v) ++x;
        else ++y;
    }
    if (hit) atomicMin(&lag[i], d);
}

// SCHED: the steps come from k_greedy_schedule -- `hdr` then holds GB_WAVES headers per step in schedule order (k = -1: no user for
// this wave), `nsteps_p` the number of steps, and `lag` is not read.
template <bool SLOT_LDS, bool SCHED = false>
__global__ __launch_bounds__(GB_WAVES * 64) void k_greedy_b(int K, int Z, const GreedyHdr* __restrict__ hdr, const int* __restrict__ lag,
                                                            const int* __restrict__ pref_all, const int* __restrict__ so_indices,
                                                            const double* __restrict__ so_data, const double* __restrict__ so_hmax,
                                                            const int* __restrict__ q_indices, double* __restrict__ gain_all,
                                                            int* __restrict__ slot_all, int* __restrict__ rem, const int* __restrict__ nsteps_p = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // layout: bad[GB_WAVES][Z] int, lag_l[K] u8 (rounded up to 4; not with SCHED), slot_l[K] int (optional)
    int* bad_all = reinterpret_cast<int*>(smem_raw);
    unsigned char* lag_l = reinterpret_cast<unsigned char*>(smem_raw + (size_t)GB_WAVES * Z * 4);
    int* slot_l = reinterpret_cast<int*>(smem_raw + (size_t)GB_WAVES * Z * 4 + (SCHED ? (size_t)0 : (((size_t)K + 3) & ~(size_t)3)));
    const int nsteps = SCHED ? *nsteps_p : 0;
    const int hdr_n = SCHED ? nsteps * GB_WAVES : K;
    __shared__ int unassigned;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int* pref = pref_all + (size_t)b * K * Z;
    double* gain = gain_all + (size_t)b * K * Z;  // [K][Z]
    int* slot_g = slot_all + (size_t)b * K;
    int* slot = SLOT_LDS ? slot_l : slot_g;
    int* bad = bad_all + (size_t)wv * Z;
    for (int i = threadIdx.x; i < K; i += GB_WAVES * 64) {
        if (!SCHED) {
            const int l = lag[i];
            lag_l[i] = (unsigned char)(l >= GB_WAVES ? 0 : l);
        }
        if (SLOT_LDS) slot_l[i] = -1;
    }
    if (threadIdx.x == 0) unassigned = 0;

    // static data of the user this wave may decide in the coming step (position start + wv), requested one step ahead
    GreedyHdr h;
    int r_n[GB_NE], r_q = 0, r_p[GB_NP];
    double r_v[GB_NE], r_h[GB_NE];
    auto prefetch = [&](int pos) {
        if (pos >= hdr_n) { h.k = -1; h.deg = 0; h.qdeg = 0; return; }
        h = hdr[pos];
        if (SCHED && h.k < 0) { h.deg = 0; h.qdeg = 0; return; }
#pragma unroll
        for (int i = 0; i < GB_NE; ++i) {
            const int e = lane + 64 * i;
            r_n[i] = e < h.deg ? so_indices[h.sb + e] : 0;
            r_v[i] = e < h.deg ? so_data[h.sb + e] : 0.0;
            r_h[i] = e < h.deg ? so_hmax[h.sb + e] : 0.0;
        }
        r_q = lane < h.qdeg ? q_indices[h.qb + lane] : 0;
#pragma unroll
        for (int i = 0; i < GB_NP; ++i) {
            const int zz = lane + 64 * i;
            r_p[i] = zz < Z ? pref[(size_t)h.k * Z + zz] : 0;
        }
    };
    prefetch(wv);
    __syncthreads();
    int start = 0, step = 0;
    while (SCHED ? step < nsteps : start < K) {
        // the longest run of positions from `start` in which no two users interact
        int ext = 0;
        if (!SCHED) {
            const int t = lane;  // candidate position start + t
            bool stop = t >= GB_WAVES || start + t >= K;
            if (!stop && t >= 1) {
                const int l = lag_l[start + t];
                stop = l != 0 && l <= t;
            }
            const unsigned long long m = __ballot(stop && t >= 1);
            ext = m ? (int)__builtin_ctzll(m) : GB_WAVES;
        }
        const int next = start + ext;
        // the decision of user `cur` (wave-uniform everything)
        const GreedyHdr cur = h;
        int c_n[GB_NE], c_q = r_q, c_p[GB_NP];
        double c_v[GB_NE], c_h[GB_NE];
#pragma unroll
        for (int i = 0; i < GB_NE; ++i) { c_n[i] = r_n[i]; c_v[i] = r_v[i]; c_h[i] = r_h[i]; }
#pragma unroll
        for (int i = 0; i < GB_NP; ++i) c_p[i] = r_p[i];
        const bool active = SCHED ? cur.k >= 0 : wv < ext;
        // the sums on the chain, requested together: own row and, through slot[], each neighbour's sum in its slot
        double g_self[GB_NP];
        int zn_e[GB_NE];
        double g_ne[GB_NE];
        if (active) {
#pragma unroll
            for (int i = 0; i < GB_NP; ++i) {
                const int z = lane + 64 * i;
                g_self[i] = z < Z ? __hip_atomic_load(&gain[(size_t)cur.k * Z + z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
#pragma unroll
            for (int i = 0; i < GB_NE; ++i) {
                const int e = lane + 64 * i;
                zn_e[i] = -1;
                g_ne[i] = 0.0;
                if (e < cur.deg) {
                    zn_e[i] = slot[c_n[i]];
                    if (zn_e[i] >= 0) g_ne[i] = __hip_atomic_load(&gain[(size_t)c_n[i] * Z + zn_e[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        prefetch(SCHED ? (step + 1) * GB_WAVES + wv : next + wv);  // after the chain loads: loads return in order, so waiting for those leaves these in flight
        if (active) {
#pragma unroll
            for (int i = 0; i < GB_NP; ++i) {
                const int z = lane + 64 * i;
                if (z < Z) bad[z] = g_self[i] > cur.hk ? 1 : 0;
            }
            for (int z = lane + 64 * GB_NP; z < Z; z += 64)
                bad[z] = __hip_atomic_load(&gain[(size_t)cur.k * Z + z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > cur.hk ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < GB_NE; ++i)
                if (zn_e[i] >= 0 && g_ne[i] + c_v[i] > c_h[i]) bad[zn_e[i]] = 1;
            for (int e = lane + 64 * GB_NE; e < cur.deg; e += 64) {  // lists longer than the register window
                const int n = so_indices[cur.sb + e];
                const int zn = slot[n];
                if (zn >= 0 && __hip_atomic_load(&gain[(size_t)n * Z + zn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + so_data[cur.sb + e] > so_hmax[cur.sb + e])
                    bad[zn] = 1;
            }
            if (lane < cur.qdeg) {
                const int zn = slot[c_q];
                if (zn >= 0) bad[zn] = 1;
            }
            for (int e = lane + 64; e < cur.qdeg; e += 64) {
                const int zn = slot[q_indices[cur.qb + e]];
                if (zn >= 0) bad[zn] = 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // first slot of the preference order that is not ruled out
            int z_take = -1;
#pragma unroll
            for (int i = 0; i < GB_NP; ++i) {
                if (z_take < 0 && 64 * i < Z) {
                    const int zz = lane + 64 * i;
                    const bool ok = zz < Z && !bad[c_p[i]];
                    const unsigned long long m = __ballot(ok);
                    if (m) z_take = __shfl(c_p[i], (int)__builtin_ctzll(m));
                }
            }
            for (int z0 = 64 * GB_NP; z_take < 0 && z0 < Z; z0 += 64) {
                const int zz = lane + z0;
                const int pz = zz < Z ? pref[(size_t)cur.k * Z + zz] : 0;
                const bool ok = zz < Z && !bad[pz];
                const unsigned long long m = __ballot(ok);
                if (m) z_take = __shfl(pz, (int)__builtin_ctzll(m));
            }
            if (z_take >= 0) {
                // fire-and-forget f64 atomic adds, one per address and step; the full barrier below completes them before the next
                // step reads the sums
#pragma unroll
                for (int i = 0; i < GB_NE; ++i)
                    if (lane + 64 * i < cur.deg) unsafeAtomicAdd(&gain[(size_t)c_n[i] * Z + z_take], c_v[i]);
                for (int e = lane + 64 * GB_NE; e < cur.deg; e += 64) unsafeAtomicAdd(&gain[(size_t)so_indices[cur.sb + e] * Z + z_take], so_data[cur.sb + e]);
                if (lane == 0) {
                    slot[cur.k] = z_take;
                    if (SLOT_LDS) slot_g[cur.k] = z_take;
                }
            } else if (lane == 0) {
                atomicAdd(&unassigned, 1);
            }
        }
        __threadfence_block();
        __syncthreads();
        start = next;
        ++step;
    }
    if (threadIdx.x == 0) rem[b] = unassigned;
}

// ---- the steps of k_greedy_b out of order -------------------------------------------------------------------------------------------
// Contiguous runs end at the first pair that interacts (~6 users per step at the benchmark, 1 670 steps for 10 003 users); the users
// behind that pair mostly interact with nobody in flight.  Any order that keeps every interacting pair in sequence order gives the
// sequential result exactly -- two users touch common state only if they interact (k_greedy_conflicts' test), so users that do not
// commute, and every address still receives its additions in sequence order.  k_greedy_cmask records, for every position, which of the
// GS_W positions before it it interacts with; k_greedy_schedule (one wavefront, once per call, shared by the attempts: the schedule does
// not depend on what the users decide) repeatedly takes the first GB_WAVES positions of a window of GS_W whose interacting predecessors
// are all done; k_greedy_sched_headers lays the users' headers out step by step for k_greedy_b<., true>.
// What it buys is bounded by the interaction graph itself: with a pair of the visiting order interacting with probability p (5.5 % at the
// benchmark: the order is by ||gX_k||, unrelated to the geometry, and two users interact when they lie within two neighbourhood radii)
// the longest chain of interacting users is ~ e p K = 1 500 -- measured: 1 670 contiguous runs -> 1 380 - 1 430 scheduled steps, 7.1
// users per step, k_greedy_b 6.47 -> 5.35 ms, + 0.29 ms for the schedule and 0.05 ms more for the masks.
constexpr int GS_W = 32;
__global__ __launch_bounds__(BLOCK) void k_greedy_cmask(int K, const GreedyHdr* __restrict__ hdr, const int* __restrict__ so_indices,
                                                        const int* __restrict__ q_indices, unsigned* __restrict__ cmask) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const int i = (int)(t / GS_W), d = (int)(t % GS_W) + 1;
    if (i >= K || i - d < 0) return;
    const GreedyHdr a = hdr[i - d], b = hdr[i];
    const int* na = so_indices + a.sb;
    const int* nb = so_indices + b.sb;
    bool hit = sorted_contains(na, a.deg, b.k) || sorted_contains(nb, b.deg, a.k) || sorted_contains(q_indices + b.qb, b.qdeg, a.k);
    for (int x = 0, y = 0; !hit && x < a.deg && y < b.deg;) {  // common out-neighbour: merge of the two sorted lists
        const int u = na[x], v = nb[y];
        if (u == v) hit = true;
        else if (u < v) ++x;
        else ++y;
    }
    if (hit) atomicOr(&cmask[i], 1u << (d - 1));
}
constexpr int GS_CHUNK = 8192;  // positions of the masks staged in LDS at a time
__global__ __launch_bounds__(WAVE) void k_greedy_schedule(int K, const unsigned* __restrict__ cmask, int* __restrict__ sched /* [steps][GB_WAVES] */,
                                                          int* __restrict__ nsteps_out) {
    __shared__ unsigned cm_l[GS_CHUNK + GS_W];
    const int lane = threadIdx.x;
    int head = 0, s = 0;
    unsigned done = 0u;  // bit j: position head + j is done
    while (head < K) {
        const int base = head;
        for (int i = lane; i < GS_CHUNK + GS_W; i += WAVE) cm_l[i] = base + i < K ? cmask[base + i] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (head < K && head - base < GS_CHUNK) {
            const int p = head + lane;
            const bool inw = lane < GS_W && p < K;
            const unsigned m = inw ? cm_l[p - base] : 0u;
            // bit d - 1 of `waits`: the position d before this one lies in the window and is not done
            const unsigned waits = (lane >= 1 && lane < GS_W) ? __brev(~done << (GS_W - lane)) : 0u;
            const bool ready = inw && !((done >> (lane & 31)) & 1u) && (m & waits) == 0u;
            const unsigned rb = (unsigned)__ballot(ready);
            const int rank = __popc(rb & ((lane < 32) ? ((1u << lane) - 1u) : 0xFFFFFFFFu));
            const bool take = ready && rank < GB_WAVES;
            const unsigned sel = (unsigned)__ballot(take);
            const int cnt = __popc(sel);
            if (take) sched[(size_t)s * GB_WAVES + rank] = p;
            if (lane >= cnt && lane < GB_WAVES) sched[(size_t)s * GB_WAVES + lane] = -1;  // (ranks are below cnt: another address)
            done |= sel;
            const unsigned nd = ~done;
            const int t = nd ? (int)__builtin_ctz(nd) : 32;  // the window moves past its leading done positions
            head += t;
            done = t >= 32 ? 0u : done >> t;
            ++s;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) *nsteps_out = s;
}
__global__ __launch_bounds__(BLOCK) void k_greedy_sched_headers(int K, const int* __restrict__ nsteps_p, const int* __restrict__ sched,
                                                                const GreedyHdr* __restrict__ hdr, GreedyHdr* __restrict__ hdr_s) {
    const size_t n = (size_t)(*nsteps_p) * GB_WAVES;
    for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < n; e += (size_t)gridDim.x * BLOCK) {
        const int pos = sched[e];
        GreedyHdr h;
        if (pos >= 0 && pos < K) h = hdr[pos];
        else { h.k = -1; h.sb = 0; h.deg = 0; h.qb = 0; h.qdeg = 0; h.pad = 0; h.hk = 0.0; }
        hdr_s[e] = h;
    }
}

}  // namespace mmw
