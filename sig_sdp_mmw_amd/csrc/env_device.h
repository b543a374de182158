// The problem generator and the scorer on the device -- the producer and the consumer either side of the hot path
// (SURVEY.md §8 f2 / f3):
//   * generate_S_Q_hmax (sim_src/env/env.py:136-196): station-to-AP distances, path loss, power control, thresholded receive
//     powers, association, then S_gain / Q_asso / h_max as CSR -- here K x A device kernels that emit the CSR arrays directly
//     (the reference densifies K x A twice, builds Q on a LIL matrix with a Python loop over the APs and takes 9.6 s at K = 10 k);
//   * evaluate_sinr / evaluate_bler (env.py:198-233): per-user SINR under a colouring, the one-survivor rule for users of one AP
//     that share a slot, and the finite-blocklength error model (env.py:107-111).
// Arithmetic is float64 in the reference's operation order (no fused multiply-adds where the reference rounds twice; the
// interference of a user is accumulated member by member in ascending user order like the reference's row sum).  log10 / pow /
// erfc come from the device math library, so values agree with NumPy to the last one or two bits rather than bit for bit; the
// tests hold them to 1e-12 relative and the sparsity patterns to equality.
// The station drop itself (np.random.default_rng(seed).uniform, env.py:59) stays with the caller: it is 2 K numbers.
#pragma once
#include <algorithm>
#include <cmath>
#include <numeric>

#include <cstring>

#include "device_utils.h"
#include "pattern_device.h"
#include "runtime.h"

namespace mmw {

struct EnvParams {
    double L0;            // 20 log10(f / 1e6) + 16 - 28 (env.py:95)
    double noise_dbm;     // env.py:9
    double min_sinr_db;   // 10 log10(min_sinr)  (env.py:140)
    double txp_off_db;    // 10 log10(txp_offset) (env.py:141)
    double thr;           // min_s_n_ratio (env.py:151)
};

// no contraction: the reference rounds every product and sum
__device__ __forceinline__ double env_dist(double x0, double y0, double x1, double y1) {
    const double dx = __dsub_rn(x0, x1), dy = __dsub_rn(y0, y1);
    return sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
}
__device__ __forceinline__ double env_loss(const EnvParams& P, double dis) { return __dadd_rn(P.L0, __dmul_rn(28.0, log10(__dadd_rn(dis, 1.0)))); }

// one wavefront per user: rx[k][a] (unthresholded), association = argmax of the thresholded row (first maximum), row count
__global__ __launch_bounds__(BLOCK) void k_env_rx(int K, int A, EnvParams P, const double* __restrict__ sta, const double* __restrict__ ap,
                                                  double* __restrict__ rx, int* __restrict__ asso) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    for (int k = blockIdx.x * WAVES_PER_BLOCK + wib; k < K; k += gridDim.x * WAVES_PER_BLOCK) {
        const double x = sta[2 * k], y = sta[2 * k + 1];
        double gmax = -1e300;  // max of -loss
        for (int a = lane; a < A; a += WAVE) {
            const double g = -env_loss(P, env_dist(x, y, ap[2 * a], ap[2 * a + 1]));
            gmax = g > gmax ? g : gmax;
        }
        gmax = wave_max(gmax);
        const double t = __dsub_rn(P.min_sinr_db, __dsub_rn(gmax, P.noise_dbm));  // env.py:140
        const double txp = __dadd_rn(t, P.txp_off_db);                            // env.py:141
        double best = -1.0;
        int besta = 0x7fffffff;
        for (int a = lane; a < A; a += WAVE) {
            const double loss = env_loss(P, env_dist(x, y, ap[2 * a], ap[2 * a + 1]));
            const double db = __dsub_rn(__dsub_rn(txp, loss), P.noise_dbm);  // env.py:148
            const double v = pow(10.0, db / 10.0);
            rx[(size_t)k * A + a] = v;
            const double vt = v < P.thr ? 0.0 : v;
            if (vt > best) { best = vt; besta = a; }  // ascending a within a lane: keeps the first maximum
        }
        // first maximum over the wave: larger value wins, ties to the smaller AP index (np.argmax, env.py:177)
        for (int o = 32; o >= 1; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int oa = __shfl_xor(besta, o);
            if (ob > best || (ob == best && oa < besta)) { best = ob; besta = oa; }
        }
        if (lane == 0) asso[k] = besta;
    }
}
// users of every AP in ascending order: one wavefront per AP scans the users (ordered compaction by ballot)
__global__ __launch_bounds__(BLOCK) void k_env_ap_members(int K, int A, const int* __restrict__ asso, const int* __restrict__ ap_ptr,
                                                          int* __restrict__ ap_cnt, int* __restrict__ ap_mem) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    for (int a = blockIdx.x * WAVES_PER_BLOCK + wib; a < A; a += gridDim.x * WAVES_PER_BLOCK) {
        int n = 0;
        const int base = ap_ptr ? ap_ptr[a] : 0;
        for (int k0 = 0; k0 < K; k0 += WAVE) {
            const int k = k0 + lane;
            const bool hit = k < K && asso[k] == a;
            const unsigned long long m = __ballot(hit);
            if (hit && ap_mem) ap_mem[base + n + __popcll(m & ((1ull << lane) - 1ull))] = k;
            n += __popcll(m);
        }
        if (lane == 0 && !ap_mem) ap_cnt[a] = n;
    }
}
// row lengths of S (thresholded) and Q: S row k holds every user j whose AP hears k above the threshold
__global__ __launch_bounds__(BLOCK) void k_env_rowlen(int K, int A, EnvParams P, const double* __restrict__ rx, const int* __restrict__ asso,
                                                      const int* __restrict__ ap_cnt, int real, int* __restrict__ s_len, int* __restrict__ q_len) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    for (int k = blockIdx.x * WAVES_PER_BLOCK + wib; k < K; k += gridDim.x * WAVES_PER_BLOCK) {
        int n = 0;
        for (int a = lane; a < A; a += WAVE) {
            const double v = rx[(size_t)k * A + a];
            if ((real ? v : (v < P.thr ? 0.0 : v)) != 0.0) n += ap_cnt[a];
        }
        n = wave_sum(n);
        if (lane == 0) {
            s_len[k] = n;
            if (q_len) q_len[k] = ap_cnt[asso[k]] - 1;
        }
    }
}
// CSR rows of S = rx[:, asso] without explicit zeros (env.py:190-192), of Q (same AP, no diagonal, env.py:181-189) and h_max
__global__ __launch_bounds__(BLOCK) void k_env_fill(int K, int A, EnvParams P, const double* __restrict__ rx, const int* __restrict__ asso, int real,
                                                    const int* __restrict__ s_ptr, int* __restrict__ s_idx, double* __restrict__ s_val,
                                                    const int* __restrict__ q_ptr, int* __restrict__ q_idx, double* __restrict__ q_val,
                                                    double min_sinr, double* __restrict__ h_max) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    for (int k = blockIdx.x * WAVES_PER_BLOCK + wib; k < K; k += gridDim.x * WAVES_PER_BLOCK) {
        int ns = s_ptr[k], nq = q_ptr ? q_ptr[k] : 0;
        const int ak = asso[k];
        if (lane == 0 && h_max) h_max[k] = -1.0;  // a user whose own link is below the threshold: 0 / min_sinr - 1
        for (int j0 = 0; j0 < K; j0 += WAVE) {
            const int j = j0 + lane;
            double v = 0.0;
            int aj = -1;
            if (j < K) {
                aj = asso[j];
                v = rx[(size_t)k * A + aj];
                if (!real && v < P.thr) v = 0.0;
            }
            const bool hs = v != 0.0;
            const unsigned long long ms = __ballot(hs);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (hs) {
                const int o = ns + __popcll(ms & below);
                s_idx[o] = j;
                s_val[o] = v;
                if (j == k && h_max) h_max[k] = __dsub_rn(v / min_sinr, 1.0);  // env.py:194
            }
            ns += __popcll(ms);
            if (q_ptr) {
                const bool hq = j < K && aj == ak && j != k;
                const unsigned long long mq = __ballot(hq);
                if (hq) {
                    const int o = nq + __popcll(mq & below);
                    q_idx[o] = j;
                    q_val[o] = 1.0;
                }
                nq += __popcll(mq);
            }
        }
    }
}
// members of every slot in ascending user order (z outside [0, Z) belongs to no slot, env.py:205)
__global__ __launch_bounds__(BLOCK) void k_env_slot_members(int K, int Z, const double* __restrict__ z, const int* __restrict__ sl_ptr,
                                                            int* __restrict__ sl_cnt, int* __restrict__ sl_mem) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    for (int s = blockIdx.x * WAVES_PER_BLOCK + wib; s < Z; s += gridDim.x * WAVES_PER_BLOCK) {
        int n = 0;
        const int base = sl_ptr ? sl_ptr[s] : 0;
        for (int k0 = 0; k0 < K; k0 += WAVE) {
            const int k = k0 + lane;
            const bool hit = k < K && z[k] == (double)s;
            const unsigned long long m = __ballot(hit);
            if (hit && sl_mem) sl_mem[base + n + __popcll(m & ((1ull << lane) - 1ull))] = k;
            n += __popcll(m);
        }
        if (lane == 0 && !sl_mem) sl_cnt[s] = n;
    }
}
// SINR before the collision rule: signal / (sum over the other members of the slot of their power at this user's AP + 1),
// the sum accumulated member by member in ascending order (the reference's row sum over a column-major slot sub-matrix)
__global__ __launch_bounds__(BLOCK) void k_env_sinr(int K, int A, int Z, const double* __restrict__ rx, const int* __restrict__ asso,
                                                    const double* __restrict__ z, const int* __restrict__ sl_ptr, const int* __restrict__ sl_mem,
                                                    double* __restrict__ sinr) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < K; i += gridDim.x * BLOCK) {
        const double zi = z[i];
        const int s = (int)zi;
        if (!(zi >= 0.0 && zi < (double)Z && (double)s == zi)) { sinr[i] = 1e-3; continue; }
        const int a = asso[i];
        double acc = 0.0;
        for (int m = sl_ptr[s]; m < sl_ptr[s + 1]; ++m) {
            const int j = sl_mem[m];
            acc = __dadd_rn(acc, j == i ? 0.0 : rx[(size_t)j * A + a]);  // own link removed (fill_diagonal, env.py:204)
        }
        sinr[i] = rx[(size_t)i * A + a] / __dadd_rn(acc, 1.0);
    }
}
// users of one AP colliding in a slot: only the strongest survives (first maximum), the others get 1e-3 (env.py:214-224);
// then the block error rate of the finite-blocklength model (env.py:107-111, 227-233)
__global__ __launch_bounds__(BLOCK) void k_env_collide_bler(int K, int Z, const int* __restrict__ asso, const double* __restrict__ z,
                                                            const int* __restrict__ ap_ptr, const int* __restrict__ ap_mem,
                                                            const double* __restrict__ sinr_in, double Lbits, double Bw, double Tslot,
                                                            double* __restrict__ sinr_out, double* __restrict__ bler) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < K; i += gridDim.x * BLOCK) {
        const double zi = z[i];
        double s = sinr_in[i];
        if (zi >= 0.0 && zi < (double)Z && (double)(int)zi == zi) {
            const int a = asso[i];
            bool lose = false;
            for (int m = ap_ptr[a]; m < ap_ptr[a + 1]; ++m) {
                const int j = ap_mem[m];
                if (j != i && z[j] == zi) {
                    const double sj = sinr_in[j];
                    if (sj > s || (sj == s && j < i)) lose = true;
                }
            }
            if (lose) s = 1e-3;
        }
        sinr_out[i] = s;
        if (bler) {
            const double nu = -Lbits * 0.6931471805599453 + Bw * Tslot * log(1.0 + s);
            const double dn = sqrt(Bw * Tslot * (1.0 - 1.0 / ((1.0 + s) * (1.0 + s))));
            bler[i] = 0.5 * erfc((nu / dn) * 0.7071067811865476);  // scipy.stats.norm.sf
        }
    }
}

struct EnvDevice {
    int device = 0, K = 0, A = 0;
    hipStream_t st = nullptr;
    EnvParams P{};
    double min_sinr = 1.0;
    DevBuf<double> sta, ap, rx, s_val, q_val, h_max, zbuf, sinr0, sinr1, bler;
    DevBuf<int> asso, ap_cnt, ap_ptr, ap_mem, s_len, q_len, s_ptr, q_ptr, s_idx, q_idx, sl_cnt, sl_ptr, sl_mem;
    int64_t nnzS = 0, nnzQ = 0;
    std::vector<int32_t> h_sptr, h_qptr;
    // what a solver handle created straight from this generator needs (mmw_create_from_env, pattern_device.h): the transposed receive
    // powers, every user's position in its AP's member list, the per-row counts of the L / S_T' / edge lists (device and host copies),
    // and the station coordinates on the host (the row blocks of such a handle follow a spatial order)
    DevBuf<double> rxT;
    DevBuf<int> appos, cnt6;          // cnt6: [6][K] = nL, nST, nGU, nQU, sdiag, nSym (k_pat_count)
    std::vector<int32_t> h_cnt6;
    std::vector<double> h_sta;
    bool pat_ready = false;
    int pattern_inputs() {
        if (pat_ready) return MMW_OK;
        MMW_HIP(hipSetDevice(device));
        MMW_TRY(rxT.alloc((size_t)A * K));
        MMW_TRY(appos.alloc(K));
        MMW_TRY(cnt6.alloc((size_t)6 * K));
        hipLaunchKernelGGL(k_pat_transpose, dim3((K + 31) / 32, (A + 31) / 32), dim3(256), 0, st, K, A, rx.p, rxT.p);
        hipLaunchKernelGGL(k_pat_appos, dim3(grid_rows(A)), dim3(BLOCK), 0, st, A, ap_ptr.p, ap_mem.p, appos.p);
        int* c = cnt6.p;
        hipLaunchKernelGGL(k_pat_count, dim3(grid_rows(K)), dim3(BLOCK), 0, st, K, A, P.thr, rx.p, rxT.p, asso.p, ap_cnt.p, appos.p, c, c + K, c + 2 * (size_t)K,
                           c + 3 * (size_t)K, c + 4 * (size_t)K, c + 5 * (size_t)K);
        MMW_HIP(hipGetLastError());
        h_cnt6.resize((size_t)6 * K);
        MMW_TRY(copy_d2h(h_cnt6.data(), cnt6.p, h_cnt6.size() * sizeof(int32_t), st));
        pat_ready = true;
        return MMW_OK;
    }
    // the bisection's bounds (binary_search_relaxation.py:13-29) without the host matrices: lower = most users of one AP, upper = longest
    // stored off-diagonal row of S + S^T, + 2 (setdiag(0) leaves a stored entry: see binary_search.py)
    int bounds(int32_t out[2]) {
        MMW_TRY(pattern_inputs());
        int lb = 0, ub = 0;
        for (int k = 0; k < K; ++k) {
            lb = std::max(lb, h_qptr[k + 1] - h_qptr[k]);
            ub = std::max(ub, h_cnt6[(size_t)5 * K + k]);
        }
        out[0] = lb + 1;
        out[1] = ub + 2;
        return MMW_OK;
    }

    ~EnvDevice() {
        if (st) {
            (void)hipSetDevice(device);
            (void)hipStreamDestroy(st);
        }
    }
    static void prefix(const std::vector<int>& len, std::vector<int>& ptr) {
        ptr.assign(len.size() + 1, 0);
        for (size_t i = 0; i < len.size(); ++i) ptr[i + 1] = ptr[i] + len[i];
    }
    int init(int dev, int K_, int A_, const double* sta_xy, const double* ap_xy, double fre_Hz, double txp_offset, double min_s_n_ratio,
             double min_sinr_, double noise_dbm) {
        device = dev; K = K_; A = A_; min_sinr = min_sinr_;
        P.L0 = 20.0 * std::log10(fre_Hz / 1e6) + 16 - 28;
        P.noise_dbm = noise_dbm;
        P.min_sinr_db = 10.0 * std::log10(min_sinr_);
        P.txp_off_db = 10.0 * std::log10(txp_offset);
        P.thr = min_s_n_ratio;
        MMW_HIP(hipSetDevice(device));
        MMW_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        std::vector<double> hs(sta_xy, sta_xy + 2 * (size_t)K), ha(ap_xy, ap_xy + 2 * (size_t)A);
        h_sta = hs;
        MMW_TRY(sta.upload(hs, st));
        MMW_TRY(ap.upload(ha, st));
        MMW_TRY(rx.alloc((size_t)K * A));
        MMW_TRY(asso.alloc(K)); MMW_TRY(ap_cnt.alloc(A)); MMW_TRY(ap_ptr.alloc(A + 1)); MMW_TRY(ap_mem.alloc(K));
        MMW_TRY(s_len.alloc(K)); MMW_TRY(q_len.alloc(K)); MMW_TRY(s_ptr.alloc(K + 1)); MMW_TRY(q_ptr.alloc(K + 1)); MMW_TRY(h_max.alloc(K));
        const int gk = grid_rows(K), ga = grid_rows(A);
        hipLaunchKernelGGL(k_env_rx, dim3(gk), dim3(BLOCK), 0, st, K, A, P, sta.p, ap.p, rx.p, asso.p);
        hipLaunchKernelGGL(k_env_ap_members, dim3(ga), dim3(BLOCK), 0, st, K, A, asso.p, (const int*)nullptr, ap_cnt.p, (int*)nullptr);
        MMW_HIP(hipGetLastError());
        std::vector<int> cnt(A), ptr;
        MMW_HIP(hipMemcpyAsync(cnt.data(), ap_cnt.p, (size_t)A * sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        prefix(cnt, ptr);
        MMW_HIP(hipMemcpyAsync(ap_ptr.p, ptr.data(), (size_t)(A + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_env_ap_members, dim3(ga), dim3(BLOCK), 0, st, K, A, asso.p, (const int*)ap_ptr.p, ap_cnt.p, ap_mem.p);
        MMW_HIP(hipGetLastError());
        MMW_HIP(hipStreamSynchronize(st));  // ptr dies
        return build_state();
    }
    int build_state() {
        const int gk = grid_rows(K);
        hipLaunchKernelGGL(k_env_rowlen, dim3(gk), dim3(BLOCK), 0, st, K, A, P, rx.p, asso.p, ap_cnt.p, 0, s_len.p, q_len.p);
        MMW_HIP(hipGetLastError());
        std::vector<int> ls(K), lq(K);
        MMW_HIP(hipMemcpyAsync(ls.data(), s_len.p, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipMemcpyAsync(lq.data(), q_len.p, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        prefix(ls, h_sptr);
        prefix(lq, h_qptr);
        nnzS = h_sptr[K];
        nnzQ = h_qptr[K];
        MMW_HIP(hipMemcpyAsync(s_ptr.p, h_sptr.data(), (size_t)(K + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        MMW_HIP(hipMemcpyAsync(q_ptr.p, h_qptr.data(), (size_t)(K + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        MMW_TRY(s_idx.alloc((size_t)nnzS)); MMW_TRY(s_val.alloc((size_t)nnzS)); MMW_TRY(q_idx.alloc((size_t)nnzQ)); MMW_TRY(q_val.alloc((size_t)nnzQ));
        hipLaunchKernelGGL(k_env_fill, dim3(gk), dim3(BLOCK), 0, st, K, A, P, rx.p, asso.p, 0, (const int*)s_ptr.p, s_idx.p, s_val.p, (const int*)q_ptr.p,
                           q_idx.p, q_val.p, min_sinr, h_max.p);
        MMW_HIP(hipGetLastError());
        MMW_HIP(hipStreamSynchronize(st));
        return MMW_OK;
    }
    int state(int32_t* Sp, int32_t* Si, double* Sx, int32_t* Qp, int32_t* Qi, double* Qx, double* h) {
        MMW_HIP(hipSetDevice(device));
        memcpy(Sp, h_sptr.data(), (size_t)(K + 1) * sizeof(int32_t));
        memcpy(Qp, h_qptr.data(), (size_t)(K + 1) * sizeof(int32_t));
        MMW_TRY(copy_d2h(Si, s_idx.p, (size_t)nnzS * sizeof(int), st));
        MMW_TRY(copy_d2h(Sx, s_val.p, (size_t)nnzS * sizeof(double), st));
        MMW_TRY(copy_d2h(Qi, q_idx.p, (size_t)nnzQ * sizeof(int), st));
        MMW_TRY(copy_d2h(Qx, q_val.p, (size_t)nnzQ * sizeof(double), st));
        MMW_TRY(copy_d2h(h, h_max.p, (size_t)K * sizeof(double), st));
        MMW_HIP(hipStreamSynchronize(st));
        return MMW_OK;
    }
    // SINR (after the collision rule) and, when bler_out != nullptr, the block error rate of every user under colouring z
    int evaluate(const double* z, int32_t Z, double packet_bit, double bandwidth, double slot_time, double* sinr_out, double* bler_out) {
        if (Z < 1) return fail(MMW_ERR_ARG, "mmw_env_evaluate: Z must be positive");
        MMW_HIP(hipSetDevice(device));
        if (zbuf.n < (size_t)K) { MMW_TRY(zbuf.alloc(K)); MMW_TRY(sinr0.alloc(K)); MMW_TRY(sinr1.alloc(K)); MMW_TRY(bler.alloc(K)); MMW_TRY(sl_mem.alloc(K)); }
        if (sl_cnt.n < (size_t)Z) { MMW_TRY(sl_cnt.alloc(Z)); MMW_TRY(sl_ptr.alloc((size_t)Z + 1)); }
        MMW_HIP(hipMemcpyAsync(zbuf.p, z, (size_t)K * sizeof(double), hipMemcpyHostToDevice, st));
        const int gz = grid_rows(Z), ge = grid_elems((size_t)K);
        hipLaunchKernelGGL(k_env_slot_members, dim3(gz), dim3(BLOCK), 0, st, K, Z, zbuf.p, (const int*)nullptr, sl_cnt.p, (int*)nullptr);
        MMW_HIP(hipGetLastError());
        std::vector<int> cnt(Z), ptr;
        MMW_HIP(hipMemcpyAsync(cnt.data(), sl_cnt.p, (size_t)Z * sizeof(int), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        prefix(cnt, ptr);
        MMW_HIP(hipMemcpyAsync(sl_ptr.p, ptr.data(), (size_t)(Z + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_env_slot_members, dim3(gz), dim3(BLOCK), 0, st, K, Z, zbuf.p, (const int*)sl_ptr.p, sl_cnt.p, sl_mem.p);
        hipLaunchKernelGGL(k_env_sinr, dim3(ge), dim3(BLOCK), 0, st, K, A, Z, rx.p, asso.p, zbuf.p, (const int*)sl_ptr.p, (const int*)sl_mem.p, sinr0.p);
        hipLaunchKernelGGL(k_env_collide_bler, dim3(ge), dim3(BLOCK), 0, st, K, Z, asso.p, zbuf.p, (const int*)ap_ptr.p, (const int*)ap_mem.p, sinr0.p, packet_bit,
                           bandwidth, slot_time, sinr1.p, bler_out ? bler.p : (double*)nullptr);
        MMW_HIP(hipGetLastError());
        MMW_HIP(hipMemcpyAsync(sinr_out, sinr1.p, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, st));
        if (bler_out) MMW_HIP(hipMemcpyAsync(bler_out, bler.p, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        return MMW_OK;
    }
};

}  // namespace mmw
