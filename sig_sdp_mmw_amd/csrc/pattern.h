// Host-side state processing: builds, once per (state, Z), everything whose shape is fixed for the
// whole solve.  Replaces mmw._process_state (sim_src/alg/mmw.py:26-41) and the index prologue of
// mmw._run (mmw.py:49-60).  Plain C++, O(nnz log deg); no device code here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace mmw {

struct HostPattern {
    int32_t K = 0, Z = 0;
    // S_T' (K x K) in CSR: row k holds S_gain[j,k] for j != k, (k,j) not in nz(Q)
    std::vector<int32_t> st_indptr, st_indices;
    std::vector<double> st_data;
    std::vector<double> S_sum, norm_H, cH, h_max;
    // pattern of L / X: diag + sym(pattern(S_T')) + pattern(Q), CSR with sorted columns
    std::vector<int32_t> l_indptr, l_indices;
    std::vector<double> sab;      // S_T'[row,col] (0 when absent)
    std::vector<double> sba;      // S_T'[col,row]
    std::vector<int32_t> pid;     // association pair id (triu-CSR order of Q) or -1
    std::vector<int32_t> mirror;  // position of (col,row)
    std::vector<int32_t> diag_pos;
    std::vector<int32_t> asso_pos;                  // [E_asso] position of (x,y), x<y
    std::vector<int32_t> gain_x, gain_y;            // upper-triangular gain edges, CSR row-major
    std::vector<int32_t> asso_x, asso_y;            // upper-triangular association pairs
    // rounding side: S_gain without its diagonal (row k = what user k emits), Q lists
    std::vector<int32_t> so_indptr, so_indices;
    std::vector<double> so_data;
    std::vector<int32_t> q_indptr, q_indices;
    // A handle built on the device (pattern_device.h) keeps the lists that only the API's reads need on the device until someone asks:
    // their sizes are known from the count pass (>= 0 here), and sq_sum stands in for the pass over st_data in update_slots.
    int64_t n_st = -1, n_gain = -1, n_asso = -1;
    std::vector<double> sq_sum;  // [K] sum_j S_T'[k][j]^2 (ascending j)
    int64_t nnzL() const { return (int64_t)l_indices.size(); }
    int64_t nnzST() const { return n_st >= 0 ? n_st : (int64_t)st_indices.size(); }
    int64_t E_asso() const { return n_asso >= 0 ? n_asso : (int64_t)asso_x.size(); }
    int64_t E_gain() const { return n_gain >= 0 ? n_gain : (int64_t)gain_x.size(); }
    int64_t C() const { return E_asso() + 2 * (int64_t)K; }
};

static inline bool row_has(const int32_t* idx, int32_t lo, int32_t hi, int32_t c, int32_t* where = nullptr) {
    const int32_t* b = idx + lo;
    const int32_t* e = idx + hi;
    const int32_t* p = std::lower_bound(b, e, c);
    if (p != e && *p == c) {
        if (where) *where = (int32_t)(p - idx);
        return true;
    }
    return false;
}

// the Z-dependent scalars (mmw.py:39,167): everything else in HostPattern is independent of the slot count
static inline std::string update_slots(HostPattern& P, int32_t Z) {
    if (Z < 2) return "Z must be >= 2 (the constraints divide by Z-1)";
    const int32_t K = P.K;
    P.Z = Z;
    for (int32_t k = 0; k < K; ++k) {
        double q = 0.0;
        if (!P.sq_sum.empty()) q = P.sq_sum[k];
        else
            for (int32_t i = P.st_indptr[k]; i < P.st_indptr[k + 1]; ++i) q += P.st_data[i] * P.st_data[i];
        const double s = P.S_sum[k];
        const double invK = 1.0 / (double)K;
        const double c = invK * P.h_max[k] - invK / (double)Z * s;
        P.norm_H[k] = std::sqrt(q) * (double)(Z - 1) / (double)(2 * Z) + std::fabs(c);
        P.cH[k] = 1.0 / (double)K * P.h_max[k] - 1.0 / ((double)K * (double)Z) * s;
        if (!(P.norm_H[k] > 0.0)) return "norm_H has a zero entry (user with no interferers and h_max == 0)";
    }
    return "";
}

// returns "" on success, else an error message
static inline std::string build_pattern(HostPattern& P, int32_t K, int32_t Z, const int32_t* Sp, const int32_t* Si,
                                        const double* Sx, const int32_t* Qp, const int32_t* Qi, const double* Qx,
                                        const double* h_max, const std::function<void()>& on_structure = {}) {
    // on_structure: called once l_indptr / l_indices are final (the mirror and edge-list passes still to come): the caller may
    // start work that only reads the pattern's structure
    if (K < 2) return "K must be >= 2";
    if (Z < 2) return "Z must be >= 2 (the constraints divide by Z-1)";
    if (Sp[0] != 0 || Qp[0] != 0) return "indptr[0] must be 0";
    for (int32_t k = 0; k < K; ++k) {
        if (Sp[k + 1] < Sp[k] || Qp[k + 1] < Qp[k]) return "indptr must be non-decreasing";
        for (int32_t i = Sp[k]; i < Sp[k + 1]; ++i) {
            if (Si[i] < 0 || Si[i] >= K) return "S_gain column index out of range";
            if (i > Sp[k] && Si[i] <= Si[i - 1]) return "S_gain must be canonical CSR (sorted, no duplicates)";
        }
        for (int32_t i = Qp[k]; i < Qp[k + 1]; ++i) {
            if (Qi[i] < 0 || Qi[i] >= K) return "Q_asso column index out of range";
            if (i > Qp[k] && Qi[i] <= Qi[i - 1]) return "Q_asso must be canonical CSR (sorted, no duplicates)";
            if (Qx[i] == 0.0) return "Q_asso must not store explicit zeros";
            if (Qi[i] == k) return "Q_asso must have an empty diagonal";
        }
    }
    P.K = K;
    P.Z = Z;
    P.h_max.assign(h_max, h_max + K);
    P.q_indptr.assign(Qp, Qp + K + 1);
    P.q_indices.assign(Qi, Qi + Qp[K]);

    // ---- filtered S (row j: what j emits, minus diagonal / association pairs / explicit zeros)
    // and its transpose S_T' by a counting pass.
    std::vector<int32_t> f_indptr(K + 1, 0), f_indices;
    std::vector<double> f_data;
    f_indices.reserve((size_t)Sp[K]);
    f_data.reserve((size_t)Sp[K]);
    P.so_indices.reserve((size_t)Sp[K]);
    P.so_data.reserve((size_t)Sp[K]);
    P.so_indptr.assign(K + 1, 0);
    std::vector<int32_t> cnt(K + 1, 0);
    for (int32_t j = 0; j < K; ++j) {
        for (int32_t i = Sp[j]; i < Sp[j + 1]; ++i) {
            const int32_t k = Si[i];
            const double v = Sx[i];
            if (k == j || v == 0.0) continue;
            P.so_indices.push_back(k);
            P.so_data.push_back(v);
            // (k, j) in nz(Q)?  mmw.py:29-30 zeroes S_T'[x,y] for (x,y) in nz(Q)
            if (row_has(Qi, Qp[k], Qp[k + 1], j)) continue;
            f_indices.push_back(k);
            f_data.push_back(v);
            cnt[k + 1]++;
        }
        f_indptr[j + 1] = (int32_t)f_indices.size();
        P.so_indptr[j + 1] = (int32_t)P.so_indices.size();
    }
    P.st_indptr.assign(K + 1, 0);
    for (int32_t k = 0; k < K; ++k) P.st_indptr[k + 1] = P.st_indptr[k] + cnt[k + 1];
    const int64_t nst = P.st_indptr[K];
    P.st_indices.resize(nst);
    P.st_data.resize(nst);
    {
        std::vector<int32_t> fill(P.st_indptr.begin(), P.st_indptr.end() - 1);
        for (int32_t j = 0; j < K; ++j)
            for (int32_t i = f_indptr[j]; i < f_indptr[j + 1]; ++i) {
                const int32_t k = f_indices[i];
                const int32_t d = fill[k]++;
                P.st_indices[d] = j;  // ascending j within row k because j ascends outside
                P.st_data[d] = f_data[i];
            }
    }
    // ---- S_sum, norm_H (mmw.py:34-39); sums run in ascending column order like csc @ ones
    P.S_sum.assign(K, 0.0);
    P.norm_H.assign(K, 0.0);
    P.cH.assign(K, 0.0);
    for (int32_t k = 0; k < K; ++k) {
        double s = 0.0, q = 0.0;
        for (int32_t i = P.st_indptr[k]; i < P.st_indptr[k + 1]; ++i) {
            s += P.st_data[i];
            q += P.st_data[i] * P.st_data[i];
        }
        P.S_sum[k] = s;
        const double invK = 1.0 / (double)K;
        const double c = invK * h_max[k] - invK / (double)Z * s;
        P.norm_H[k] = std::sqrt(q) * (double)(Z - 1) / (double)(2 * Z) + std::fabs(c);
        P.cH[k] = 1.0 / (double)K * h_max[k] - 1.0 / ((double)K * (double)Z) * s;
        if (!(P.norm_H[k] > 0.0)) return "norm_H has a zero entry (user with no interferers and h_max == 0)";
    }
    // ---- association pair ids in triu-CSR order (mmw.py:57)
    std::vector<int32_t> q_pid(Qp[K], -1);
    for (int32_t x = 0; x < K; ++x)
        for (int32_t i = Qp[x]; i < Qp[x + 1]; ++i)
            if (Qi[i] > x) {
                q_pid[i] = (int32_t)P.asso_x.size();
                P.asso_x.push_back(x);
                P.asso_y.push_back(Qi[i]);
            }
    for (int32_t x = 0; x < K; ++x)
        for (int32_t i = Qp[x]; i < Qp[x + 1]; ++i)
            if (Qi[i] < x) {
                int32_t w;
                if (!row_has(Qi, Qp[Qi[i]], Qp[Qi[i] + 1], x, &w)) return "Q_asso must be symmetric";
                q_pid[i] = q_pid[w];
            }
    if ((int64_t)P.asso_x.size() * 2 != (int64_t)Qp[K]) return "Q_asso must be symmetric";
    // ---- L pattern rows: merge {a}, S_T' row a, filtered-S row a, Q row a
    P.l_indptr.assign(K + 1, 0);
    P.diag_pos.assign(K, -1);
    {
        const size_t cap = (size_t)2 * (size_t)nst + (size_t)Qp[K] + (size_t)K;
        P.l_indices.reserve(cap); P.sab.reserve(cap); P.sba.reserve(cap); P.pid.reserve(cap);
    }
    for (int32_t a = 0; a < K; ++a) {
        int32_t i1 = P.st_indptr[a], e1 = P.st_indptr[a + 1];  // S_T'[a, .]
        int32_t i2 = f_indptr[a], e2 = f_indptr[a + 1];        // S_T'[., a] (filtered S row a)
        int32_t i3 = Qp[a], e3 = Qp[a + 1];
        bool diag_done = false;
        while (true) {
            int32_t c = INT32_MAX;
            if (i1 < e1) c = std::min(c, P.st_indices[i1]);
            if (i2 < e2) c = std::min(c, f_indices[i2]);
            if (i3 < e3) c = std::min(c, Qi[i3]);
            if (!diag_done && a <= c) {
                P.diag_pos[a] = (int32_t)P.l_indices.size();
                P.l_indices.push_back(a);
                P.sab.push_back(0.0);
                P.sba.push_back(0.0);
                P.pid.push_back(-1);
                diag_done = true;
                continue;
            }
            if (c == INT32_MAX) break;
            double vab = 0.0, vba = 0.0;
            int32_t id = -1;
            if (i1 < e1 && P.st_indices[i1] == c) vab = P.st_data[i1++];
            if (i2 < e2 && f_indices[i2] == c) vba = f_data[i2++];
            if (i3 < e3 && Qi[i3] == c) id = q_pid[i3++];
            if (id >= 0 && (vab != 0.0 || vba != 0.0)) return "internal: gain edge on an association pair";
            P.l_indices.push_back(c);
            P.sab.push_back(vab);
            P.sba.push_back(vba);
            P.pid.push_back(id);
        }
        P.l_indptr[a + 1] = (int32_t)P.l_indices.size();
    }
    const int64_t nnz = P.nnzL();
    if (nnz > (int64_t)INT32_MAX) return "pattern too large for int32 indexing";
    if (on_structure) on_structure();
    // ---- mirrors, edge lists.  The pattern is symmetric and every row is sorted, so while the rows a are swept in ascending
    // order the entries (b, a) of any row b are met in that row's own order: one cursor per row replaces a binary search per entry.
    P.mirror.assign(nnz, -1);
    P.asso_pos.assign(P.asso_x.size(), -1);
    P.gain_x.reserve((size_t)nnz / 2);
    P.gain_y.reserve((size_t)nnz / 2);
    {
        std::vector<int32_t> cursor(P.l_indptr.begin(), P.l_indptr.end() - 1);
        for (int32_t a = 0; a < K; ++a)
            for (int32_t e = P.l_indptr[a]; e < P.l_indptr[a + 1]; ++e) {
                const int32_t b = P.l_indices[e];
                const int32_t w = cursor[b]++;
                if (w >= P.l_indptr[b + 1] || P.l_indices[w] != a) return "internal: asymmetric pattern";
                P.mirror[e] = w;
                if (b > a) {
                    if (P.pid[e] >= 0) {
                        P.asso_pos[P.pid[e]] = e;
                    } else {
                        P.gain_x.push_back(a);
                        P.gain_y.push_back(b);
                    }
                }
            }
    }
    return "";
}

}  // namespace mmw
