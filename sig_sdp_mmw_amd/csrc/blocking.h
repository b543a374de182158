// Host-side locality blocking of the fixed L pattern for the LDS-staged SpMM / SDDMM kernels.
//
// Interference graphs from the journal generator are geometric: users near the same access points
// share most of their neighbours.  A reverse Cuthill-McKee ordering of the symmetric pattern makes that
// locality one-dimensional; consecutive rows are then cut greedily into row blocks whose *union* of
// column indices fits one LDS tile (<= BLK_UNION rows of 256 bytes).  A workgroup stages the union's rows
// of the dense block once in LDS and serves every nonzero of the row block from there, so each gathered
// row is fetched from L2/HBM once per row block instead of once per nonzero (reuse ~9x at N = 10 k, 1 %).
// Only the traversal order is permuted: the dense blocks, the value arrays and every index that crosses
// the C-ABI stay in the caller's original user order.
#pragma once
#include <functional>
#include <thread>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include "blk_config.h"
#include <queue>
#include <vector>

namespace mmw {

constexpr int BLK_UNION = MMW_BLK_UNION;      // rows of the staged tile (448 x 256 B = 112 KiB of the 160 KiB LDS)
constexpr int BLK_ROWS = 64;        // max matrix rows per block
constexpr int BLK_META_BYTES = MMW_BLK_META;  // LDS bytes for the block's (local index, value) entries
constexpr int BLK_CHUNK = 16;       // entries per wave step (4 lane groups x 4); rows are padded to this
constexpr int MF_UNION = 640;       // union columns of a matrix-core block (its LDS holds 16 of them at a time, not all)
constexpr int MF_KSTEP_PAD = 4;     // a block's 16-row k-steps are padded to a multiple of this in the fragment image (== MF_KPAD)

struct HostBlocking {
    bool usable = false;
    bool fits_half_tile = false;        // every block also fits the half-tile kernel's LDS budget
    double reuse = 0.0;                 // nnz / sum of union sizes
    std::vector<int32_t> order;         // RCM order: position -> original row
    std::vector<int32_t> blk_rowptr;    // [nb+1] into `order`
    std::vector<int32_t> un_ptr;        // [nb+1] into un_cols
    std::vector<int32_t> un_cols;       // original column ids of each block's union
    std::vector<int32_t> bptr;          // [K+1] blocked entry ranges (each row padded to a multiple of BLK_CHUNK)
    std::vector<uint16_t> lidx;         // [nent] local index of the column inside the block's union (padding: 0)
    std::vector<int32_t> bpos;          // [nnz] original CSR position -> blocked entry position
    std::vector<int32_t> bepos;         // [nent] blocked entry -> original CSR position (padding: -1)
    int64_t nent = 0;                   // entries including padding
    std::vector<uint16_t> self_li;      // [K] by position: local index of the row's own column (0 if absent)
    // SDDMM side: the upper-triangular entries (col > row) of every block, by local indices
    std::vector<int32_t> sd_ptr;        // [nb+1]
    std::vector<uint16_t> sd_la, sd_lb; // local index of the row / of the column in the block's union
    std::vector<int32_t> sd_epos;       // original CSR position of the entry
    int sd_max = 0;                     // largest per-block entry count
    bool sd_ready = false;              // sd_* / sd2_* below are built (build_sd_tables): only the LDS-staged SDDMM kernels read them
    // half-tile SDDMM (k_sddmm_blk2): the same entries dealt to thread slots, slot = round * 512 + thread.  The two
    // lanes of an LDS service group that read the same 16-byte chunk hold entries of complementary row parity (both
    // for the row side and the column side), so every read is bank-conflict free; -1 marks an idle slot.
    std::vector<int32_t> sd2_ptr;       // [nb+1] slot ranges (multiples of 512)
    std::vector<uint32_t> sd2_ab;       // [slots] la | lb << 16
    std::vector<int32_t> sd2_epos;      // [slots] original CSR position, -1 idle
    int sd2_rounds = 0;                 // most rounds any block needs
    int un8_max = 0;                    // largest union, rounded up to 8 rows
    // one 8-int record per row block {q0, rows, m0, entries, un0, union size, chunks, 0} and the union's column
    // ids at a fixed stride (BLK_UNION per block, padded with the block's first column): a workgroup finds
    // everything it needs from its block id alone, without a chain of dependent index loads
    std::vector<int32_t> desc;
    std::vector<int32_t> un_fixed;
    // matrix-core SpMM (k_spmm_mfma): the block's rows x union as a dense 32 x (16 * ksteps) operand in MFMA fragment order.
    // kbase[b] = k-steps (16 union rows each) of the blocks before b; fpos[e] = position (32-bit words) of CSR entry e in the
    // fragment image [k-step][row tile][half j >> 2][lane][j & 3], lane = (local row & 31) + 32 * ((li >> 3) & 1), j = li & 7.
    bool fits_mfma = false;
    int mfma_mt = 1;                    // row tiles of 32 per block (2 when some block has more than 32 rows)
    std::vector<int32_t> kbase;         // [nbm+1]
    std::vector<int32_t> fpos;          // [nnz]
    // The matrix-core kernel has no per-entry LDS budget, so it runs on row blocks of its own: up to 64 rows grown the same
    // way (fewer, larger patches: each staged union row serves ~2x the nonzeros).  m_* mirror order / blk_rowptr / desc / un_fixed.
    std::vector<int32_t> m_order, m_rowptr, m_desc, m_unfixed;
    std::vector<int32_t> rcm_cache;     // the order both blockings start from: RCM of the pattern, or an order the caller knows to be local
    bool grow = true;                   // false: blocks are consecutive runs of that order (a spatial order of a geometric graph needs no growing:
                                        // boustrophedon strips give 21.8 nonzeros per staged row at the benchmark instance, RCM + growing 20.0)
    // matrix-core SDDMM (k_sddmm_mfma): a block's rows x union product comes out in 32 x 32 tiles (row tile, union tile); the
    // off-diagonal pattern entries of every tile, as (row in tile << 5 | column in tile) and CSR position.
    // Tile index = m_tbase[b] + union tile * row tiles + row tile.  A block's union is sorted by blocked position, so the columns
    // that belong to EARLIER blocks come first: the tiles they fill entirely hold no listed entry and are skipped (m_desc[b][6] = first
    // union tile to compute).
    std::vector<int32_t> m_tbase;       // [nbm+1]
    std::vector<int32_t> m_tptr;        // [tiles+1]
    std::vector<uint16_t> m_trc;        // [off-diagonal nnz]
    std::vector<uint16_t> m_tmask;      // [tiles][64]: per lane of the 32x32 accumulator (column lane & 31, rows (v & 3) + 8 (v >> 2) + 4 (lane >> 5)) bit v set where the pattern has an off-diagonal entry
    std::vector<int32_t> m_tepos;       // [listed entries] CSR position of the entry (row -> column) ...
    std::vector<int32_t> m_temir;       // ... and of its mirror (column -> row): X is symmetric, every undirected edge is computed ONCE, by the block of
                                        // the endpoint that comes first in the blocked order, and stored to both positions
    // X in the SDDMM's own order ("tile order"): slot w of the tile lists holds edge w, the K diagonal entries follow by row id.
    // m_e2w[e] = slot of CSR entry e (an edge's two entries share one slot; a diagonal entry has slot n_edges + row).
    std::vector<int32_t> m_e2w;         // [nnz]
    int64_t m_nedges = 0;               // listed entries = undirected edges
    int m_ntile_max = 0;                // most union tiles any block has to compute (from its first tile with a listed entry)
    double m_reuse = 0.0;
    int nbm() const { return (int)m_rowptr.size() - 1; }
    int nb() const { return (int)blk_rowptr.size() - 1; }
};

// reverse Cuthill-McKee on a symmetric CSR pattern (diagonal entries ignored)
inline std::vector<int32_t> rcm_order(int K, const std::vector<int32_t>& indptr, const std::vector<int32_t>& indices) {
    std::vector<int32_t> deg(K), order;
    order.reserve(K);
    for (int k = 0; k < K; ++k) deg[k] = indptr[k + 1] - indptr[k];
    std::vector<char> seen(K, 0);
    std::vector<int32_t> by_deg(K);
    std::iota(by_deg.begin(), by_deg.end(), 0);
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    std::vector<int32_t> nb;
    auto bfs = [&](int start, std::vector<int32_t>& out, std::vector<char>& mark) {
        size_t head = out.size();
        out.push_back(start);
        mark[start] = 1;
        int last_level_first = start;
        while (head < out.size()) {
            const int u = out[head++];
            nb.clear();
            for (int e = indptr[u]; e < indptr[u + 1]; ++e) {
                const int v = indices[e];
                if (!mark[v]) {
                    mark[v] = 1;
                    nb.push_back(v);
                }
            }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
            for (int v : nb) out.push_back(v);
            if (!nb.empty()) last_level_first = nb.back();
        }
        return last_level_first;
    };
    size_t cursor = 0;
    while ((int)order.size() < K) {
        while (seen[by_deg[cursor]]) ++cursor;
        int start = by_deg[cursor];
        // pseudo-peripheral start: two BFS sweeps from the min-degree node of the component
        if (deg[start] > 1) {
            std::vector<char> tmp(seen);
            std::vector<int32_t> scratch;
            int far = bfs(start, scratch, tmp);
            std::vector<char> tmp2(seen);
            scratch.clear();
            far = bfs(far, scratch, tmp2);
            start = far;
        }
        bfs(start, order, seen);
    }
    std::reverse(order.begin(), order.end());
    return order;
}

// LDS budget of the half-tile kernel (k_spmm_blk2: two workgroups per CU): a fixed header, the union's rows at
// 128 B each, then the block's staged entries
constexpr int BLK2_LDS_BYTES = 79872;
constexpr int BLK2_HEADER_BYTES = 5120;
constexpr int BLK2_ROW_BYTES = 128;
constexpr int SD2_THREADS = 1024;    // workgroup size of the half-tile SDDMM
struct BlockingLimits {
    int max_entries_per_block;  // staged entries of the full-tile kernel
    int entry_bytes;            // sizeof one staged entry (offset + value)
};
inline int blk2_lds_need(int nun, int64_t entries, int entry_bytes) {
    return BLK2_HEADER_BYTES + ((nun + 7) & ~7) * BLK2_ROW_BYTES + ((int)entries + 2 * 16) * entry_bytes;  // two chunks of slack: the pair prefetch reads ahead
}

inline void build_blocking(HostBlocking& B, int K, const std::vector<int32_t>& indptr, const std::vector<int32_t>& indices,
                           const BlockingLimits& lim) {
    const int max_entries_per_block = lim.max_entries_per_block;
    const int64_t nnz = indptr[K];
    B.usable = false;
    auto padded = [](int n) { return (n + BLK_CHUNK - 1) / BLK_CHUNK * BLK_CHUNK; };
    for (int k = 0; k < K; ++k)
        if (padded(indptr[k + 1] - indptr[k]) + BLK_CHUNK > max_entries_per_block ||
            blk2_lds_need(indptr[k + 1] - indptr[k], padded(indptr[k + 1] - indptr[k]) + BLK_CHUNK, lim.entry_bytes) > BLK2_LDS_BYTES)
            return;
    for (int k = 0; k < K; ++k)
        if (indptr[k + 1] - indptr[k] > BLK_UNION) return;  // a single row overflows the tile: generic kernel
    if (B.rcm_cache.size() != (size_t)K) B.rcm_cache = rcm_order(K, indptr, indices);  // the caller may have made it already
    const std::vector<int32_t>& rcm = B.rcm_cache;
    std::vector<int32_t> rank(K);
    for (int p = 0; p < K; ++p) rank[rcm[p]] = p;
    // Row blocks.  Seeds sweep the RCM order; a block then GROWS from its seed: the next row is the unassigned member of the
    // union that brings the fewest new columns (ties: lowest RCM rank).  On a geometric graph this makes compact
    // two-dimensional patches instead of slices of the one-dimensional RCM order: ~25 % smaller unions for the same rows,
    // i.e. fewer bytes gathered into LDS per nonzero (the gathers run at the CU's L2 rate).
    // MMW_BLK_GROW=0 keeps consecutive RCM rows.
    const bool grow = B.grow && !(getenv("MMW_BLK_GROW") && atoi(getenv("MMW_BLK_GROW")) == 0);
    B.order.assign(K, -1);  // filled block by block: position -> original row
    std::vector<char> assigned(K, 0);
    std::vector<int32_t> in_union(K, 0), in_stamp(K, -1);  // neighbours of v inside the current union (valid for stamp == blk)
    int seed_pos = 0;
    B.blk_rowptr.assign(1, 0);
    B.un_ptr.assign(1, 0);
    B.un_cols.clear();
    std::vector<int32_t> stamp(K, -1), loc(K, 0), cur;
    int blk = 0, p = 0;
    const int row_cap = getenv("MMW_BLK_ROWS") ? atoi(getenv("MMW_BLK_ROWS")) : BLK_ROWS;
    const int row_quant = getenv("MMW_BLK_QUANT") ? atoi(getenv("MMW_BLK_QUANT")) : 1;
    while (p < K) {
        cur.clear();
        int rows = 0, entries = 0;
        const int p_start = p;
        while (assigned[rcm[seed_pos]]) ++seed_pos;
        int r = rcm[seed_pos];
        while (p < K && rows < row_cap) {
            int fresh = 0;
            for (int e = indptr[r]; e < indptr[r + 1]; ++e)
                if (stamp[indices[e]] != blk) ++fresh;
            const int pe = padded(indptr[r + 1] - indptr[r]) + BLK_CHUNK;  // room for the parity padding (exact count below)
            if (rows > 0 && ((int)cur.size() + fresh > BLK_UNION || entries + pe > max_entries_per_block ||
                             blk2_lds_need((int)cur.size() + fresh, entries + pe, lim.entry_bytes) > BLK2_LDS_BYTES))
                break;
            entries += pe;
            for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
                const int c = indices[e];
                if (stamp[c] != blk) {
                    stamp[c] = blk;
                    cur.push_back(c);
                    if (grow)
                        for (int f = indptr[c]; f < indptr[c + 1]; ++f) {  // symmetric pattern: c's row lists who has c as a column
                            const int v = indices[f];
                            if (in_stamp[v] != blk) { in_stamp[v] = blk; in_union[v] = 0; }
                            ++in_union[v];
                        }
                }
            }
            assigned[r] = 1;
            B.order[p] = r;
            ++rows;
            ++p;
            if (p >= K) break;
            // next row
            int best = -1;
            if (grow) {
                int best_fresh = INT32_MAX;
                for (int c : cur)
                    if (!assigned[c]) {
                        const int fr = (indptr[c + 1] - indptr[c]) - (in_stamp[c] == blk ? in_union[c] : 0);
                        if (fr < best_fresh || (fr == best_fresh && rank[c] < rank[best])) { best_fresh = fr; best = c; }
                    }
            } else {
                while (seed_pos < K && assigned[rcm[seed_pos]]) ++seed_pos;
                best = seed_pos < K ? rcm[seed_pos] : -1;
            }
            if (best < 0) break;  // the union holds no unassigned row (a finished component): next block, next seed
            r = best;
        }
        if (row_quant > 1 && rows > row_quant && rows % row_quant && p < K) {  // trim to a multiple of the wave count
            const int keep = rows / row_quant * row_quant;
            for (int q = p_start + keep; q < p; ++q) assigned[B.order[q]] = 0;
            p = p_start + keep;
            seed_pos = 0;
            cur.clear();
            ++blk;  // fresh stamp generation for the rebuilt union
            for (int q = p_start; q < p; ++q) {
                const int r = B.order[q];
                for (int e = indptr[r]; e < indptr[r + 1]; ++e)
                    if (stamp[indices[e]] != blk) {
                        stamp[indices[e]] = blk;
                        cur.push_back(indices[e]);
                    }
            }
        }
        std::sort(cur.begin(), cur.end(), [&](int a, int b) { return rank[a] < rank[b]; });
        // exact staged entry count under the parity arrangement; shed rows until both kernels' budgets hold
        for (;;) {
            for (size_t u = 0; u < cur.size(); ++u) loc[cur[u]] = (int32_t)u;
            int64_t exact = 0;
            for (int q = p_start; q < p; ++q) {
                const int r = B.order[q];
                int ne = 0, no = 0;
                for (int e = indptr[r]; e < indptr[r + 1]; ++e) ((loc[indices[e]] & 1) ? no : ne)++;
                exact += (int64_t)BLK_CHUNK * std::max(1, std::max((ne + 7) / 8, (no + 7) / 8));
            }
            if (p - p_start <= 1 || (exact <= max_entries_per_block && blk2_lds_need((int)cur.size(), exact, lim.entry_bytes) <= BLK2_LDS_BYTES)) break;
            --p;
            assigned[B.order[p]] = 0;  // given back: it seeds or joins a later block
            seed_pos = 0;
            cur.clear();
            ++blk;
            for (int q = p_start; q < p; ++q) {
                const int r = B.order[q];
                for (int e = indptr[r]; e < indptr[r + 1]; ++e)
                    if (stamp[indices[e]] != blk) {
                        stamp[indices[e]] = blk;
                        cur.push_back(indices[e]);
                    }
            }
            std::sort(cur.begin(), cur.end(), [&](int a, int b) { return rank[a] < rank[b]; });
        }
        B.un_cols.insert(B.un_cols.end(), cur.begin(), cur.end());
        B.un_ptr.push_back((int32_t)B.un_cols.size());
        B.blk_rowptr.push_back(p);
        ++blk;
    }
    // blocked nnz arrays
    B.bptr.assign(K + 1, 0);
    B.self_li.assign(K, 0);
    B.un8_max = 0;
    B.sd_ready = false;
    B.bpos.resize(nnz);
    // Entries in chunks of 16, eight staged rows of even local index and eight of odd: the half-tile kernel
    // reads 128-byte staged rows with 8-lane groups, and the lane groups that share an LDS service group
    // (0|3, 1|2, 4|7, 5|6) then always land on different bank halves.  Chunk positions with bit 2 clear
    // hold even rows, the others odd rows; holes are (row 0 or 1, value 0) entries.
    // The blocks are independent here: a few host threads take contiguous runs of blocks, first to count every row's chunks
    // (positions follow from a prefix sum), then to fill.
    const int nbk = B.nb();
    const int nthreads = std::max(1, std::min({8, nbk, (int)std::thread::hardware_concurrency()}));
    std::vector<int> cut(nthreads + 1, nbk);
    cut[0] = 0;
    for (int t = 1; t < nthreads; ++t) {  // equal shares of the rows
        const int target = (int)((int64_t)K * t / nthreads);
        int bb = cut[t - 1];
        while (bb < nbk && B.blk_rowptr[bb] < target) ++bb;
        cut[t] = bb;
    }
    auto run = [&](const std::function<void(int, int)>& body) {
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(body, cut[t], cut[t + 1]);
        body(cut[0], cut[1]);
        for (auto& x : th) x.join();
    };
    std::vector<int32_t> nch_row(K, 1);
    run([&](int b0, int b1) {
        std::vector<int32_t> local(K, -1);
        for (int b = b0; b < b1; ++b) {
            for (int u = B.un_ptr[b]; u < B.un_ptr[b + 1]; ++u) local[B.un_cols[u]] = u - B.un_ptr[b];
            for (int q = B.blk_rowptr[b]; q < B.blk_rowptr[b + 1]; ++q) {
                const int r = B.order[q];
                B.self_li[q] = local[r] >= 0 ? (uint16_t)local[r] : (uint16_t)0;
                int ne = 0, no = 0;
                for (int e = indptr[r]; e < indptr[r + 1]; ++e) ((local[indices[e]] & 1) ? no : ne)++;
                nch_row[q] = std::max(1, std::max((ne + 7) / 8, (no + 7) / 8));
            }
            for (int u = B.un_ptr[b]; u < B.un_ptr[b + 1]; ++u) local[B.un_cols[u]] = -1;
        }
    });
    int64_t w = 0;
    for (int q = 0; q < K; ++q) {
        w += (int64_t)nch_row[q] * BLK_CHUNK;
        B.bptr[q + 1] = (int32_t)w;
    }
    B.lidx.assign((size_t)w, 0);
    B.bepos.assign((size_t)w, -1);
    run([&](int b0, int b1) {
        std::vector<int32_t> local(K, -1);
        std::vector<std::pair<uint16_t, int32_t>> rowbuf, oddbuf;
        for (int b = b0; b < b1; ++b) {
            for (int u = B.un_ptr[b]; u < B.un_ptr[b + 1]; ++u) local[B.un_cols[u]] = u - B.un_ptr[b];
            for (int q = B.blk_rowptr[b]; q < B.blk_rowptr[b + 1]; ++q) {
                const int r = B.order[q];
                rowbuf.clear();
                oddbuf.clear();
                for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
                    const uint16_t li = (uint16_t)local[indices[e]];
                    ((li & 1) ? oddbuf : rowbuf).emplace_back(li, e);
                }
                std::sort(rowbuf.begin(), rowbuf.end());
                std::sort(oddbuf.begin(), oddbuf.end());
                int64_t at = B.bptr[q];
                for (int c = 0; c < nch_row[q]; ++c)
                    for (int p = 0; p < BLK_CHUNK; ++p, ++at) {
                        const bool odd = (p & 4) != 0;
                        const size_t i = (size_t)c * 8 + (p & 3) + ((p & 8) ? 4 : 0);
                        const auto& src = odd ? oddbuf : rowbuf;
                        if (i < src.size()) {
                            B.lidx[at] = src[i].first;
                            B.bepos[at] = src[i].second;
                            B.bpos[src[i].second] = (int32_t)at;
                        } else {
                            B.lidx[at] = odd ? 1 : 0;
                        }
                    }
            }
            for (int u = B.un_ptr[b]; u < B.un_ptr[b + 1]; ++u) local[B.un_cols[u]] = -1;
        }
    });
    B.nent = w;
    B.desc.assign((size_t)B.nb() * 8, 0);
    B.un_fixed.assign((size_t)B.nb() * BLK_UNION, 0);
    for (int b = 0; b < B.nb(); ++b) {
        const int q0 = B.blk_rowptr[b], q1 = B.blk_rowptr[b + 1];
        const int nun = B.un_ptr[b + 1] - B.un_ptr[b];
        int32_t* d = &B.desc[(size_t)b * 8];
        d[0] = q0; d[1] = q1 - q0; d[2] = B.bptr[q0]; d[3] = B.bptr[q1] - B.bptr[q0]; d[4] = B.un_ptr[b]; d[5] = nun;
        d[6] = (B.bptr[q1] - B.bptr[q0]) / BLK_CHUNK;
        d[7] = 0;  // first slot of the half-tile SDDMM: build_sd_tables
        B.un8_max = std::max(B.un8_max, (nun + 7) & ~7);
        for (int u = 0; u < BLK_UNION; ++u) B.un_fixed[(size_t)b * BLK_UNION + u] = B.un_cols[B.un_ptr[b] + (u < nun ? u : 0)];
    }
    B.reuse = B.un_cols.empty() ? 0.0 : (double)nnz / (double)B.un_cols.size();
    B.fits_half_tile = true;
    bool fits_full = true;
    for (int b = 0; b < B.nb(); ++b) {
        const int32_t* d = &B.desc[(size_t)b * 8];
        if (d[3] > max_entries_per_block) fits_full = false;
        if (blk2_lds_need(d[5], d[3], lim.entry_bytes) > BLK2_LDS_BYTES) B.fits_half_tile = false;
    }
    B.usable = B.reuse >= 2.0 && fits_full;
}

// The entry tables of the LDS-staged SDDMM kernels (k_sddmm_blk, k_sddmm_blk2) for the blocks build_blocking made.  A third of
// the blocking's host time, and dead weight for a handle whose SDDMM runs on the matrix cores: built on first need.
inline void build_sd_tables(HostBlocking& B, int K, const std::vector<int32_t>& indptr, const std::vector<int32_t>& indices) {
    if (B.sd_ready || !B.usable) return;
    B.sd_ptr.assign(1, 0);
    B.sd2_ptr.assign(1, 0);
    B.sd2_ab.clear(); B.sd2_epos.clear();
    B.sd2_rounds = 0;
    B.sd_la.clear(); B.sd_lb.clear(); B.sd_epos.clear();
    B.sd_max = 0;
    std::vector<int32_t> local(K, -1);
    for (int b = 0; b < B.nb(); ++b) {
        if (b > 0)
            for (int u = B.un_ptr[b - 1]; u < B.un_ptr[b]; ++u) local[B.un_cols[u]] = -1;
        for (int u = B.un_ptr[b]; u < B.un_ptr[b + 1]; ++u) local[B.un_cols[u]] = u - B.un_ptr[b];
        for (int q = B.blk_rowptr[b]; q < B.blk_rowptr[b + 1]; ++q) {
            const int r = B.order[q];
            for (int e = indptr[r]; e < indptr[r + 1]; ++e)
                if (indices[e] > r) {
                    B.sd_la.push_back((uint16_t)local[r]);
                    B.sd_lb.push_back((uint16_t)local[indices[e]]);
                    B.sd_epos.push_back(e);
                }
        }
        B.sd_ptr.push_back((int32_t)B.sd_epos.size());
        B.sd_max = std::max(B.sd_max, B.sd_ptr[b + 1] - B.sd_ptr[b]);
        {   // slots of the half-tile SDDMM
            struct Ed { uint16_t a, b; int32_t e; };
            std::vector<Ed> ee, oo, mx;  // both rows even / both odd / mixed (stored as even row first)
            for (int i = B.sd_ptr[b]; i < B.sd_ptr[b + 1]; ++i) {
                Ed d{B.sd_la[i], B.sd_lb[i], B.sd_epos[i]};
                const int pa = d.a & 1, pb = d.b & 1;
                if (pa == pb) (pa ? oo : ee).push_back(d);
                else {
                    if (pa) std::swap(d.a, d.b);  // the dot product is symmetric
                    mx.push_back(d);
                }
            }
            // pairs (first lane, second lane): (EE, OO) and (mixed even-first, mixed odd-first); leftovers pair with idle
            std::vector<std::pair<Ed, Ed>> pairs;
            const Ed idle{0, 0, -1};
            const size_t nz = std::max(ee.size(), oo.size());
            for (size_t i = 0; i < nz; ++i) pairs.push_back({i < ee.size() ? ee[i] : idle, i < oo.size() ? oo[i] : idle});
            for (size_t i = 0; i < mx.size(); i += 2) {
                Ed second = idle;
                if (i + 1 < mx.size()) { second = mx[i + 1]; std::swap(second.a, second.b); }
                pairs.push_back({mx[i], second});
            }
            const int per_round = SD2_THREADS / 2;
            const int rounds = std::max(1, (int)((pairs.size() + per_round - 1) / per_round));
            B.sd2_rounds = std::max(B.sd2_rounds, rounds);
            const size_t base = B.sd2_ab.size();
            B.sd2_ab.resize(base + (size_t)rounds * SD2_THREADS, 0u);
            B.sd2_epos.resize(base + (size_t)rounds * SD2_THREADS, -1);
            for (size_t i = 0; i < pairs.size(); ++i) {
                const int round = (int)(i / per_round), pi = (int)(i % per_round);
                const int wave = pi / 32, j = pi % 32;          // 32 pairs per wave: 16 per 32-lane half
                const int half = j / 16, l = j % 16;
                const int lane1 = half * 32 + l, lane2 = lane1 + (l < 8 ? 24 : 8);
                const Ed* d[2] = {&pairs[i].first, &pairs[i].second};
                const int lanes[2] = {lane1, lane2};
                for (int w = 0; w < 2; ++w) {
                    const size_t slot = base + (size_t)round * SD2_THREADS + (size_t)wave * 64 + lanes[w];
                    B.sd2_ab[slot] = (uint32_t)d[w]->a | ((uint32_t)d[w]->b << 16);
                    B.sd2_epos[slot] = d[w]->e;
                }
            }
            B.sd2_ptr.push_back((int32_t)B.sd2_ab.size());
        }
    }
    for (int b = 0; b < B.nb(); ++b) B.desc[(size_t)b * 8 + 7] = B.sd2_ptr[b];
    B.sd_ready = true;
}

// Row blocks for the matrix-core SpMM (kernels_mfma.h): grown like the blocks above (seed = next unassigned row of the RCM
// order, then always the unassigned union member that brings the fewest new columns), bounded only by the union size and
// `max_rows` (32 or 64).  Fills m_order / m_rowptr / m_desc {q0, rows, 0, 0, 0, union size, 0, 0} / m_unfixed, kbase and fpos.
inline void build_mfma_blocking(HostBlocking& B, int K, const std::vector<int32_t>& indptr, const std::vector<int32_t>& indices, int max_rows) {
    const int64_t nnz = indptr[K];
    B.fits_mfma = false;
    if (B.rcm_cache.size() != (size_t)K) return;
    for (int k = 0; k < K; ++k)
        if (indptr[k + 1] - indptr[k] > MF_UNION) return;
    const std::vector<int32_t>& rcm = B.rcm_cache;
    std::vector<int32_t> rank(K);
    for (int p = 0; p < K; ++p) rank[rcm[p]] = p;
    static const int cap_env = getenv("MMW_MF_UNION_CAP") ? atoi(getenv("MMW_MF_UNION_CAP")) : 0;
    const int union_cap = cap_env > 0 ? std::min(cap_env, MF_UNION) : MF_UNION;
    B.m_order.assign(K, -1);
    B.m_rowptr.assign(1, 0);
    std::vector<int32_t> un_ptr(1, 0), un_cols;
    std::vector<char> assigned(K, 0);
    std::vector<int32_t> stamp(K, -1), in_union(K, 0), in_stamp(K, -1), cur;
    int seed_pos = 0, p = 0, blk = 0;
    while (p < K) {
        cur.clear();
        while (assigned[rcm[seed_pos]]) ++seed_pos;
        int r = rcm[seed_pos], rows = 0;
        while (p < K && rows < max_rows) {
            int fresh = 0;
            for (int e = indptr[r]; e < indptr[r + 1]; ++e)
                if (stamp[indices[e]] != blk) ++fresh;
            if (rows > 0 && (int)cur.size() + fresh > union_cap) break;
            for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
                const int c = indices[e];
                if (stamp[c] != blk) {
                    stamp[c] = blk;
                    cur.push_back(c);
                    if (B.grow)
                        for (int f = indptr[c]; f < indptr[c + 1]; ++f) {
                            const int v = indices[f];
                            if (in_stamp[v] != blk) { in_stamp[v] = blk; in_union[v] = 0; }
                            ++in_union[v];
                        }
                }
            }
            assigned[r] = 1;
            B.m_order[p++] = r;
            ++rows;
            int best = -1, best_fresh = INT32_MAX;
            if (B.grow) {
                for (int c : cur)
                    if (!assigned[c]) {
                        const int fr = (indptr[c + 1] - indptr[c]) - (in_stamp[c] == blk ? in_union[c] : 0);
                        if (fr < best_fresh || (fr == best_fresh && rank[c] < rank[best])) { best_fresh = fr; best = c; }
                    }
            } else {  // the next row of the given order
                while (seed_pos < K && assigned[rcm[seed_pos]]) ++seed_pos;
                best = seed_pos < K ? rcm[seed_pos] : -1;
            }
            if (best < 0) break;
            r = best;
        }
        std::sort(cur.begin(), cur.end(), [&](int a, int b) { return rank[a] < rank[b]; });
        un_cols.insert(un_cols.end(), cur.begin(), cur.end());
        un_ptr.push_back((int32_t)un_cols.size());
        B.m_rowptr.push_back(p);
        ++blk;
    }
    const int nbm = B.nbm();
    // unions in blocked order: first the columns of earlier blocks (their edges into this block are the earlier block's to compute),
    // then the block's own rows, then the columns of later blocks
    std::vector<int32_t> mpos(K);
    for (int q = 0; q < K; ++q) mpos[B.m_order[q]] = q;
    for (int b = 0; b < nbm; ++b)
        std::sort(un_cols.begin() + un_ptr[b], un_cols.begin() + un_ptr[b + 1], [&](int a, int c) { return mpos[a] < mpos[c]; });
    B.m_reuse = un_cols.empty() ? 0.0 : (double)nnz / (double)un_cols.size();
    B.mfma_mt = 1;
    B.m_desc.assign((size_t)nbm * 8, 0);
    B.m_unfixed.assign((size_t)nbm * MF_UNION, 0);
    B.kbase.assign(nbm + 1, 0);
    for (int b = 0; b < nbm; ++b) {
        const int nun = un_ptr[b + 1] - un_ptr[b], rows = B.m_rowptr[b + 1] - B.m_rowptr[b];
        int32_t* d = &B.m_desc[(size_t)b * 8];
        d[0] = B.m_rowptr[b]; d[1] = rows; d[5] = nun;
        int n_lo = 0;
        while (n_lo < nun && mpos[un_cols[un_ptr[b] + n_lo]] < B.m_rowptr[b]) ++n_lo;
        d[6] = n_lo / 32;  // first union tile of the SDDMM (the tile that straddles the boundary is computed)
        if (rows > 32) B.mfma_mt = 2;
        for (int u = 0; u < MF_UNION; ++u) B.m_unfixed[(size_t)b * MF_UNION + u] = un_cols[un_ptr[b] + (u < nun ? u : 0)];
        B.kbase[b + 1] = B.kbase[b] + ((nun + 15) / 16 + MF_KSTEP_PAD - 1) / MF_KSTEP_PAD * MF_KSTEP_PAD;
    }
    if ((int64_t)B.kbase[nbm] * B.mfma_mt * 512 >= (int64_t)INT32_MAX || B.m_reuse < 2.0) return;
    const int MT = B.mfma_mt;
    B.fpos.assign(nnz, -1);
    // the blocks are independent from here on: contiguous runs of them on a few host threads
    const int nthreads = std::max(1, std::min({8, nbm, (int)std::thread::hardware_concurrency()}));
    std::vector<int> cut(nthreads + 1, nbm);
    cut[0] = 0;
    for (int t = 1; t < nthreads; ++t) {
        const int target = (int)((int64_t)K * t / nthreads);
        int bb = cut[t - 1];
        while (bb < nbm && B.m_rowptr[bb] < target) ++bb;
        cut[t] = bb;
    }
    auto run = [&](const std::function<void(int, int)>& body) {
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(body, cut[t], cut[t + 1]);
        body(cut[0], cut[1]);
        for (auto& x : th) x.join();
    };
    // tile lists of the SDDMM: counting pass (with the fragment positions), prefix, fill
    B.m_tbase.assign(nbm + 1, 0);
    B.m_ntile_max = 0;
    for (int b = 0; b < nbm; ++b) {
        const int nt = (un_ptr[b + 1] - un_ptr[b] + 31) / 32;
        B.m_ntile_max = std::max(B.m_ntile_max, nt - B.m_desc[(size_t)b * 8 + 6]);
        B.m_tbase[b + 1] = B.m_tbase[b] + nt * MT;
    }
    B.m_tptr.assign((size_t)B.m_tbase[nbm] + 1, 0);
    std::vector<int32_t> fill;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            for (size_t t = 0; t + 1 < B.m_tptr.size(); ++t) B.m_tptr[t + 1] += B.m_tptr[t];
            fill.assign(B.m_tptr.begin(), B.m_tptr.end() - 1);
            B.m_trc.assign((size_t)B.m_tptr.back(), 0);
            B.m_tepos.assign((size_t)B.m_tptr.back(), -1);
            B.m_temir.assign((size_t)B.m_tptr.back(), -1);
            B.m_nedges = B.m_tptr.back();
            B.m_e2w.assign((size_t)nnz, -1);
            B.m_tmask.assign((size_t)B.m_tbase[nbm] * 64, 0);
        }
        run([&, pass](int b0, int b1) {
            std::vector<int32_t> loc2(K, -1);
            for (int b = b0; b < b1; ++b) {
                for (int u = un_ptr[b]; u < un_ptr[b + 1]; ++u) loc2[un_cols[u]] = u - un_ptr[b];
                for (int q = B.m_rowptr[b]; q < B.m_rowptr[b + 1]; ++q) {
                    const int r = B.m_order[q], rl = q - B.m_rowptr[b];
                    for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
                        const int li = loc2[indices[e]];
                        if (pass == 0) {
                            const int ks = li >> 4, j = li & 7, lane = (rl & 31) + 32 * ((li >> 3) & 1), mt = rl >> 5;
                            B.fpos[e] = ((((B.kbase[b] + ks) * MT + mt) * 2 + (j >> 2)) * 64 + lane) * 4 + (j & 3);
                        }
                        if (mpos[indices[e]] <= q) continue;  // the diagonal comes from the exact row norms; an edge to an earlier position is that row's
                        const int tile = B.m_tbase[b] + (li >> 5) * MT + (rl >> 5);  // a block's tiles are its own
                        if (pass == 0) ++B.m_tptr[(size_t)tile + 1];
                        else {
                            const int w = fill[tile]++;
                            B.m_trc[w] = (uint16_t)(((rl & 31) << 5) | (li & 31));
                            B.m_tepos[w] = e;
                            {   // the mirror entry (column -> row): the pattern is symmetric and its rows are sorted
                                const int c = indices[e];
                                const int32_t* lo = std::lower_bound(indices.data() + indptr[c], indices.data() + indptr[c + 1], r);
                                B.m_temir[w] = (int32_t)(lo - indices.data());
                                B.m_e2w[e] = w;
                                B.m_e2w[B.m_temir[w]] = w;
                            }
                            const int r5 = rl & 31;  // accumulator slot of (row r5, column li & 31): lane = column + 32 ((r5 >> 2) & 1), register (r5 & 3) + 4 (r5 >> 3)
                            B.m_tmask[(size_t)tile * 64 + (li & 31) + 32 * ((r5 >> 2) & 1)] |= (uint16_t)(1u << ((r5 & 3) + 4 * (r5 >> 3)));
                        }
                    }
                }
                for (int u = un_ptr[b]; u < un_ptr[b + 1]; ++u) loc2[un_cols[u]] = -1;
            }
        });
    }
    for (int k = 0; k < K; ++k) {  // diagonal entries (stored in every row of the L / X pattern)
        const int32_t* lo = std::lower_bound(indices.data() + indptr[k], indices.data() + indptr[k + 1], k);
        if (lo != indices.data() + indptr[k + 1] && *lo == k) B.m_e2w[(size_t)(lo - indices.data())] = (int32_t)(B.m_nedges + k);
    }
    B.fits_mfma = true;
}

// Invariants of a built blocking, checked on the host (CPU tests, MMW_CHECK_BLOCKING=1): every row in exactly one
// block; every block's union holds all columns of its rows, in budget; the blocked entries are a permutation of the CSR
// entries plus (row 0|1, idle) holes, chunk positions with bit 2 clear hold even staged rows and the others odd ones;
// the SDDMM slots hold every upper-triangular entry once and paired lanes hold complementary row parities.
inline std::string verify_blocking(const HostBlocking& B, int K, const std::vector<int32_t>& indptr, const std::vector<int32_t>& indices,
                                   const BlockingLimits& lim) {
    const int64_t nnz = indptr[K];
    std::vector<char> seen(K, 0);
    if ((int)B.order.size() != K) return "order has the wrong length";
    for (int p = 0; p < K; ++p) {
        const int r = B.order[p];
        if (r < 0 || r >= K || seen[r]) return "order is not a permutation";
        seen[r] = 1;
    }
    if (B.blk_rowptr.front() != 0 || B.blk_rowptr.back() != K) return "blocks do not cover the rows";
    std::vector<int32_t> local(K, -1);
    std::vector<char> ent_seen(nnz, 0), edge_seen(nnz, 0);
    int64_t nupper = 0;
    for (int k = 0; k < K; ++k)
        for (int e = indptr[k]; e < indptr[k + 1]; ++e) nupper += indices[e] > k;
    int64_t slots_used = 0;
    for (int b = 0; b < B.nb(); ++b) {
        const int u0 = B.un_ptr[b], u1 = B.un_ptr[b + 1], nun = u1 - u0;
        if (nun < 1 || nun > MF_UNION) return "union size out of range";
        for (int u = u0; u < u1; ++u) {
            if (local[B.un_cols[u]] >= 0) return "duplicate column in a union";
            local[B.un_cols[u]] = u - u0;
        }
        const int32_t* d = &B.desc[(size_t)b * 8];
        if (d[0] != B.blk_rowptr[b] || d[1] != B.blk_rowptr[b + 1] - B.blk_rowptr[b] || d[5] != nun) return "block record mismatch";
        if (d[2] != B.bptr[d[0]] || d[3] != B.bptr[d[0] + d[1]] - B.bptr[d[0]]) return "block entry range mismatch";
        if (d[1] > BLK_ROWS) return "too many rows in a block";
        if (d[1] > 1 && (d[3] > lim.max_entries_per_block || blk2_lds_need(nun, d[3], lim.entry_bytes) > BLK2_LDS_BYTES) && B.fits_half_tile)
            return "block over its LDS budget";
        for (int q = B.blk_rowptr[b]; q < B.blk_rowptr[b + 1]; ++q) {
            const int r = B.order[q];
            if ((B.bptr[q + 1] - B.bptr[q]) % BLK_CHUNK) return "row not padded to whole chunks";
            int real = 0;
            for (int w = B.bptr[q]; w < B.bptr[q + 1]; ++w) {
                const int pos = (w - B.bptr[q]) % BLK_CHUNK;
                const bool odd_slot = (pos & 4) != 0;
                if ((B.lidx[w] & 1) != (odd_slot ? 1 : 0)) return "staged-row parity does not match the chunk position";
                const int e = B.bepos[w];
                if (e < 0) continue;
                if (e < indptr[r] || e >= indptr[r + 1] || ent_seen[e]) return "blocked entry maps to the wrong CSR position";
                ent_seen[e] = 1;
                if (local[indices[e]] != (int)B.lidx[w]) return "local index does not address the entry's column";
                if (B.bpos[e] != w) return "bpos is not the inverse of bepos";
                ++real;
            }
            if (real != indptr[r + 1] - indptr[r]) return "row lost entries";
            if (local[r] >= 0 && B.self_li[q] != (uint16_t)local[r]) return "self index mismatch";
        }
        // SDDMM slots
        const int s0 = B.sd2_ptr[b], s1 = B.sd2_ptr[b + 1];
        if ((s1 - s0) % SD2_THREADS || s1 - s0 < SD2_THREADS) return "slot range is not whole rounds";
        for (int sidx = s0; sidx < s1; ++sidx) {
            const int e = B.sd2_epos[sidx];
            if (e < 0) continue;
            if (e >= nnz || edge_seen[e]) return "slot maps to a wrong or repeated entry";
            edge_seen[e] = 1;
            ++slots_used;
            const int la = (int)(B.sd2_ab[sidx] & 0xFFFFu), lb = (int)(B.sd2_ab[sidx] >> 16);
            // the entry's row is in this block and (la, lb) address {row, col} in either order
            int row = -1;
            {
                int lo = 0, hi = K;  // row of CSR position e
                while (hi - lo > 1) { const int mid = (lo + hi) / 2; (indptr[mid] <= e ? lo : hi) = mid; }
                row = lo;
            }
            const int col = indices[e];
            if (col <= row) return "slot holds a lower-triangular entry";
            const bool straight = local[row] == la && local[col] == lb, swapped = local[row] == lb && local[col] == la;
            if (!straight && !swapped) return "slot indices do not address the entry's rows";
            // partner lane of the LDS service group
            const int t = (sidx - s0) % SD2_THREADS, lane = t & 63, l32 = lane & 31;
            int partner = -1;
            if (l32 < 8) partner = lane + 24; else if (l32 < 16) partner = lane + 8; else if (l32 < 24) partner = lane - 8; else partner = lane - 24;
            const int ps = sidx - lane + partner;
            if (B.sd2_epos[ps] >= 0) {
                const int pa = (int)(B.sd2_ab[ps] & 0xFFFFu), pb = (int)(B.sd2_ab[ps] >> 16);
                if (((pa ^ la) & 1) == 0 || ((pb ^ lb) & 1) == 0) return "paired lanes read staged rows of equal parity";
            }
        }
        for (int u = u0; u < u1; ++u) local[B.un_cols[u]] = -1;
    }
    for (int64_t e = 0; e < nnz; ++e)
        if (!ent_seen[e]) return "a CSR entry is missing from the blocked arrays";
    if (slots_used != nupper) return "the SDDMM slots do not hold every upper-triangular entry exactly once";
    return "";
}

// Invariants of the matrix-core blocking: every row in exactly one block of at most 64 rows; a block's padded union list holds
// every column of its rows; fpos maps the CSR entries one-to-one into the fragment image, at the lane / word the A operand
// map of v_mfma_f32_32x32x16_bf16 prescribes for (local row, local union index).
inline std::string verify_mfma_blocking(const HostBlocking& B, int K, const std::vector<int32_t>& indptr, const std::vector<int32_t>& indices) {
    const int64_t nnz = indptr[K];
    if (!B.fits_mfma) return "";
    const int nbm = B.nbm(), MT = B.mfma_mt;
    if ((int)B.m_order.size() != K || B.m_rowptr.front() != 0 || B.m_rowptr.back() != K) return "matrix-core blocks do not cover the rows";
    std::vector<char> seen(K, 0);
    for (int p = 0; p < K; ++p) {
        const int r = B.m_order[p];
        if (r < 0 || r >= K || seen[r]) return "matrix-core order is not a permutation";
        seen[r] = 1;
    }
    const int64_t image = (int64_t)B.kbase[nbm] * MT * 512;
    std::vector<char> used((size_t)image, 0);
    std::vector<int32_t> loc(K, -1);
    for (int b = 0; b < nbm; ++b) {
        const int32_t* d = &B.m_desc[(size_t)b * 8];
        const int rows = d[1], nun = d[5];
        if (d[0] != B.m_rowptr[b] || rows != B.m_rowptr[b + 1] - B.m_rowptr[b] || rows < 1 || rows > 32 * MT) return "matrix-core block record mismatch";
        if (nun < 1 || nun > MF_UNION || B.kbase[b + 1] - B.kbase[b] != ((nun + 15) / 16 + MF_KSTEP_PAD - 1) / MF_KSTEP_PAD * MF_KSTEP_PAD) return "matrix-core union size / k-step count mismatch";
        for (int u = 0; u < nun; ++u) {
            const int c = B.m_unfixed[(size_t)b * MF_UNION + u];
            if (c < 0 || c >= K || loc[c] >= 0) return "matrix-core union list has a bad or repeated column";
            loc[c] = u;
        }
        for (int u = nun; u < MF_UNION; ++u)
            if (B.m_unfixed[(size_t)b * MF_UNION + u] != B.m_unfixed[(size_t)b * MF_UNION]) return "matrix-core union padding is not the first column";
        for (int q = B.m_rowptr[b]; q < B.m_rowptr[b + 1]; ++q) {
            const int r = B.m_order[q], rl = q - B.m_rowptr[b];
            for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
                const int li = loc[indices[e]];
                if (li < 0) return "a column of a matrix-core block is missing from its union";
                const int64_t f = B.fpos[e];
                if (f < (int64_t)B.kbase[b] * MT * 512 || f >= (int64_t)B.kbase[b + 1] * MT * 512 || used[(size_t)f]) return "fpos out of its block or repeated";
                used[(size_t)f] = 1;
                const int w = (int)(f & 3), lane = (int)((f >> 2) & 63), half = (int)((f >> 8) & 1);
                const int64_t tile = f >> 9;  // (k-step, row tile)
                const int mt = (int)(tile % MT), ks = (int)(tile / MT) - B.kbase[b];
                if (mt != rl / 32 || (lane & 31) != rl % 32 || ks * 16 + 8 * (lane >> 5) + 4 * half + w != li) return "fpos does not address (row, union index) in the A operand map";
            }
        }
        for (int u = 0; u < nun; ++u) loc[B.m_unfixed[(size_t)b * MF_UNION + u]] = -1;
    }
    for (int64_t e = 0; e < nnz; ++e)
        if (B.fpos[e] < 0) return "a CSR entry has no fragment position";
    // SDDMM tile lists: every undirected edge exactly once -- listed with the endpoint that comes first in the blocked order as its
    // row --, in the tile its (row, union index) falls into, with the CSR position of its mirror; no entry below a block's first tile
    std::vector<char> seen_e((size_t)nnz, 0);
    std::vector<int32_t> mpos(K);
    for (int q = 0; q < K; ++q) mpos[B.m_order[q]] = q;
    int64_t listed = 0;
    for (int b = 0; b < nbm; ++b) {
        const int32_t* d = &B.m_desc[(size_t)b * 8];
        const int nun = d[5], nt = (nun + 31) / 32;
        if (B.m_tbase[b + 1] - B.m_tbase[b] != nt * MT) return "tile count of a matrix-core block is off";
        for (int t = B.m_tbase[b]; t < B.m_tbase[b + 1]; ++t) {
            const int ut = (t - B.m_tbase[b]) / MT, mt = (t - B.m_tbase[b]) % MT;
            if (ut < d[6] && B.m_tptr[t + 1] != B.m_tptr[t]) return "a tile below the block's first SDDMM tile holds entries";
            int bits = 0;  // the accumulator mask of the tile marks exactly the listed entries
            for (int l = 0; l < 64; ++l) bits += __builtin_popcount((unsigned)B.m_tmask[(size_t)t * 64 + l]);
            if (B.m_tmask.size() < (size_t)(t + 1) * 64 || bits != B.m_tptr[t + 1] - B.m_tptr[t]) return "accumulator mask of a tile does not match its list";
            for (int w = B.m_tptr[t]; w < B.m_tptr[t + 1]; ++w, ++listed) {
                const int e = B.m_tepos[w], rl = 32 * mt + (B.m_trc[w] >> 5), li = 32 * ut + (B.m_trc[w] & 31);
                {
                    const int r5 = B.m_trc[w] >> 5, c5 = B.m_trc[w] & 31;
                    if (!((B.m_tmask[(size_t)t * 64 + c5 + 32 * ((r5 >> 2) & 1)] >> ((r5 & 3) + 4 * (r5 >> 3))) & 1)) return "accumulator mask misses a listed entry";
                }
                if (e < 0 || e >= nnz || seen_e[e]) return "tile list holds a bad or repeated entry";
                seen_e[e] = 1;
                if (rl >= d[1] || li >= nun) return "tile list entry outside its block";
                const int r = B.m_order[d[0] + rl];
                if (e < indptr[r] || e >= indptr[r + 1] || indices[e] != B.m_unfixed[(size_t)b * MF_UNION + li] || indices[e] == r) return "tile list entry addresses the wrong (row, column)";
                if (mpos[indices[e]] <= mpos[r]) return "tile list entry is not the first endpoint's";
                const int m = B.m_temir[w], c = indices[e];
                if (m < indptr[c] || m >= indptr[c + 1] || indices[m] != r || seen_e[m]) return "mirror of a tile list entry is wrong";
                seen_e[m] = 1;
            }
        }
    }
    if (2 * listed != nnz - K) return "the tile lists do not hold every undirected edge once";
    if (B.m_nedges != listed || (int64_t)B.m_e2w.size() != nnz) return "tile-order slot map has the wrong size";
    for (int k = 0; k < K; ++k)
        for (int e = indptr[k]; e < indptr[k + 1]; ++e) {
            const int w = B.m_e2w[e];
            if (indices[e] == k) { if (w != B.m_nedges + k) return "diagonal slot is off"; }
            else if (w < 0 || w >= listed || (B.m_tepos[w] != e && B.m_temir[w] != e)) return "edge slot does not list the entry";
        }
    return "";
}

}  // namespace mmw
