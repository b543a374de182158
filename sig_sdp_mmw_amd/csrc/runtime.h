// Host-side runtime plumbing: error strings, RAII device buffers, launch geometry.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mmw_hip.h"
#include "device_utils.h"

namespace mmw {

inline std::string& last_error_ref() {
    static thread_local std::string e;
    return e;
}
inline int fail(int code, const std::string& msg) {
    last_error_ref() = msg;
    return code;
}

#define MMW_HIP(call)                                                                                         \
    do {                                                                                                      \
        hipError_t err__ = (call);                                                                            \
        if (err__ != hipSuccess) {                                                                            \
            char buf__[512];                                                                                  \
            snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, \
                     __LINE__);                                                                               \
            return ::mmw::fail(MMW_ERR_HIP, buf__);                                                           \
        }                                                                                                     \
    } while (0)

#define MMW_TRY(expr)              \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != MMW_OK) return rc__; \
    } while (0)

// Per-device one-time setup.  A kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) and the CU count belong to the
// CURRENT device, and handles of one process may sit on different devices and be driven from different host threads (bench.py's
// resident instances, the speculative search): both are keyed by the device id and guarded.
inline int set_max_lds(const void* fn, int bytes) {
    int dev = 0;
    MMW_HIP(hipGetDevice(&dev));
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int>* done = new std::map<std::pair<const void*, int>, int>;  // never destroyed (runtime teardown order)
    std::lock_guard<std::mutex> g(mu);
    auto it = done->find({fn, dev});
    if (it != done->end() && it->second >= bytes) return MMW_OK;
    MMW_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    (*done)[{fn, dev}] = bytes;
    return MMW_OK;
}
inline int device_cus() {  // compute units of the current device (256 on MI355X)
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    c = 256;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cache[dev].store(c, std::memory_order_relaxed);
    return c;
}

// Page-locked host staging for the large transfers of a handle (the factor handed back, the rounding's inputs).  An asynchronous
// copy to or from pageable memory makes the runtime pin the caller's pages and unpin them later, at a moment of its own choosing:
// measured as 20 - 40 ms stalls in front of an unrelated launch of the next probe.  Copies go through this buffer instead.
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    int ensure(size_t bytes) {
        if (p && bytes <= cap) return MMW_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        if (bytes == 0) bytes = 1;
        MMW_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        cap = bytes;
        return MMW_OK;
    }
};

// A small pool of such buffers for the one-off transfers (uploads at creation, API reads): leased for the duration of one copy.
// The pool and its buffers live until the process ends (no destructor races with the HIP runtime's own teardown).
struct StagePool {
    std::mutex m;
    std::vector<PinnedBuf*> idle;
    PinnedBuf* get() {
        std::lock_guard<std::mutex> g(m);
        if (!idle.empty()) { PinnedBuf* b = idle.back(); idle.pop_back(); return b; }
        return new PinnedBuf;
    }
    void put(PinnedBuf* b) { std::lock_guard<std::mutex> g(m); idle.push_back(b); }
};
inline StagePool& stage_pool() { static StagePool* pool = new StagePool; return *pool; }
struct StageLease {
    PinnedBuf* b;
    StageLease() : b(stage_pool().get()) {}
    ~StageLease() { stage_pool().put(b); }
    StageLease(const StageLease&) = delete;
    StageLease& operator=(const StageLease&) = delete;
};
constexpr size_t STAGE_MIN_BYTES = 64 << 10;  // smaller copies go straight: the runtime bounces them through its own pinned chunks
// host -> device and back, synchronous on `st`; large ones through a leased page-locked buffer
inline int copy_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t st) {
    if (bytes == 0) return MMW_OK;
    if (bytes < STAGE_MIN_BYTES) {
        MMW_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        MMW_HIP(hipStreamSynchronize(st));
        return MMW_OK;
    }
    StageLease L;
    MMW_TRY(L.b->ensure(bytes));
    memcpy(L.b->p, src_host, bytes);
    MMW_HIP(hipMemcpyAsync(dst_dev, L.b->p, bytes, hipMemcpyHostToDevice, st));
    MMW_HIP(hipStreamSynchronize(st));
    return MMW_OK;
}
inline int copy_d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t st) {
    if (bytes == 0) return MMW_OK;
    if (bytes < STAGE_MIN_BYTES) {
        MMW_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        MMW_HIP(hipStreamSynchronize(st));
        return MMW_OK;
    }
    StageLease L;
    MMW_TRY(L.b->ensure(bytes));
    MMW_HIP(hipMemcpyAsync(L.b->p, src_dev, bytes, hipMemcpyDeviceToHost, st));
    MMW_HIP(hipStreamSynchronize(st));
    memcpy(dst_host, L.b->p, bytes);
    return MMW_OK;
}

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;    // elements in use
    size_t cap = 0;  // elements allocated
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        cap = 0;
    }
    // Contents are undefined afterwards.  An allocation that is large enough is kept: hipFree waits for the whole device (every
    // stream of every handle), so a handle that shrinks its blocks for the next slot count must not free them.
    int alloc(size_t count) {
        static const bool exact = getenv("MMW_DEVBUF_EXACT") != nullptr;
        if (p && count <= cap && !exact) {
            n = count;
            return MMW_OK;
        }
        release();
        n = count;
        if (count == 0) count = 1;
        MMW_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        cap = count;
        return MMW_OK;
    }
    int upload(const std::vector<T>& h, hipStream_t st) {
        MMW_TRY(alloc(h.size()));
        return copy_h2d(p, h.data(), h.size() * sizeof(T), st);  // synchronous: the source may be a short-lived host vector
    }
    // upload a double host vector converted to T
    template <typename S> int upload_cast(const std::vector<S>& h, hipStream_t st) {
        std::vector<T> tmp(h.size());
        for (size_t i = 0; i < h.size(); ++i) tmp[i] = (T)h[i];
        MMW_TRY(alloc(tmp.size()));
        return copy_h2d(p, tmp.data(), tmp.size() * sizeof(T), st);
    }
};

// Optional per-kernel-class device timers (HIP events on the solver's own stream).  Off by default:
// when off, begin/end cost nothing and no event is recorded.
enum { KT_SPMM = 0, KT_SDDMM = 1, KT_DUAL = 2, KT_LOSS = 3, KT_KRYLOV_VEC = 4, KT_SKETCH = 5, KT_PROJECT = 6, KT_GREEDY = 7,
       KT_FACTOR = 8, KT_NSLOT = 9 };
struct KernelTimers {
    bool on = false;
    hipStream_t st = nullptr;
    struct Rec { int slot; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double total_us[KT_NSLOT] = {0};
    double count[KT_NSLOT] = {0};
    int cur = -1;
    ~KernelTimers() {
        for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
    int get(hipEvent_t* e) {
        if (!pool.empty()) { *e = pool.back(); pool.pop_back(); return MMW_OK; }
        MMW_HIP(hipEventCreate(e));
        return MMW_OK;
    }
    int begin(int slot) {
        if (!on) return MMW_OK;
        Rec r; r.slot = slot;
        MMW_TRY(get(&r.a)); MMW_TRY(get(&r.b));
        MMW_HIP(hipEventRecord(r.a, st));
        recs.push_back(r);
        return MMW_OK;
    }
    int end() {
        if (!on || recs.empty()) return MMW_OK;
        if (attached) { attached = false; return MMW_OK; }  // the launch itself records the pair (begin_attached)
        MMW_HIP(hipEventRecord(recs.back().b, st));
        return MMW_OK;
    }
    // A bracket of ONE launch whose events the launch carries itself (hipExtLaunchKernelGGL's start / stop events: the dispatch's own
    // begin and end, what a kernel trace reports) instead of two marker packets around it, which add the marker's and the dispatch's
    // latency (2-3 us) to a 17 us kernel.  `attach` is set for the profiling mode that times the shipped path as launched.
    bool attach = false, attached = false;
    int begin_attached(int slot, hipEvent_t* a, hipEvent_t* b) {
        *a = nullptr; *b = nullptr;
        if (!on) return MMW_OK;
        if (!attach) return begin(slot);
        Rec r; r.slot = slot;
        MMW_TRY(get(&r.a)); MMW_TRY(get(&r.b));
        recs.push_back(r);
        *a = r.a; *b = r.b;
        attached = true;
        return MMW_OK;
    }
    int flush() {  // call after the stream has been synchronised
        for (auto& r : recs) {
            float ms = 0;
            MMW_HIP(hipEventElapsedTime(&ms, r.a, r.b));
            total_us[r.slot] += ms * 1e3; count[r.slot] += 1;
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
        return MMW_OK;
    }
    void clear() { for (int i = 0; i < KT_NSLOT; ++i) { total_us[i] = 0; count[i] = 0; } }
};

inline int grid_rows(int rows) {  // one wavefront per row, 4 rows per workgroup, grid-stride beyond the cap
    int g = (rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (g < 1) g = 1;
    return g > ROW_GRID_MAX ? ROW_GRID_MAX : g;  // short rows are latency chains: one row per wave beats grid-striding
}
inline int grid_slabs(int rows) { return std::min(grid_rows(rows), MAX_PART); }  // kernels that write a Dpad-wide slab per workgroup
inline int grid_elems(size_t n) {
    size_t g = (n + BLOCK - 1) / BLOCK;
    if (g < 1) g = 1;
    return (int)(g > 2048 ? 2048 : g);
}

}  // namespace mmw
