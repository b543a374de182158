// Host-side runtime plumbing: error strings, RAII device buffers, launch geometry.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
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

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        MMW_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        return MMW_OK;
    }
    int upload(const std::vector<T>& h, hipStream_t st) {
        MMW_TRY(alloc(h.size()));
        if (!h.empty()) MMW_HIP(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
        return MMW_OK;
    }
    // upload a double host vector converted to T
    template <typename S> int upload_cast(const std::vector<S>& h, hipStream_t st) {
        std::vector<T> tmp(h.size());
        for (size_t i = 0; i < h.size(); ++i) tmp[i] = (T)h[i];
        MMW_TRY(alloc(tmp.size()));
        if (!tmp.empty()) {
            MMW_HIP(hipMemcpyAsync(p, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, st));
            MMW_HIP(hipStreamSynchronize(st));  // tmp dies at scope exit
        }
        return MMW_OK;
    }
};

inline int grid_rows(int rows) {  // one wavefront per row, 4 rows per workgroup, grid-stride beyond the cap
    int g = (rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (g < 1) g = 1;
    return g > MAX_PART ? MAX_PART : g;
}
inline int grid_elems(size_t n) {
    size_t g = (n + BLOCK - 1) / BLOCK;
    if (g < 1) g = 1;
    return (int)(g > 2048 ? 2048 : g);
}

}  // namespace mmw
