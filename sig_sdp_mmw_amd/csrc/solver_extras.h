// Rounding, epilogue factor and duality-gap routines that hang off a solver handle.
#pragma once
#include "pattern.h"
#include "runtime.h"

namespace mmw {

template <typename T> struct Extras {
    hipStream_t st = nullptr;
    const HostPattern* H = nullptr;
    int K = 0;
    int init(hipStream_t s, const HostPattern* h, int K_) {
        st = s; H = h; K = K_;
        return MMW_OK;
    }
    int gap(double*) { return fail(MMW_ERR_STATE, "mmw_gap: not built yet"); }
    int factor(int32_t, double*, uint64_t) { return fail(MMW_ERR_STATE, "mmw_factor: not built yet"); }
    int read_factor(double*, int64_t) { return fail(MMW_ERR_STATE, "mmw_factor: not built yet"); }
    int round(int32_t, int32_t, const double*, int32_t, const double*, int32_t*, int32_t*) {
        return fail(MMW_ERR_STATE, "mmw_round: not built yet");
    }
};

}  // namespace mmw
