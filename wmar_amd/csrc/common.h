// Shared host-side helpers for libwmar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/wmar_hip.h"

namespace wmar {

void set_error(const char* fmt, ...);

#define WMAR_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ::wmar::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                       \
            return WMAR_EHIP;                                                                  \
        }                                                                                      \
    } while (0)

#define WMAR_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ::wmar::set_error(__VA_ARGS__); \
            return WMAR_EINVAL;            \
        }                                  \
    } while (0)

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return WMAR_EHIP;
    }
    return WMAR_OK;
}

// Context-row selection shared by the logit processor, the fused sampler and the
// generation graph (gentime_watermark.py:233-263).  Returns -1 when the row must be skipped.
__device__ __forceinline__ long long ctx_row(const long long* p, long long t, int seed_mode, int h, int S) {
    if (seed_mode == WMAR_SEED_FIXED) return 0;
    if (seed_mode == WMAR_SEED_LINEAR) {
        if (t < h) return -1;
        long long s = 0;
        for (int i = 0; i < h; ++i) s += p[t - h + i];
        return s;
    }
    if (h == 3) {
        if (t < S + 1) return -1;
        return p[t - S - 1] + p[t - S] + p[t - 1];
    }
    if (h == 1) {
        if (t < 1) return -1;
        if (t % S == 0) {
            if (S == 1 || t < S) return -1;
            return p[t - S];
        }
        return p[t - 1];
    }
    return -1;
}

}  // namespace wmar
