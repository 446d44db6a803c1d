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

// (double)a * (double)b, kept OUT of a fused multiply-add.  A float product has 48 significant bits, so it is exact in fp64 and
// `acc + prod_f64(a, b)` is bit for bit what `fma(a, b, acc)` returns -- but the instruction is v_mul_f64 + v_add_f64, not v_fmac_f64.
// Why (round 3, gfx950): the sums of squares behind the LayerNorm statistics were chains of dependent v_fmac_f64 (x0^2, += x1^2,
// += x2^2, += x3^2).  In k_qkvx_bx such a chain returned, about once per 25,000 launches, a result that was off by roughly one x^2
// term in one 16-lane pass of the wave (rows 48..63 of the batch; the plain sum, a v_add_f64 chain over the same registers, stayed
// exact) -- a 1e-5 run-to-run difference of those rows' logits, enough to flip a sampled token once in ~10 full generations.  The rate
// moved between 0 % and 100 % of 256-step passes when nothing but the kernel's code address changed (s_nop padding in front of it),
// i.e. it depends on how the instruction stream is fetched, and it was 0 in 108 passes at six paddings with the products taken out
// of the fmac chain (and with s_nop 3 between the fmacs).  Found and bisected with scripts/stress_logits.py / stress_kv.py; every
// fp64 accumulation of float products in this library goes through this helper since.
#ifdef __HIPCC__
__device__ __forceinline__ double prod_f64(float a, float b) {
#pragma clang fp contract(off)
    double p = (double)a * (double)b;
    asm volatile("" : "+v"(p));      // opaque to the optimiser: the product cannot be re-fused into the following add
    return p;
}
// E[x^2] - mean^2 in fp64 without a fused multiply-add (same reason; differs from the fused form by at most one rounding of mean^2,
// i.e. ~1e-16 relative, far below the float the result is converted to)
__device__ __forceinline__ double var_f64(double ex2, double mean) {
#pragma clang fp contract(off)
    double m2 = mean * mean;
    asm volatile("" : "+v"(m2));
    return ex2 - m2;
}
// 1 / n for a count n < 2^24 (row widths, pixel counts) without the fp64 division's Newton chain of fused multiply-adds: a float
// reciprocal refined by two Newton steps in unfused fp64 arithmetic (n * r is exact: 24 + 24 bits) -- within one ulp of 1.0 / n.
__device__ __forceinline__ double inv_count_f64(double n) {
#pragma clang fp contract(off)
    double r = (double)(1.0f / (float)n);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        double e = n * r;
        asm volatile("" : "+v"(e));
        e = 1.0 - e;
        double c = r * e;
        asm volatile("" : "+v"(c));
        r = r + c;
    }
    return r;
}
// 1 / sqrt(v), v > 0, the same way: float estimate, three Newton steps y <- y (1.5 - 0.5 v y^2) in unfused fp64 (relative error
// 6e-8 -> 5e-15 -> 4e-29: the fp64 rounding of the last step is what remains)
__device__ __forceinline__ double rsqrt_f64(double v) {
#pragma clang fp contract(off)
    double y = (double)rsqrtf((float)v);
    const double hv = 0.5 * v;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        double y2 = y * y;
        asm volatile("" : "+v"(y2));
        double t = hv * y2;
        asm volatile("" : "+v"(t));
        t = 1.5 - t;
        double yn = y * t;
        asm volatile("" : "+v"(yn));
        y = yn;
    }
    return y;
}
__device__ __forceinline__ double sq4_f64(const float4& v) {        // ((x^2 + y^2) + z^2) + w^2: the order the fused chain had
    return ((prod_f64(v.x, v.x) + prod_f64(v.y, v.y)) + prod_f64(v.z, v.z)) + prod_f64(v.w, v.w);
}
#endif


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
