/*
 * wmar_math.h -- the exact scalar arithmetic of the sampling stage.
 *
 * The sampling stage of wmar (deps/taming/modules/transformer/mingpt.py:348-363
 * plus the HF TopK/TopP warpers it calls at :355-357) decides TOKEN IDS from
 * floating-point intermediates (softmax, cumulative sums).  PyTorch's CPU
 * kernels for those use vector-width-dependent reductions and a vendor exp, so
 * they cannot be reproduced bit-for-bit on another machine.  This header pins
 * ONE arithmetic that is reproducible everywhere:
 *
 *   - exp is a fixed fp32 polynomial evaluated with explicit fmaf (IEEE-754
 *     fused multiply-add gives the same bits on x86 and gfx950);
 *   - every sum is an ORDER-INDEPENDENT integer sum of 2^-46 fixed-point
 *     values, so a wave-parallel reduction and a sequential C loop agree.
 *
 * It is included by the HIP kernels (wmar_amd/csrc) and, independently, by the
 * CPU checker under oracle/.  Compile every translation unit that includes it
 * with -ffp-contract=off.
 */
#ifndef WMAR_MATH_H
#define WMAR_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define WMAR_HD __host__ __device__ __forceinline__
#else
#define WMAR_HD static inline
#endif

#define WMAR_FX_BITS 46
#define WMAR_FX_SCALE 70368744177664.0            /* 2^46  */
#define WMAR_FX_INV 1.4210854715202004e-14        /* 2^-46 */

/* Monotone map fp32 -> u32: a < b (as floats, no NaN)  <=>  key(a) < key(b).
 * -0.0f and +0.0f get adjacent, distinct keys (-0 < +0); torch's `<` treats
 * them as equal, which only matters for logits that are exactly +-0. */
WMAR_HD uint32_t wmar_f32_key(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

WMAR_HD float wmar_key_f32(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float x;
    memcpy(&x, &b, 4);
    return x;
}

/* exp(x) for x <= 0 (the softmax argument after max subtraction).
 * x < -87 (incl. -inf) returns exactly 0: nothing denormal is ever produced.
 * Max error ~1 ulp (Cephes expf coefficients). */
WMAR_HD float wmar_expf(float x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (!(x >= -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float rr = r * r;
    p = fmaf(p, rr, r);
    p = p + 1.0f;
    int32_t ni = (int32_t)n;
    uint32_t b;
    memcpy(&b, &p, 4);
    b += ((uint32_t)ni) << 23;
    memcpy(&p, &b, 4);
    return p;
}

/* fixed-point image of a non-negative fp32 value <= 1 (truncation). */
WMAR_HD uint64_t wmar_fx(float v) {
    return (uint64_t)((double)v * WMAR_FX_SCALE);
}

/* fp32 value of a fixed-point sum. */
WMAR_HD float wmar_fx_to_f32(uint64_t s) {
    return (float)((double)s * WMAR_FX_INV);
}

#endif /* WMAR_MATH_H */
