// fp32 operands for the bf16 matrix pipe: x = h + m + l EXACTLY, three bf16 pieces of 8 significand bits each (round-to-nearest
// pieces: |m| <= 2^-8 |x|, |l| <= 2^-17 |x|; exact for every finite fp32 whose low pieces do not underflow, |x| > 2^-100).
// A product w x is then accumulated in fp32 by v_mfma_f32_32x32x16_bf16 as the six piece products of combined order <= 2,
//     wl xh + wh xl + wm xm + (wm xh + wh xm) + wh xh ,
// the dropped ones (wm xl, wl xm, wl xl) being at most 2^-24 |w x| (worst case, typically 2^-27; tests/test_bx_split_math.py) --
// no more than the rounding of the fp32 product itself: an fp32
// contraction at 6 x 32 cycles per 32 x 32 x 16 instead of the 8 x 64 of v_mfma_f32_32x32x2_f32 (the fp32-input MFMA runs at the
// fp32 VECTOR rate).  Measured against fp64 (scripts/bx6_bench.hip, K = 1536..6144): max error 0.06-0.15x that of an fp32 fma chain
// over the same K, because the MFMA rounds once per 16 products.  An infinite input gives NaN (inf - inf in the split).
#pragma once
#include <hip/hip_runtime.h>

namespace wmar {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using bxf32x2 = __attribute__((ext_vector_type(2))) float;

// plain v_sub_f32: the SLP vectoriser pairs these into v_pk_add_f32, which costs more beside MFMAs
__device__ __forceinline__ float bx_fsub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned bx_pk(float a, float b) {
    const bxf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32, round to nearest even
}
// two fp32 -> packed bf16 pairs (low half = first value) of the three pieces
__device__ __forceinline__ void bx_split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = bx_pk(a, b);
    const float ra = bx_fsub(a, __uint_as_float(h << 16)), rb = bx_fsub(b, __uint_as_float(h & 0xffff0000u));
    m = bx_pk(ra, rb);
    const float sa = bx_fsub(ra, __uint_as_float(m << 16)), sb = bx_fsub(rb, __uint_as_float(m & 0xffff0000u));
    l = bx_pk(sa, sb);
}

// Activation planes Xq[ku][mt][piece][lane] (16 bytes = 8 bf16 per lane): lane holds row m = 32 mt + lane % 32, features
// k = 16 ku + 8 (lane / 32) + 0..7 -- the B operand of v_mfma_f32_32x32x16_bf16.  A producer thread owns four consecutive
// features (k = 8 kb + 4 hf + 0..3, the float4 of the packed fp32 layout) and stores its half of the 16 bytes of each piece.
__device__ __forceinline__ void bx_store_planes4(u32x4* __restrict__ Xq, int MT, int kb, int hf, int mt, int m32, const float4 v) {
    unsigned h0, h1, m0, m1, l0, l1;
    bx_split2(v.x, v.y, h0, m0, l0);
    bx_split2(v.z, v.w, h1, m1, l1);
    u32x2* p = (u32x2*)(Xq + ((long long)((kb >> 1) * MT + mt) * 3) * 64 + m32 + 32 * (kb & 1)) + hf;
    p[0] = u32x2{h0, h1}; p[128] = u32x2{m0, m1}; p[256] = u32x2{l0, l1};
}
__device__ __forceinline__ void bx_split8(const float4 a, const float4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
    unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    bx_split2(a.x, a.y, h0, m0, l0);
    bx_split2(a.z, a.w, h1, m1, l1);
    bx_split2(b.x, b.y, h2, m2, l2);
    bx_split2(b.z, b.w, h3, m3, l3);
    const u32x4 uh = {h0, h1, h2, h3}, um = {m0, m1, m2, m3}, ul = {l0, l1, l2, l3};
    h = __builtin_bit_cast(bf16x8, uh); m = __builtin_bit_cast(bf16x8, um); l = __builtin_bit_cast(bf16x8, ul);
}

}  // namespace wmar
