// k_xg (DEV TOOL, round 4 -- not part of libwmar_hip.so): a 64-row fp32 GEMM on the bf16 matrix pipe whose split-K reduction stays
// INSIDE an XCD.  Built to test one idea; the measured outcome (profiles/r04_xg_*.log) is that it does NOT beat the
// shipped kernels, so it lives here with its bench (scripts/xg_bench.hip) as evidence and as the starting point of anything that
// needs a cheap intra-XCD barrier.
//
// The idea.  Every decode GEMM of rounds 1-3 fits  t ~ 1 us + (bytes through ONE CU's vector-memory path) / 33 GB/s : an L2-hit
// activation byte costs what an HBM weight byte costs.  Whole-K / deep-K tiles make every CU read most of the 64 x K activation (FC1:
// 393 KB beside 147 KB of weights).  Shallow-K wide tiles (96 columns x 384 k: 98 KB + 147 KB) halve the bytes per CU but need a split-K
// reduction; a second launch or a fold in the consumer returns the gain.  Here:
//   * block b runs on an XCD that depends on b % 8 only (measured: XCC id = (b + r) % 8 with r rotating with the launches before it,
//     32 blocks of a 256-block grid per XCD every time); "XCD x" owns output columns [x N/8, (x+1) N/8) for ALL of K;
//   * phase 1: CU c of the XCD takes (column group g, K slice s), c = g S + s: its waves split the slice's 16-k steps, split fp32
//     weights (and fp32 activations, or read them pre-split: -DXG_XQ) into bf16 pieces in registers, meet in LDS in a fixed order and
//     park ONE partial tile per workgroup in the XCD's L2 (plain stores + s_waitcnt vmcnt(0): a store is acknowledged by the L2);
//   * an XCD-LOCAL barrier: arrival counter and generation word are L2 atomics WITHOUT sc1 (they never leave the XCD), polled with a
//     returning L2 atomic (inline asm: a compiler-level fetch_or(0) folds into an L1-cached load and spins forever);
//   * phase 2: wave (c, w) reduces the S partials of one (16-column group, row tile) in slice order and finishes it (LayerNorm
//     algebra, bias, GELU / residual add / q, k, v with the KV-cache append).
// Measured on MI355X (FC1 shape 6144 x 1536, 64 rows, 12 distinct weight buffers):
//   * the XCD barrier costs 1.8 us (launch + barrier + phase 2 5.9 us against launch + phase 2 4.1 us) -- not the 17 us of the
//     device-wide barrier of round 1, and less than the ~3.5 us a second launch adds; results bit-identical to the two-launch form;
//     (a first version raised a "placement" flag with an sc1 store from 7/8 of the blocks: those contended stores alone cost 8 us);
//   * phase 1 alone 13-15 us, fused 16.4-18 us: no better than k_fc1x (17 us) / k_gemm FC2 (15.4 us).  Stamped timeline: the initial
//     loads take 3-5 k ticks to be ACCEPTED (the CU's memory pipeline holds ~64 KB of requests and blocks the issuing wave, MFMAs
//     included), the first two steps wait for data (2.2-3.4 k ticks each), later steps run 1.5-2.3 k ticks against 1152 of MFMA
//     time: with fp32 activations the step is VALU-bound (11 operations per split pair, ~400 VALU instructions per 36 MFMAs), with
//     pre-split activations (-DXG_XQ) it is bound by the 294 KB per CU at ~33 GB/s.  Ablations: no MFMAs 9.7 us, no split 12.8, neither
//     7.8, neither and no reloads 6.4 (= launch + first loads + LDS reduction + store).  Eight waves (two per SIMD) 14.1 us; a deeper
//     raw ring (all six steps requested up front) 16.7 us: the waves then sit blocked at the loads.
// Results never depend on timing: every sum has a fixed order.
//
// Layouts: weights k_pack_bx order Wq[n/32][k/16][half][lane] float4; activations "Xh"[k/16][row tile][lane][8 floats]: lane holds row
// 32 mt + lane % 32, features 16 ku + 8 (lane / 32) + 0..7 -- split in registers it IS the B operand of the bf16 MFMA; phase 2 writes
// its outputs in the same layout (32 contiguous bytes per lane).
#pragma once
#include "../wmar_amd/csrc/common.h"
#include "../wmar_amd/csrc/bx_split.h"

namespace wmar {

using xg_f32x16 = __attribute__((ext_vector_type(16))) float;
using xg_f32x4 = __attribute__((ext_vector_type(4))) float;

enum { XG_EPI_GELU = 1, XG_EPI_RESID = 2, XG_EPI_QKV = 3 };

struct XgArgs {
    const float4* Wq;          // k_pack_bx layout (LayerNorm gamma folded in for the LN epilogues)
    const float4* Xh;          // [KU][2][64][2] float4
    const u32x4* Xq;           // XG_XQ builds: the activation as bf16 pieces, planes [KU][2][3][64] (bx_store_planes4)
    float4* part;              // [8][S][TX*4][2][64] float4: per XCD, K slice, column octet, row tile
    double2* statp;            // [8][S][64]: (sum, sum of squares) of the rows of X over K slice s (LN epilogues), one copy per XCD
    unsigned* sync;            // [8][64] words: word 0 arrival count, word 32 generation (two 128-byte lines per XCD)
    unsigned* fail;            // [0] placement mismatch, [1] barrier timeout
    int KU, S, G, TX, K;       // K / 16; K slices and column groups per XCD (G * S <= 32); column tiles per XCD (N / 256)
    double invK;
    const float* bias;         // [N] (LN epilogues: bias + W beta)
    const float* c1;           // [N] row sums of the gamma-folded weights (LN epilogues)
    const float4* Xres;        // XG_EPI_RESID: the residual stream, Xh layout over the OUTPUT features
    float4* out;               // XG_EPI_GELU / XG_EPI_RESID: Xh layout over the output features
    // XG_EPI_QKV
    float* qbuf;               // [64][D] row-major
    float* kcache; float* vcache;   // [B][H][Tmax][hd] of this layer
    const int* pos_dev;
    int D, H, hd, Tmax, B;
    unsigned long long* trace; // dev only (XG_TRACE): 6 words per workgroup
};

__device__ __forceinline__ float4 xg_ld_nt(const float4* p) {
    const xg_f32x4 v = __builtin_nontemporal_load((const xg_f32x4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float xg_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// the current value of a word as THIS XCD's L2 holds it: a returning atomic without sc1 (a compiler-level fetch_or(0) is folded into
// a load that may hit the L1)
__device__ __forceinline__ unsigned xg_l2_read(unsigned* p) {
    unsigned v; const unsigned z = 0;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(z) : "memory");
    return v;
}

static __global__ void k_xg_probe(unsigned* o) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) o[blockIdx.x] = x & 15;
}

#ifndef XG_ABL
#define XG_ABL 0      // dev ablations: 1 no MFMAs, 2 no operand split, 4 no reloads inside the loop, 8 no row statistics
#endif
#if XG_ABL & 1
#define WMAR_XG_MFMA(A, B, C) C[0] += (float)A[0] + (float)B[0]
#else
#define WMAR_XG_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#endif
__device__ __forceinline__ void xg_split8(const float4 a, const float4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
#if XG_ABL & 2
    const u32x4 q = {__float_as_uint(a.x) ^ __float_as_uint(a.y), __float_as_uint(a.z) ^ __float_as_uint(a.w), __float_as_uint(b.x) ^ __float_as_uint(b.y), __float_as_uint(b.z) ^ __float_as_uint(b.w)};
    h = __builtin_bit_cast(bf16x8, q); m = h; l = h;
#else
    bx_split8(a, b, h, m, l);
#endif
}

// NT column tiles x (NW waves x PER 16-k steps) per workgroup; S_ = K slices (compile time: every phase-2 load is issued up front).
// NW = 4: one wave per SIMD, the NEXT step's operands are split between this step's MFMAs (two sets of pieces).
// NW = 8: two waves per SIMD with 256 registers each: one set of pieces (split, then multiply); the partner wave fills the gaps -- a
// wave blocked at a load (the CU's memory pipeline accepts ~64 KB of requests) or waiting for one cannot issue its own MFMAs.
template <int NT, int PER, int EPI, int MODE, int S_, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_xg(XgArgs a) {
    constexpr bool LN = EPI != XG_EPI_RESID;
    constexpr int ROWS = NT * 8;                 // float4 rows (tile, row tile, register group) of the workgroup's partial tile
    constexpr bool DB = NW == 4;
#ifndef XG_RW
#define XG_RW 3
#endif
#ifndef XG_POLL
#define XG_POLL 1
#endif
    constexpr int RW = (DB ? XG_RW : 2) < PER ? (DB ? XG_RW : 2) : PER;   // raw operand ring
    __shared__ __attribute__((aligned(16))) float4 red[MODE == 2 ? 1 : 4][MODE == 2 ? 1 : ROWS][64];
    __shared__ double2 sred[NW][2][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = (int)blockIdx.x & 7, c = (int)blockIdx.x >> 3;
    const int g = c / S_, s = c - g * S_;
    const bool active = c < a.G * S_;
    const int TX4 = a.TX * 4;
    unsigned gen0 = 0, xid = 0;
    if (MODE == 0 && threadIdx.x == 0) {
        // Placement evidence: the XCD barrier below only needs the 32 blocks with the same blockIdx % 8 to share an XCD (the XCC id a
        // group lands on rotates with the launches before it).  Block c == 0 of the group publishes its XCC id, every block compares
        // after the barrier (a block on a foreign XCD would normally not even get there: its arrival counts in another L2).
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
        xid &= 15u;
        if (c == 0) __hip_atomic_store(a.sync + xcd * 64 + 16, xid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        gen0 = __hip_atomic_load(a.sync + xcd * 64 + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load sc1: served by the L2
    }

    if constexpr (MODE != 2) if (active) {
        // ------------------------------------------------------------------------------------------------ phase 1
#ifdef XG_STAMP
        unsigned long long ts[PER + 5];
#define XG_T(I) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ts[I]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
        XG_T(0)
#else
#define XG_T(I)
#endif
        const int u0 = s * (NW * PER) + w * PER;                         // host: KU == S * NW * PER
        const int tile0 = xcd * a.TX + g * NT;
        xg_f32x16 acc[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
        const float4* wp = a.Wq + ((long long)tile0 * a.KU + u0) * 128 + lane;
        const long long wt = (long long)a.KU * 128;                     // next column tile
        const float4* xp = a.Xh + ((long long)u0 * 128 + lane) * 2;
        float4 wr[RW][NT][2], xr[RW][2][2];
#ifdef XG_XQ
        const u32x4* xqp = a.Xq + (long long)u0 * 384 + lane;
        u32x4 xq[2][2][3];
#define WMAR_XG_LOADXQ(SL, U) { _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int p = 0; p < 3; ++p) xq[SL][i][p] = xqp[(long long)(U) * 384 + (i * 3 + p) * 64]; }
#define WMAR_XG_LOAD(SL, U)                                                                        \
        { _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                         \
              wr[SL][t][0] = xg_ld_nt(wp + t * wt + (long long)(U) * 128);                         \
              wr[SL][t][1] = xg_ld_nt(wp + t * wt + (long long)(U) * 128 + 64); } }
#else
#define WMAR_XG_LOAD(SL, U)                                                                        \
        { _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                         \
              wr[SL][t][0] = xg_ld_nt(wp + t * wt + (long long)(U) * 128);                         \
              wr[SL][t][1] = xg_ld_nt(wp + t * wt + (long long)(U) * 128 + 64); }                  \
          _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                          \
              xr[SL][i][0] = xp[((long long)(U) * 128 + i * 64) * 2];                              \
              xr[SL][i][1] = xp[((long long)(U) * 128 + i * 64) * 2 + 1]; } }
#endif
        const bool do_stats = LN && g == 0 && !(XG_ABL & 8);             // one workgroup per K slice also sums the rows of X
        double ssum[2] = {0.0, 0.0}, ssq[2] = {0.0, 0.0};
#define WMAR_XG_STATS(SL)                                                                          \
        if (do_stats) {                                                                            \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                        \
                const float4 p = xr[SL][i][0], q = xr[SL][i][1];                                   \
                ssum[i] += (((double)p.x + (double)p.y) + ((double)p.z + (double)p.w)) + (((double)q.x + (double)q.y) + ((double)q.z + (double)q.w)); \
                ssq[i] += sq4_f64(p) + sq4_f64(q);          /* never a v_fmac_f64 chain: common.h */ \
            } }
#ifdef XG_XQ
#define WMAR_XG_SPLIT(SL, PB)                                                                      \
        { _Pragma("unroll") for (int t = 0; t < NT; ++t) xg_split8(wr[SL][t][0], wr[SL][t][1], wh[PB][t], wm[PB][t], wl[PB][t]); }
#else
#define WMAR_XG_SPLIT(SL, PB)                                                                      \
        { _Pragma("unroll") for (int t = 0; t < NT; ++t) xg_split8(wr[SL][t][0], wr[SL][t][1], wh[PB][t], wm[PB][t], wl[PB][t]); \
          _Pragma("unroll") for (int i = 0; i < 2; ++i) xg_split8(xr[SL][i][0], xr[SL][i][1], xh[PB][i], xm[PB][i], xl[PB][i]); \
          WMAR_XG_STATS(SL) }
#endif
// six piece products per fp32 product, the small ones first
#define WMAR_XG_ROUND(WP, XP, PB)                                                                  \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) WMAR_XG_MFMA(WP[PB][t], XP[PB][i], acc[t][i]);
#define WMAR_XG_MMA(PB)                                                                            \
        WMAR_XG_ROUND(wl, xh, PB) WMAR_XG_ROUND(wh, xl, PB) WMAR_XG_ROUND(wm, xm, PB) WMAR_XG_ROUND(wm, xh, PB) WMAR_XG_ROUND(wh, xm, PB) WMAR_XG_ROUND(wh, xh, PB)
#pragma unroll
        for (int j = 0; j < RW && j < PER; ++j) WMAR_XG_LOAD(j, j)
#ifdef XG_XQ
        WMAR_XG_LOADXQ(0, 0)
        if (1 < PER) WMAR_XG_LOADXQ(1, 1)
#endif
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 wh[DB ? 2 : 1][NT], wm[DB ? 2 : 1][NT], wl[DB ? 2 : 1][NT], xh[DB ? 2 : 1][2], xm[DB ? 2 : 1][2], xl[DB ? 2 : 1][2];
        XG_T(1)
        if constexpr (DB) {
            WMAR_XG_SPLIT(0, 0)
            if (!(XG_ABL & 4) && RW < PER) WMAR_XG_LOAD(0, RW)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int cu = j & 1, nx = cu ^ 1;
                XG_T(2 + j)
                // the NEXT step's operands are split (VALU) between this step's MFMAs; their registers are refilled at once
                if (j + 1 < PER) {
                    const int sl = (j + 1) % RW;
                    WMAR_XG_SPLIT(sl, nx)
                    if (!(XG_ABL & 4) && j + 1 + RW < PER) WMAR_XG_LOAD(sl, j + 1 + RW)
                }
#ifdef XG_XQ
#pragma unroll
                for (int i = 0; i < 2; ++i) { xh[cu][i] = __builtin_bit_cast(bf16x8, xq[cu][i][0]); xm[cu][i] = __builtin_bit_cast(bf16x8, xq[cu][i][1]); xl[cu][i] = __builtin_bit_cast(bf16x8, xq[cu][i][2]); }
#endif
                WMAR_XG_MMA(cu)
#ifdef XG_XQ
                if (j + 2 < PER) WMAR_XG_LOADXQ(cu, j + 2)
#endif
#pragma unroll
                for (int i = 0; i < 12 * NT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    if (i < 2 * NT + 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int sl = j % RW;
                XG_T(2 + j)
                WMAR_XG_SPLIT(sl, 0)
                __builtin_amdgcn_sched_barrier(0);
                if (!(XG_ABL & 4) && j + RW < PER) WMAR_XG_LOAD(sl, j + RW)
                WMAR_XG_MMA(0)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef WMAR_XG_LOAD
#undef WMAR_XG_STATS
#undef WMAR_XG_SPLIT
#undef WMAR_XG_ROUND
#undef WMAR_XG_MMA
        XG_T(2 + PER)
        // the K parts meet in LDS; wave w sums rows w, w + NW, ... in fixed order and parks them in the XCD's L2
        if constexpr (NW == 8) {
            // waves 4..7 park their accumulators, waves 0..3 add their own (wave w + wave w+4: a fixed order) and park the sums
            if (w >= 4) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            red[w - 4][(t * 2 + i) * 4 + q][lane] = make_float4(acc[t][i][4 * q], acc[t][i][4 * q + 1], acc[t][i][4 * q + 2], acc[t][i][4 * q + 3]);
            }
            __syncthreads();
            if (w < 4) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 z = red[w][(t * 2 + i) * 4 + q][lane];
                            red[w][(t * 2 + i) * 4 + q][lane] = make_float4(acc[t][i][4 * q] + z.x, acc[t][i][4 * q + 1] + z.y, acc[t][i][4 * q + 2] + z.z, acc[t][i][4 * q + 3] + z.w);
                        }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        red[w][(t * 2 + i) * 4 + q][lane] = make_float4(acc[t][i][4 * q], acc[t][i][4 * q + 1], acc[t][i][4 * q + 2], acc[t][i][4 * q + 3]);
        }
        if (do_stats) {
            sred[w][0][lane] = make_double2(ssum[0], ssq[0]);
            sred[w][1][lane] = make_double2(ssum[1], ssq[1]);
        }
        __syncthreads();
        float4* pout = a.part + ((long long)(xcd * S_ + s) * TX4 + g * NT * 4) * 128;
#pragma unroll
        for (int r = 0; r < ROWS / NW; ++r) {
            const int row = r * NW + w, t = row >> 3, i = (row >> 2) & 1, q = row & 3;
            float4 v = red[0][row][lane];
#pragma unroll
            for (int o = 1; o < 4; ++o) { const float4 z = red[o][row][lane]; v.x += z.x; v.y += z.y; v.z += z.z; v.w += z.w; }
            pout[((t * 4 + q) * 2 + i) * 64 + lane] = v;
        }
#ifdef XG_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        XG_T(3 + PER)
        if (a.trace && lane == 0 && w == 0) for (int i = 0; i < PER + 4; ++i) a.trace[(long long)blockIdx.x * 16 + i] = ts[i] - ts[0];
#endif
        if (do_stats && threadIdx.x < 64) {
            const int i = threadIdx.x >> 5, r = threadIdx.x & 31;
            double ts = 0.0, tq = 0.0;
#pragma unroll
            for (int o = 0; o < NW; ++o) { ts += sred[o][i][r].x + sred[o][i][r + 32].x; tq += sred[o][i][r].y + sred[o][i][r + 32].y; }
            a.statp[(long long)(xcd * S_ + s) * 64 + threadIdx.x] = make_double2(ts, tq);
        }
    }
    if (MODE == 1) return;

    if (MODE == 0) {
        // ------------------------------------------------------------------------------------------ XCD-local barrier
#ifdef XG_TRACE
        unsigned long long tr[6] = {0, 0, 0, 0, 0, 0};
        tr[0] = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's partial stores are in the L2
        __syncthreads();
        if (threadIdx.x == 0) {
#ifdef XG_TRACE
            tr[1] = __builtin_amdgcn_s_memtime();
#endif
            unsigned* cnt = a.sync + xcd * 64;
            unsigned* gen = cnt + 32;
            // workgroup-scope atomics on global memory: global_atomic without sc1, performed by THIS XCD's L2
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef XG_TRACE
            tr[2] = __builtin_amdgcn_s_memtime(); tr[4] = old;
#endif
            if (old == 31u) {
                __hip_atomic_exchange(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                int spins = 0;
#if XG_POLL == 1
                while (xg_l2_read(gen) == gen0) {
#else
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0) {
#endif
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 22)) { __hip_atomic_store(a.fail + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
#ifdef XG_TRACE
                tr[5] = spins;
#endif
            }
            if (xg_l2_read(a.sync + xcd * 64 + 16) != xid) __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef XG_TRACE
            tr[3] = __builtin_amdgcn_s_memtime();
            if (a.trace) for (int i = 0; i < 6; ++i) a.trace[(long long)blockIdx.x * 6 + i] = tr[i];
#endif
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------------------------------------------- phase 2
    const int nunits = a.TX * 4;                                  // (16-column group, row tile) units of this XCD
    const int half = lane >> 5, r = lane & 31;
    for (int unit = c * NW + w; unit < nunits; unit += 32 * NW) {
        const int i16 = unit >> 1, mt = unit & 1;
        const int o = 2 * i16 + half;                            // this lane's column octet within the XCD
        const float4* pp = a.part + (((long long)(xcd * S_) * TX4 + o) * 2 + mt) * 64 + r;
        const long long sstr = (long long)TX4 * 128;
        float4 plo[S_], phi[S_];
#pragma unroll
        for (int ss = 0; ss < S_; ++ss) { plo[ss] = xg_ld_nt(pp + ss * sstr); phi[ss] = xg_ld_nt(pp + ss * sstr + 32); }   // nt: L1 bypass
        const int ng = (xcd * a.TX * 2 + i16) * 16 + 8 * half;   // first of this lane's 8 output features
        const int m = 32 * mt + r;
        const float4 b0 = *(const float4*)(a.bias + ng), b1 = *(const float4*)(a.bias + ng + 4);
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1v = c0, x0 = c0, x1 = c0;
        double2 stv[LN ? S_ : 1];
        if (LN) {
            c0 = *(const float4*)(a.c1 + ng); c1v = *(const float4*)(a.c1 + ng + 4);
#pragma unroll
            for (int ss = 0; ss < S_; ++ss) {
                const double* sp = (const double*)(a.statp + (long long)(xcd * S_ + ss) * 64 + m);
                stv[ss] = make_double2(__builtin_nontemporal_load(sp), __builtin_nontemporal_load(sp + 1));
            }
        } else {
            const float4* xr_ = a.Xres + ((long long)((ng >> 4) * 2 + mt) * 64 + lane) * 2;
            x0 = xr_[0]; x1 = xr_[1];
        }
        float4 lo = plo[0], hi = phi[0];
#pragma unroll
        for (int ss = 1; ss < S_; ++ss) {
            lo.x += plo[ss].x; lo.y += plo[ss].y; lo.z += plo[ss].z; lo.w += plo[ss].w;
            hi.x += phi[ss].x; hi.y += phi[ss].y; hi.z += phi[ss].z; hi.w += phi[ss].w;
        }
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        if (LN) {
            double sm = 0.0, sq = 0.0;
#pragma unroll
            for (int ss = 0; ss < S_; ++ss) { sm += stv[ss].x; sq += stv[ss].y; }
            const double mean = sm * a.invK;
            const float mu = (float)mean;
            const float rstd = rsqrtf((float)var_f64(sq * a.invK, mean) + 1e-5f);
            const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1v.x, c1v.y, c1v.z, c1v.w};
            // LN(x) W^T = rstd * (x W'^T - mean * rowsum(W')) + (bias + W beta): phase 1 ran on the raw rows
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rstd * (v[e] - mu * cc[e]) + bb[e];
        } else {
            const float xx[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xx[e] + (bb[e] + v[e]);
        }
        if (EPI == XG_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xg_gelu(v[e]);
        }
        if (EPI == XG_EPI_QKV) {
            if (m < a.B) {
                const int which = ng / a.D, cq = ng - which * a.D;
                float* dst;
                if (which == 0) dst = a.qbuf + (long long)m * a.D + cq;
                else {
                    const int hh = cq / a.hd, d = cq - hh * a.hd;
                    dst = (which == 1 ? a.kcache : a.vcache) + (((long long)m * a.H + hh) * a.Tmax + *a.pos_dev) * a.hd + d;
                }
                *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        } else {
            float4* od = a.out + ((long long)((ng >> 4) * 2 + mt) * 64 + lane) * 2;
            od[0] = make_float4(v[0], v[1], v[2], v[3]);
            od[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

}  // namespace wmar
