// Device kernels of the Chameleon (Llama-style, bf16) decode engine for gfx950 -- SURVEY.md
// section 8a row C1.  Reference: deps/chameleon/inference/transformer.py:37-353 (Attention with
// qk LayerNorm + RoPE + KV cache, SwiGLU FeedForward, RMSNorm blocks, output head).
//
// Decode at 3*B rows (three guidance streams) is HBM-bound: 13.5 GB of bf16 weights and about as
// much KV cache per step against ~1 GFLOP/row, so the GEMM is built around the weight stream:
//   * weights `W[N][K]` -> `Wp[n/32][k/16][lane][8 bf16]`: one wave-wide 16-byte load (1 KiB
//     contiguous) is the A operand of one v_mfma_f32_32x32x16_bf16 per row tile;
//   * activations `X[M][K]` -> `Xp[k/16][m/32][lane][8 bf16]` (the B operand); a workgroup's four
//     waves own four column tiles over the SAME k range, so the activation chunk is staged once in
//     LDS (double buffered, one barrier per 256-column chunk) instead of four times through L1;
//   * the 32 weight rows of a tile are permuted at pack time so that the 16 accumulator registers
//     of a lane are two runs of 8 consecutive output features: every epilogue stores the next
//     layer's operand layout directly (16 B of bf16 or 32 B of fp32 per lane per run);
//   * RMSNorm's weight is folded into the following GEMM's weights; 1/rms is applied to the
//     accumulator (by the epilogue or by the consumer of the split-K slabs);
//   * SwiGLU: a w13 tile interleaves 16 rows of w1 with the matching 16 rows of w3, so
//     silu(x1)*x3 happens in registers and lands as one k-block of w2's input.
// Rounding points follow the reference's bf16 module boundaries (Linear outputs, LayerNorm, RoPE,
// SiLU, products, residual adds, attention output are rounded to bf16; accumulation is fp32).
#pragma once
#include "decoder_host.h"

namespace wmar {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// streamed-once data (weights): non-temporal 16-byte load
__device__ __forceinline__ uint4 ld_nt_u4(const uint4* p) {
    u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t bf_round_bits(float f) {   // fp32 -> bf16 bits, round to nearest even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf_round(float f) { return __uint_as_float(bf_round_bits(f) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ void bf_unpack8(const uint4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 bf_pack8(const float* f) {   // rounds
    uint4 v;
    v.x = bf_round_bits(f[0]) | (bf_round_bits(f[1]) << 16);
    v.y = bf_round_bits(f[2]) | (bf_round_bits(f[3]) << 16);
    v.z = bf_round_bits(f[4]) | (bf_round_bits(f[5]) << 16);
    v.w = bf_round_bits(f[6]) | (bf_round_bits(f[7]) << 16);
    return v;
}

// MFMA row i of a 32-row tile carries output feature cham_row_feature(i) of that tile.
__host__ __device__ __forceinline__ int cham_row_feature(int i) {
    const int g = i >> 3, h = (i >> 2) & 1, r = i & 3;
    return 16 * (g >> 1) + 8 * h + 4 * (g & 1) + r;
}

// ------------------------------------------------------------------------------- packing
// src: [N][K] bf16 (src_bf16) or fp32.  mode 0: tile nt holds rows nt*32 + feature.
// mode 1 (w13, rows [0,Hd) = w1, [Hd,2Hd) = w3): tile nt holds w1 rows 16nt..16nt+15 as
// features 0..15 and w3 rows 16nt..16nt+15 as features 16..31.
// gamma (nullable, [K], same dtype as src): W'[n][k] = bf16(W[n][k] * gamma[k]).
struct BPackArgs {
    const void* src; const void* gamma; uint4* dst;
    int N, K, mode, Hd, src_bf16;
};

static __global__ __launch_bounds__(64) void k_bpack(BPackArgs a) {
    const int KB = a.K / 16;
    const int nt = blockIdx.x / KB, kb = blockIdx.x % KB;
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int f = cham_row_feature(i);
    long long row;
    if (a.mode == 0) row = (long long)nt * 32 + f;
    else row = f < 16 ? (long long)nt * 16 + f : (long long)a.Hd + (long long)nt * 16 + (f - 16);
    const int k0 = kb * 16 + 8 * h;
    float v[8], gm[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gm[j] = 1.0f;
    if (row < a.N) {
        if (a.src_bf16) {
            const uint4 w = *(const uint4*)((const uint16_t*)a.src + row * a.K + k0);
            bf_unpack8(w, v);
            if (a.gamma) { const uint4 g4 = *(const uint4*)((const uint16_t*)a.gamma + k0); bf_unpack8(g4, gm); }
        } else {
            const float* s = (const float*)a.src + row * a.K + k0;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s[j];
            if (a.gamma)
#pragma unroll
                for (int j = 0; j < 8; ++j) gm[j] = ((const float*)a.gamma)[k0 + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    if (a.gamma)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * gm[j];
    a.dst[((long long)nt * KB + kb) * 64 + lane] = bf_pack8(v);
}

// fp32 copy of a (possibly bf16) vector
static __global__ void k_to_f32(const void* src, float* dst, long long n, int src_bf16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[i] = src_bf16 ? __uint_as_float((uint32_t)((const uint16_t*)src)[i] << 16) : ((const float*)src)[i];
}
// bf16 copy of a (possibly fp32) matrix (embedding table)
static __global__ void k_to_bf16(const void* src, uint16_t* dst, long long n, int src_bf16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[i] = src_bf16 ? ((const uint16_t*)src)[i] : (uint16_t)bf_round_bits(((const float*)src)[i]);
}

// ---------------------------------------------------------------------- row statistics
// 1/rms of row m from the per-chunk fp64 sums of squares: rsqrt(mean(x^2) + eps)  (xformers RMSNorm)
__device__ __forceinline__ float cham_rstd(const double* __restrict__ ssq, int n_chunks, int Mpad, int m, int K, float eps) {
    double s = 0;
    for (int c0 = 0; c0 < n_chunks; c0 += 16) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = ssq[(long long)min(c0 + i, n_chunks - 1) * Mpad + m];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (c0 + i < n_chunks) s += v[i];
    }
    return rsqrtf((float)(s * inv_count_f64((double)K)) + eps);
}

// ------------------------------------------------------------------------------- GEMM
// Work decomposition ("stream-K"): a unit is (group of 8 column tiles, chunk of BG_KC k-blocks).
// The grid is a fixed number of workgroups G (one or two per CU); workgroup w takes the
// contiguous units [w*U/G, (w+1)*U/G) in (group-major, chunk-minor) order, so every CU streams the
// same number of weight bytes whatever N and K are.  When a workgroup leaves a group it writes its
// partial sums as one "piece" of that group; consumers add the pieces of a group in piece order
// (fixed, so results do not depend on timing).  sk_* below is the shared index arithmetic.
#ifndef WMAR_BG_KC
#define WMAR_BG_KC 8
#endif
constexpr int BG_KC = WMAR_BG_KC;         // k-blocks (of 16) per unit
#ifndef WMAR_BG_TG
#define WMAR_BG_TG 4
#endif
// column tiles per group: the waves of a workgroup share one staged activation chunk.  4 (one tile per wave), not 8: a group's K range is
// then cut into half as many pieces -- the fp32 partial-sum pieces of the four GEMMs of a block were 66 MB written and read again per
// 422 MB of weights; measured 5.15 -> 4.67 ms per step on the 7B model (the activation chunk is read twice as often from L2 instead)
constexpr int BG_TG = WMAR_BG_TG;
constexpr int BG_MAXP = 16;      // pieces per group the slab buffers are sized for
enum { BEPI_SLAB = 0, BEPI_LOGITS = 2 };

struct SkInfo { int C, U, G; };  // chunks per group, total units, workgroups
__host__ __device__ __forceinline__ int sk_wg_of(const SkInfo& k, long long u) { return (int)(((u + 1) * k.G - 1) / k.U); }
__host__ __device__ __forceinline__ int sk_first(const SkInfo& k, int g) { return sk_wg_of(k, (long long)g * k.C); }
__host__ __device__ __forceinline__ int sk_count(const SkInfo& k, int g) {
    return sk_wg_of(k, (long long)g * k.C + k.C - 1) - sk_first(k, g) + 1;
}

struct BGemmArgs {
    const uint4* Wp;       // [NT][KB][64]
    const uint4* Xp;       // [KB][MT][64]
    int KB, NT, M;
    SkInfo sk;
    float* slabs;          // BEPI_SLAB: [piece][NT*2][MT][64][8] fp32 partial sums
    long long slab_stride; // floats per piece
    float* logits;         // BEPI_LOGITS: [M][V]  (every workgroup must own whole groups)
    int V;
    const double* ssq; int n_chunks; int K; float eps;   // 1/rms of the input rows (LOGITS)
};

// NWV waves per workgroup share the group's 8 column tiles: 4 waves x 2 tiles (one wave per SIMD) or 8 waves x 1 tile (two per SIMD:
// while one waits for its weight loads the other multiplies).
template <int MT, int EPI, int NWV = 4>
__global__ __launch_bounds__(NWV * 64) void k_bgemm(BGemmArgs a) {
    constexpr int TPW = BG_TG / NWV, NTH = NWV * 64;
    __shared__ __attribute__((aligned(16))) uint4 xs[2 * BG_KC * MT * 64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SkInfo sk = a.sk;
    const long long u0 = (long long)blockIdx.x * sk.U / sk.G, u1 = ((long long)blockIdx.x + 1) * sk.U / sk.G;
    const int nu = (int)(u1 - u0);
    constexpr int XPT = BG_KC * MT / NWV;     // uint4 per thread per chunk

    f32x16 acc[TPW][MT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;

    float rstd[MT];
    if (EPI == BEPI_LOGITS) {
#pragma unroll
        for (int i = 0; i < MT; ++i) rstd[i] = cham_rstd(a.ssq, a.n_chunks, MT * 32, i * 32 + (lane & 31), a.K, a.eps);
    }
    // rows past M are padding: their lanes are not fetched (they read as zeros)
    bool xact[XPT];
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
        const int e = (threadIdx.x + u * NTH) % (MT * 64);
        xact[u] = (e / 64) * 32 + (e & 31) < a.M;
    }

    uint4 wA[TPW][BG_KC], wB[TPW][BG_KC], xr[XPT];
    // unit ui of this workgroup: group g, k-blocks [c*BG_KC, ...) clamped to KB-1 (clamped blocks are skipped by the MFMA loop)
#define CH_LOADW(WB, UI)                                                                  \
    {                                                                                     \
        const long long uu = u0 + (UI);                                                   \
        const int g_ = (int)(uu / sk.C), c_ = (int)(uu % sk.C);                           \
        _Pragma("unroll") for (int t = 0; t < TPW; ++t) {                                 \
            const int nt_ = min(g_ * BG_TG + TPW * w + t, a.NT - 1);                        \
            const uint4* wp_ = a.Wp + (long long)nt_ * a.KB * 64 + lane;                  \
            _Pragma("unroll") for (int u = 0; u < BG_KC; ++u)                             \
                WB[t][u] = ld_nt_u4(wp_ + (long long)min(c_ * BG_KC + u, a.KB - 1) * 64); \
        }                                                                                 \
    }
#define CH_LOADX(UI)                                                                      \
    {                                                                                     \
        const int c_ = (int)((u0 + (UI)) % sk.C);                                         \
        _Pragma("unroll") for (int u = 0; u < XPT; ++u) {                                 \
            const int e = threadIdx.x + u * NTH;              /* element of [BG_KC][MT][64] */ \
            const int kk = min(c_ * BG_KC + e / (MT * 64), a.KB - 1);                     \
            xr[u] = a.Xp[(long long)kk * (MT * 64) + e % (MT * 64)];      /* unconditional: a load under a branch spoils hipcc's vmcnt counts */ \
            if (!xact[u]) xr[u] = make_uint4(0u, 0u, 0u, 0u);                             \
        }                                                                                 \
    }
#define CH_STOREX(BUF)                                                                    \
    _Pragma("unroll") for (int u = 0; u < XPT; ++u) xs[(BUF) * (BG_KC * MT * 64) + threadIdx.x + u * NTH] = xr[u];
#define CH_MMA1(WB, BUF, U)                                                               \
    {                                                                                     \
        uint4 xv[MT];                                                                     \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) xv[i] = xs[(BUF) * (BG_KC * MT * 64) + ((U) * MT + i) * 64 + lane]; \
        _Pragma("unroll") for (int t = 0; t < TPW; ++t) {                                 \
            const bf16x8 wv = __builtin_bit_cast(bf16x8, WB[t][U]);                       \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                \
                acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, __builtin_bit_cast(bf16x8, xv[i]), acc[t][i], 0, 0, 0); \
        }                                                                                 \
    }
    // MFMAs of unit UI, then (when the workgroup leaves the group) the epilogue of that group
#define CH_UNIT(WB, BUF, UI)                                                              \
    {                                                                                     \
        const long long uu = u0 + (UI);                                                   \
        const int g_ = (int)(uu / sk.C), c_ = (int)(uu % sk.C);                           \
        const int nk = a.KB - c_ * BG_KC;                                                 \
        if (nk >= BG_KC) {                                                                \
            _Pragma("unroll") for (int u = 0; u < BG_KC; ++u) CH_MMA1(WB, BUF, u)         \
        } else {                                                                          \
            _Pragma("unroll") for (int u = 0; u < BG_KC; ++u)                             \
                if (u < nk) CH_MMA1(WB, BUF, u)                                           \
        }                                                                                 \
        if ((UI) + 1 == nu || c_ + 1 == sk.C) flush(g_);                                  \
    }
    auto flush = [&](int g) {
        const int hh = lane >> 5;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int nt = g * BG_TG + TPW * w + t;
            if (nt < a.NT) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (EPI == BEPI_SLAB) {
                        const int piece = (int)blockIdx.x - sk_first(sk, g);
                        if (i * 32 + (lane & 31) >= a.M) continue;      // padding rows are never stored: a slot keeps zero (create-time memset) or the finite sums an earlier, larger batch left
                                                                         // there.  Safe because every consumer works row by row (a row of the residual stream, of the SwiGLU product, of q/k/v
                                                                         // depends on the same row only) and rows >= M are never read back as results or appended to the KV cache
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            float* dst = a.slabs + (long long)piece * a.slab_stride + ((((long long)nt * 2 + b) * MT + i) * 64 + lane) * 8;
                            *(float4*)dst = make_float4(acc[t][i][8 * b], acc[t][i][8 * b + 1], acc[t][i][8 * b + 2], acc[t][i][8 * b + 3]);
                            *(float4*)(dst + 4) = make_float4(acc[t][i][8 * b + 4], acc[t][i][8 * b + 5], acc[t][i][8 * b + 6], acc[t][i][8 * b + 7]);
                        }
                    } else {
                        const int m = i * 32 + (lane & 31);
                        if (m < a.M) {
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                float* dst = a.logits + (long long)m * a.V + nt * 32 + 16 * b + 8 * hh;
                                float o[8];
#pragma unroll
                                for (int j = 0; j < 8; ++j) o[j] = bf_round(rstd[i] * acc[t][i][8 * b + j]);   // logits.float() of a bf16 Linear
                                *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
                                *(float4*)(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
        }
    };
    if (nu == 0) return;
    CH_LOADX(0)
    CH_LOADW(wA, 0)
    __builtin_amdgcn_sched_barrier(0);
    CH_STOREX(0)
    __syncthreads();
    // The prefetch of the next unit is UNCONDITIONAL (past the end it re-reads the last unit: L2 hits): under a run-time branch hipcc's
    // s_waitcnt pass assumes the smaller outstanding count at the merge, every MFMA of unit u then also waits for unit u+1's loads
    // and the register double buffer degenerates to one unit in flight.
    for (int ui = 0; ui < nu; ui += 2) {
        { const int un = min(ui + 1, nu - 1); CH_LOADX(un) CH_LOADW(wB, un) }   // X first: waiting for it must not drain the weight loads
        __builtin_amdgcn_sched_barrier(0);
        CH_UNIT(wA, 0, ui)
        __builtin_amdgcn_sched_barrier(0);
        CH_STOREX(1)
        __syncthreads();
        if (ui + 1 >= nu) break;
        { const int un = min(ui + 2, nu - 1); CH_LOADX(un) CH_LOADW(wA, un) }
        __builtin_amdgcn_sched_barrier(0);
        CH_UNIT(wB, 1, ui + 1)
        __builtin_amdgcn_sched_barrier(0);
        CH_STOREX(0)
        __syncthreads();
    }
#undef CH_LOADW
#undef CH_LOADX
#undef CH_STOREX
#undef CH_MMA1
#undef CH_UNIT
}

// h = silu(x1) * x3 from the pieces of the w13 GEMM (FeedForward.forward, transformer.py:214-216):
// x13 = bf16(rstd * sum of pieces); silu and the product are bf16 ops.  Tile nt of that GEMM holds
// x1 features 16nt..16nt+15 (block 2nt) and the matching x3 features (block 2nt+1); the result is
// k-block nt of w2's input.  One wave per (tile, row tile).
struct SwigluArgs {
    const float* slabs; long long slab_stride; SkInfo sk;
    uint4* out;            // [NT][MT][64]
    const double* ssq; int n_chunks; int K; float eps;
    int NT, MT;
};
static __global__ __launch_bounds__(256) void k_cham_swiglu(SwigluArgs a) {
    // grid: (ceil(NT / 4), MT); the four waves take four consecutive tiles of one row tile
    __shared__ double red[8][32];
    const int mt = blockIdx.y, lane = threadIdx.x & 63;
    const int nt = min(blockIdx.x * 4 + (int)(threadIdx.x >> 6), a.NT - 1);
    const bool live = blockIdx.x * 4 + (int)(threadIdx.x >> 6) < a.NT;
    const int g = nt / BG_TG, np = sk_count(a.sk, g);
    const long long i1 = ((((long long)nt * 2) * a.MT + mt) * 64 + lane) * 8, i3 = ((((long long)nt * 2 + 1) * a.MT + mt) * 64 + lane) * 8;
    // the first four pieces are requested before the row statistics: both arrive in one round trip
    float4 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const long long po = (long long)min(d, np - 1) * a.slab_stride;
        a0[d] = *(const float4*)(a.slabs + po + i1); a1[d] = *(const float4*)(a.slabs + po + i1 + 4);
        b0[d] = *(const float4*)(a.slabs + po + i3); b1[d] = *(const float4*)(a.slabs + po + i3 + 4);
    }
    {   // 1/rms of the 32 rows: 8 thread groups x (n_chunks / 8) statistics chunks each
        const int r = threadIdx.x & 31, grp = threadIdx.x >> 5;
        // (round 5: the group's chunks are requested sixteen at a time with clamped indices -- a load inside a counted loop is one L2 round
        // trip per chunk, 8 of them at 4096 features; same summation order)
        double ssum = 0;
        for (int c0 = grp; c0 < a.n_chunks; c0 += 8 * 16) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = a.ssq[(long long)min(c0 + 8 * i, a.n_chunks - 1) * a.MT * 32 + mt * 32 + r];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (c0 + 8 * i < a.n_chunks) ssum += v[i];
        }
        red[grp][r] = ssum;
    }
    __syncthreads();
    if (!live) return;
    double tot = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i][lane & 31];
    const float rstd = rsqrtf((float)(tot * inv_count_f64((double)a.K)) + a.eps);
    float x1[8], x3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x1[j] = 0.f; x3[j] = 0.f; }
    for (int s0 = 0; s0 < np; s0 += 4) {        // four pieces per round trip
        if (s0 > 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const long long po = (long long)min(s0 + d, np - 1) * a.slab_stride;
                a0[d] = *(const float4*)(a.slabs + po + i1); a1[d] = *(const float4*)(a.slabs + po + i1 + 4);
                b0[d] = *(const float4*)(a.slabs + po + i3); b1[d] = *(const float4*)(a.slabs + po + i3 + 4);
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (s0 + d < np) {
                x1[0] += a0[d].x; x1[1] += a0[d].y; x1[2] += a0[d].z; x1[3] += a0[d].w;
                x1[4] += a1[d].x; x1[5] += a1[d].y; x1[6] += a1[d].z; x1[7] += a1[d].w;
                x3[0] += b0[d].x; x3[1] += b0[d].y; x3[2] += b0[d].z; x3[3] += b0[d].w;
                x3[4] += b1[d].x; x3[5] += b1[d].y; x3[6] += b1[d].z; x3[7] += b1[d].w;
            }
    }
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float u1 = bf_round(rstd * x1[j]), u3 = bf_round(rstd * x3[j]);
        o[j] = bf_round(u1 / (1.0f + expf(-u1))) * u3;
    }
    a.out[((long long)nt * a.MT + mt) * 64 + lane] = bf_pack8(o);
}

// ------------------------------------------------------------- embedding / residual update
// x = tok_embeddings[tok[m]] (bf16), per-chunk sums of squares.  One workgroup per (chunk of
// CHAM_STAT_KB k-blocks, row tile), one k-block per wave: the kernel is a few dependent L2 round
// trips over little data, so it wants many small workgroups.
constexpr int CHAM_STAT_KB = 4;
struct ChamResidArgs {
    uint4* x;                   // [KB][MT][64]
    const float* slabs; long long slab_stride; SkInfo sk;   // residual branch: sum of the group's pieces, rounded to bf16 first
    const uint16_t* emb;        // embed: [V][K] bf16
    const long long* tok;       // [M]
    double* ssq;                // [n_chunks][Mpad]
    int KB, MT, M, K;
};

template <bool EMBED>
__global__ __launch_bounds__(256) void k_cham_resid(ChamResidArgs a) {
    __shared__ double red[4][64];
    const int c = blockIdx.x / a.MT, mt = blockIdx.x % a.MT;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = mt * 32 + (lane & 31), h = lane >> 5;
    const int kb0 = c * CHAM_STAT_KB, kb1 = min(a.KB, kb0 + CHAM_STAT_KB);
    const long long tk = (EMBED && m < a.M) ? a.tok[m] : 0;
    double ss = 0.0;
    for (int kb = kb0 + w; kb < kb1; kb += 4) {
        const long long idx = ((long long)kb * a.MT + mt) * 64 + lane;
        float r[8];
        if (EMBED) {
            const uint4 e = *(const uint4*)(a.emb + tk * a.K + kb * 16 + 8 * h);
            bf_unpack8(e, r);
            if (m >= a.M)
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = 0.f;
        } else {
            float xv[8], br[8];
            bf_unpack8(a.x[idx], xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) br[j] = 0.f;
            const int np = sk_count(a.sk, (kb >> 1) / BG_TG);
            for (int s0 = 0; s0 < np; s0 += 8) {        // eight pieces per round trip
                float4 p0[8], p1[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const float* sp = a.slabs + (long long)min(s0 + d, np - 1) * a.slab_stride + idx * 8;
                    p0[d] = *(const float4*)sp; p1[d] = *(const float4*)(sp + 4);
                }
#pragma unroll
                for (int d = 0; d < 8; ++d)
                    if (s0 + d < np) {
                        br[0] += p0[d].x; br[1] += p0[d].y; br[2] += p0[d].z; br[3] += p0[d].w;
                        br[4] += p1[d].x; br[5] += p1[d].y; br[6] += p1[d].z; br[7] += p1[d].w;
                    }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = bf_round(xv[j] + bf_round(br[j]));   // h = x + Linear(...) in bf16
        }
        a.x[idx] = bf_pack8(r);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += prod_f64(r[j], r[j]);       // never a v_fmac_f64 chain: common.h
    }
    red[w][lane] = ss;            // all 64 lanes, no shuffle (decoder_kernels.h, k_qkvx_bx's keeper reduction)
    __syncthreads();
    if (threadIdx.x < 32) {
        const int r = threadIdx.x;
        a.ssq[(long long)c * a.MT * 32 + mt * 32 + r] = (red[0][r] + red[0][r + 32]) + (red[1][r] + red[1][r + 32]) + (red[2][r] + red[2][r + 32]) + (red[3][r] + red[3][r + 32]);
    }
}

// ------------------------------------------------------------------------ decode attention
struct ChamAttnArgs {
    const float* qkv_slabs; long long slab_stride; SkInfo sk;   // pieces of [(D + 2*Dkv)/16][MT][64][8]
    const double* ssq; int n_chunks; int K; float eps;      // attention_norm statistics of the input rows
    const float *qn_w, *qn_b, *kn_w, *kn_b;                 // [hd] LayerNorm(head_dim) of q and k; null: no qk normalisation
    uint16_t* kcache; uint16_t* vcache;                     // [M][Hkv][Tmax][hd] bf16 (this layer)
    uint4* y;                                               // packed [D/16][MT][64]
    const int* pos;                                         // [M] position of each row's token
    int D, H, Hkv, Tmax, MT;
    float scale;
    const float2* rope;                                     // [Tmax][hd/2] (cos, sin) of position * theta^(-2i/hd)
};

// (cos, sin)(p * theta^(-2i/hd)) in fp32 -- what rope_padded evaluates per element, tabulated once per engine
static __global__ void k_rope_table(float2* tab, int T, int half, float theta) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)T * half) return;
    const int p = (int)(idx / half), i = (int)(idx % half);
    const float freq = powf(theta, -2.0f * (float)i / (float)(2 * half));
    float sn, cs;
    sincosf((float)p * freq, &sn, &cs);
    tab[idx] = make_float2(cs, sn);
}

#ifndef WMAR_CHAM_ATT_CH
#define WMAR_CHAM_ATT_CH 4      // 1-KiB loads of K (and of V) per chunk of k_cham_attn (bf16 rows: 16 cached rows of head_dim 128 at 4)
#endif
template <int HD, int NWA>
__global__ __launch_bounds__(NWA * 64) void k_cham_attn(ChamAttnArgs a) {
    constexpr int LPR = HD / 8;            // lanes per cached row (8 bf16 each)
    constexpr int RPI = 64 / LPR;          // rows per wave-wide load
    constexpr int CH = WMAR_CHAM_ATT_CH;
    constexpr int ROWS = CH * RPI;
    __shared__ __attribute__((aligned(16))) float part[NWA][HD + 4];
    __shared__ __attribute__((aligned(16))) float qkv_s[3][HD];
    const int m = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int hk = h / (a.H / a.Hkv);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = a.pos[m];
    const int T = P + 1;
    const int sub = lane % LPR, rsel = lane / LPR;
    uint16_t* Kc = a.kcache + (((long long)m * a.Hkv + hk) * a.Tmax) * HD + sub * 8;
    uint16_t* Vc = a.vcache + (((long long)m * a.Hkv + hk) * a.Tmax) * HD + sub * 8;
    const int nchunk = (T + ROWS - 1) / ROWS;

// K/V rows stream once per step: non-temporal loads (see k_attn_decode; -DWMAR_ATT_PLAIN_KV restores plain loads)
#ifndef WMAR_ATT_PLAIN_KV
#define CA_KV_LD(P_) ld_nt_u4((const uint4*)(P_))
#else
#define CA_KV_LD(P_) (*(const uint4*)(P_))
#endif
#define CA_LOAD(KB_, VB_, C0)                                                             \
    _Pragma("unroll") for (int u = 0; u < CH; ++u) {                                      \
        const int t = min((C0) * ROWS + u * RPI + rsel, T - 1);                           \
        KB_[u] = CA_KV_LD(Kc + (long long)t * HD);                                        \
        VB_[u] = CA_KV_LD(Vc + (long long)t * HD);                                        \
    }
    uint4 kA[CH], vA[CH], kB[CH], vB[CH];
    if (w < nchunk) { CA_LOAD(kA, vA, w) }
    __builtin_amdgcn_sched_barrier(0);
    if (w == 0) {
        // 1/rms of this row (one statistics chunk per lane), the q/k/v columns of this head from the slabs
        // straight-line (host: n_chunks <= 128): a load inside a loop makes hipcc wait for EVERY outstanding load at its head, the first
        // K / V chunk requested above included (k_attn_decode's prologue, decoder_kernels.h)
        const double s0_ = a.ssq[(long long)min(lane, a.n_chunks - 1) * a.MT * 32 + m];
        const double s1_ = a.ssq[(long long)min(lane + 64, a.n_chunks - 1) * a.MT * 32 + m];
        double ssum = (lane < a.n_chunks ? s0_ : 0.0) + (lane + 64 < a.n_chunks ? s1_ : 0.0);
        float v3[3][8];
        float4 rp0 = make_float4(1.f, 0.f, 1.f, 0.f), rp1 = rp0;
        if (rsel == 0) {
            const float4* rp = (const float4*)(a.rope + (long long)P * (HD / 2) + sub * 4);
            rp0 = rp[0]; rp1 = rp[1];
            const int mt = m >> 5;
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                const int n = (which == 0 ? h * HD : (which == 1 ? a.D + hk * HD : a.D + a.Hkv * HD + hk * HD)) + sub * 8;
                const long long idx = ((((long long)(n >> 4)) * a.MT + mt) * 64 + (m & 31) + 32 * ((n >> 3) & 1)) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) v3[which][j] = 0.f;
                const int np = sk_count(a.sk, (n >> 5) / BG_TG);
                for (int s0 = 0; s0 < np; s0 += 4) {      // four pieces per round trip
                    float4 p0[4], p1[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float* sp = a.qkv_slabs + (long long)min(s0 + d, np - 1) * a.slab_stride + idx;
                        p0[d] = *(const float4*)sp; p1[d] = *(const float4*)(sp + 4);
                    }
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (s0 + d < np) {
                            v3[which][0] += p0[d].x; v3[which][1] += p0[d].y; v3[which][2] += p0[d].z; v3[which][3] += p0[d].w;
                            v3[which][4] += p1[d].x; v3[which][5] += p1[d].y; v3[which][6] += p1[d].z; v3[which][7] += p1[d].w;
                        }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o);
        const float rstd = rsqrtf((float)(ssum * inv_count_f64((double)a.K)) + a.eps);
        if (rsel == 0) {
#pragma unroll
            for (int which = 0; which < 3; ++which)
#pragma unroll
                for (int j = 0; j < 8; ++j) v3[which][j] = bf_round(rstd * v3[which][j]);       // xqkv = wqkv(x) in bf16
            if (a.qn_w) {
                // q_normalization / k_normalization: LayerNorm(head_dim), eps 1e-5, computed in fp32, stored bf16
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    float sm = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) sm += v3[which][j];
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) sm += __shfl_xor(sm, o);
                    const float mean = sm * (1.0f / HD);
                    float sq = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float d = v3[which][j] - mean; sq += d * d; }
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) sq += __shfl_xor(sq, o);
                    const float rs = rsqrtf(sq * (1.0f / HD) + 1e-5f);
                    const float* gw = (which == 0 ? a.qn_w : a.kn_w) + sub * 8;
                    const float* gb = (which == 0 ? a.qn_b : a.kn_b) + sub * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v3[which][j] = bf_round((v3[which][j] - mean) * rs * gw[j] + gb[j]);
                }
            }
            // rotary embedding on adjacent pairs (xformers rope_padded, adjacents=True): angle = pos * theta^(-2i/hd)
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {
                    const float cs = p2 == 0 ? rp0.x : (p2 == 1 ? rp0.z : (p2 == 2 ? rp1.x : rp1.z));
                    const float sn = p2 == 0 ? rp0.y : (p2 == 1 ? rp0.w : (p2 == 2 ? rp1.y : rp1.w));
                    const float x0 = v3[which][2 * p2], x1 = v3[which][2 * p2 + 1];
                    v3[which][2 * p2] = bf_round(x0 * cs - x1 * sn);
                    v3[which][2 * p2 + 1] = bf_round(x0 * sn + x1 * cs);
                }
#pragma unroll
            for (int which = 0; which < 3; ++which)
#pragma unroll
                for (int j = 0; j < 8; ++j) qkv_s[which][sub * 8 + j] = v3[which][j];
            if (h % (a.H / a.Hkv) == 0) {
                *(uint4*)(Kc + (long long)P * HD) = bf_pack8(v3[1]);
                *(uint4*)(Vc + (long long)P * HD) = bf_pack8(v3[2]);
            }
        }
    }
    __syncthreads();
    float q[8], kn[8], vn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { q[j] = qkv_s[0][sub * 8 + j]; kn[j] = qkv_s[1][sub * 8 + j]; vn[j] = qkv_s[2][sub * 8 + j]; }
    __builtin_amdgcn_sched_barrier(0);

    float mx = -INFINITY, l = 0.f;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#define CA_CHUNK(KB_, VB_, C0)                                                            \
    {                                                                                     \
        float sc[CH];                                                                     \
        float cm = -INFINITY;                                                             \
        _Pragma("unroll") for (int u = 0; u < CH; ++u) {                                  \
            const int t = (C0) * ROWS + u * RPI + rsel;                                   \
            float kf[8];                                                                  \
            bf_unpack8(KB_[u], kf);                                                       \
            if (t >= T - 1) { _Pragma("unroll") for (int j = 0; j < 8; ++j) kf[j] = kn[j]; } \
            float p = 0.f;                                                                \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) p += kf[j] * q[j];              \
            _Pragma("unroll") for (int o = 1; o < LPR; o <<= 1) p += __shfl_xor(p, o);    \
            p = (t < T) ? p * a.scale : -INFINITY;                                        \
            sc[u] = p;                                                                    \
            cm = fmaxf(cm, p);                                                            \
        }                                                                                 \
        _Pragma("unroll") for (int o = LPR; o < 64; o <<= 1) cm = fmaxf(cm, __shfl_xor(cm, o)); \
        const float mn = fmaxf(mx, cm);                                                   \
        const float rs = __expf(mx - mn);                                                 \
        l *= rs;                                                                          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] *= rs;                       \
        _Pragma("unroll") for (int u = 0; u < CH; ++u) {                                  \
            const int t = (C0) * ROWS + u * RPI + rsel;                                   \
            float vf[8];                                                                  \
            bf_unpack8(VB_[u], vf);                                                       \
            if (t >= T - 1) { _Pragma("unroll") for (int j = 0; j < 8; ++j) vf[j] = vn[j]; } \
            const float e = __expf(sc[u] - mn);                                           \
            l += e;                                                                       \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] += e * vf[j];            \
        }                                                                                 \
        mx = mn;                                                                          \
    }
    // refills are unconditional (rows past the cache clamp to row T-1): see k_attn_decode
    for (int c = w; c < nchunk; c += 2 * NWA) {
        CA_LOAD(kB, vB, c + NWA)
        __builtin_amdgcn_sched_barrier(0);
        CA_CHUNK(kA, vA, c)
        __builtin_amdgcn_sched_barrier(0);
        CA_LOAD(kA, vA, c + 2 * NWA)
        __builtin_amdgcn_sched_barrier(0);
        if (c + NWA < nchunk) { CA_CHUNK(kB, vB, c + NWA) }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CA_LOAD
#undef CA_CHUNK
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        l += __shfl_xor(l, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o);
    }
    if (rsel == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[w][sub * 8 + j] = acc[j];
        if (sub == 0) { part[w][HD] = mx; part[w][HD + 1] = l; }
    }
    __syncthreads();
    if (w == 0 && rsel == 0) {
        float M = part[0][HD];
#pragma unroll
        for (int i = 1; i < NWA; ++i) M = fmaxf(M, part[i][HD]);
        float L = 0.f, o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NWA; ++i) {
            const float f = __expf(part[i][HD] - M);
            L += part[i][HD + 1] * f;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += part[i][sub * 8 + j] * f;
        }
        const float inv = 1.0f / L;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= inv;
        const int k = h * HD + sub * 8;
        a.y[((long long)(k >> 4) * a.MT + (m >> 5)) * 64 + (m & 31) + 32 * ((k >> 3) & 1)] = bf_pack8(o);
    }
}

static __global__ void k_cham_set3(int* p, int a, int b, int c) { p[0] = a; p[1] = b; p[2] = c; }

// per-row step bookkeeping of the generation loop
//   mode 0: (tok, pos)[m] = table row *step of the prefill tables
//   mode 1: tok[m] = last sampled token of image m % B (+ bpe offset already applied by the sampler), pos[m] += 1
static __global__ void k_cham_step(long long* tok, int* pos, const long long* tab_tok, const int* tab_pos, int row, int M,
                                   const long long* sampled, long long sampled_stride, const int* step_dev, int B, int mode) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    if (mode == 0) {
        tok[m] = tab_tok[(long long)row * M + m];
        pos[m] = tab_pos[(long long)row * M + m];
    } else {
        const int n = *step_dev;     // tokens sampled so far (>= 1)
        tok[m] = sampled[(long long)(m % B) * sampled_stride + (n - 1)];
        pos[m] += 1;
    }
}

}  // namespace wmar
