// Device kernels of the transformer decode engines (Taming minGPT in gpt.hip, RAR in rar.hip).
// Included by both translation units: everything here is a template or `static`.
#pragma once
#include <cstdint>

#include "sampler.h"
#include "bx_split.h"

namespace wmar {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// streamed-once data (weights): non-temporal 16-byte load
__device__ __forceinline__ float4 ld_nt(const float4* p) {
    f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// Output the NEXT kernel reads (QKV split-K pieces: 8.3 MB per launch, FC2 slabs, the hidden activation): a write-through store (sc1),
// so the bytes leave the XCD's L2 while the kernel runs instead of in the write-back at its end, which the next kernel waits for
// (guide: boundary + dirty bytes / 6 TB/s).  Round 4, same-box A/B over the 256-step loop: 4.140 -> 4.110 ms per step.  Same values,
// only the cache policy differs; -DWMAR_PLAIN_STORES restores plain stores.  (Data a later phase of the SAME launch reads through the
// L2 -- the slabs of k_bx_xr -- must stay plain: an sc1 store drops the line.)
__device__ __forceinline__ void st_out(float4* p, const float4 v) {
#ifndef WMAR_PLAIN_STORES
    const f32x4 q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(q) : "memory");
#else
    *p = v;
#endif
}
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ float2 ld_nt2(const float2* p) {
    f32x2 v = __builtin_nontemporal_load((const f32x2*)p);
    return make_float2(v.x, v.y);
}

// ------------------------------------------------------------------------ in-loop timeline stamps (dev builds: -DWMAR_STAMPS)
// Wave 0 of every workgroup of the five per-layer launches records eight 64-bit stamps into its launch's slot of a trace buffer
// (StepPlan sets the pointers when the engine was created with WMAR_STAMPS=1; scripts/stamp_table.py turns them into the per-launch
// table of DESIGN section 6): [0] s_memrealtime at entry (100 MHz, chip-wide: gaps BETWEEN launches), [1] s_memtime at entry,
// [2] first operands landed, [3] main loop done, [4] exit (stores issued and acknowledged), [5] s_memrealtime at exit, [6] a kernel-
// specific mark (k_bx_xr: barrier passed; attention: q/k/v finished), [7] the XCC id.  The "landed" stamp waits for the first loads
// with s_waitcnt vmcnt(0): a diagnostic build, a few percent slower than the product.
#ifdef WMAR_STAMPS
#define WMAR_ST_BEGIN unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; st_[0] = __builtin_amdgcn_s_memrealtime(); st_[1] = __builtin_amdgcn_s_memtime();
#define WMAR_ST(I) { asm volatile("" ::: "memory"); st_[I] = __builtin_amdgcn_s_memtime(); }
#define WMAR_ST_LANDED(I) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_[I] = __builtin_amdgcn_s_memtime(); }
#define WMAR_ST_END(PTR, UNIT)                                                                          \
    if ((PTR) && threadIdx.x == 0) {                                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        st_[4] = __builtin_amdgcn_s_memtime(); st_[5] = __builtin_amdgcn_s_memrealtime();               \
        unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); st_[7] = xcc_ & 15u; \
        unsigned long long* o_ = (PTR) + (long long)(UNIT) * 8;                                         \
        for (int i_ = 0; i_ < 8; ++i_) o_[i_] = st_[i_];                                                \
    }
#else
#define WMAR_ST_BEGIN
#define WMAR_ST(I)
#define WMAR_ST_LANDED(I)
#define WMAR_ST_END(PTR, UNIT)
#endif
constexpr int WMAR_STAMP_UNITS = 1536;     // workgroup slots per launch in the stamp buffer (the attention's grid at 64 rows x 24 heads)

constexpr int MAX_SLABS = 8;
constexpr int QKV_SLABS_MAX = 8;   // the QKV projection arrives in at most 8 split-K pieces
constexpr int STAT_CHUNKS_MAX = 64;   // n_embd <= 8192
constexpr int GEMM_STAGE = 4;   // k-blocks (of 8) per register stage of the skinny GEMM
#ifndef WMAR_GEMM_INTERLEAVE
#define WMAR_GEMM_INTERLEAVE 2   // MFMAs between two operand loads of the main loop (0: loads in one batch per stage)
#endif

// ------------------------------------------------------------------------ weight packing
// gamma (nullable): the LayerNorm scale of the layer that feeds this Linear, folded into
// the weights ( LN(x) W^T = xhat (W*gamma)^T + W beta ), so the GEMM's inner loop only
// has to form xhat = (x - mean) * rstd.
static __global__ void k_pack_linear(const float* __restrict__ W, float4* __restrict__ Wp, int N, int K, int nt_off,
                              int KB, const float* __restrict__ gamma) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (N/32)*KB*64
    long long total = (long long)(N / 32) * KB * 64;
    if (idx >= total) return;
    int lane = (int)(idx & 63);
    long long r = idx >> 6;
    int kb = (int)(r % KB);
    int nt = (int)(r / KB);
    int n = nt * 32 + (lane & 31);
    int k = kb * 8 + 4 * (lane >> 5);
    const float* src = W + (long long)n * K + k;
    float4 v = make_float4(src[0], src[1], src[2], src[3]);
    if (gamma) { v.x *= gamma[k]; v.y *= gamma[k + 1]; v.z *= gamma[k + 2]; v.w *= gamma[k + 3]; }
    Wp[((long long)(nt + nt_off) * KB + kb) * 64 + lane] = v;
}

// out[n] = (bias ? bias[n] : 0) + sum_k W[n][k] * v[k]   (one wave per output row; v = LN beta or gamma)
static __global__ __launch_bounds__(64) void k_fold_bias(const float* __restrict__ W, const float* __restrict__ bias,
                                                  const float* __restrict__ beta, float* __restrict__ out, int K) {
    const int n = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    for (int k = lane; k < K; k += 64) acc += prod_f64(W[(long long)n * K + k], beta[k]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[n] = (float)((bias ? (double)bias[n] : 0.0) + acc);
}

static __global__ void k_set_int(int* p, int v) { *p = v; }
static __global__ void k_advance3(int* p) { p[0] += 1; p[1] += 1; p[2] += 1; }  // pos, step, len

// --------------------------------------------------- residual update + LayerNorm statistics
// x_new = EMBED ? tok_emb[tok[m]] + pos_emb[pos]
//               : x + bias + sum_s slab[s]          (fixed summation order)
// and per (chunk, row) partial sums (sum, sum of squares) in fp64 for the consumer's fused LN.
struct ResidArgs {
    float4* x;                 // packed [KB][MT][64]
    const float4* slabs;       // [S][KB*MT*64]
    long long slab_stride;     // float4 units
    int S;
    const float* bias;         // [K]
    const float* gate;         // nullable: row-major [M][gate_stride] per-row, per-channel gate (RAR adaLN)
    long long gate_stride;
    const float* gate_u;       // nullable: rows >= gate_split share row *pos_dev of this [T][gate_stride] table
    int gate_split;
    int n_hi;                  // > 0: slab S-1 exists only for column tiles < n_hi (non-uniform split of the producing GEMM)
    double* stats;             // [n_chunks][Mpad][2]
    int KB, MT, n_chunks;
    // embed
    const float* tok_emb;      // [V][K]
    const float* pos_emb;      // [block][K]
    const long long* tok;      // token of row m: tok[m*tok_stride + (tok_use_pos ? *pos : 0)]
    long long tok_stride;
    int tok_use_pos;
    const int* pos_dev;
    int B, K;
};

// S = number of partial slabs (compile-time so that every load of a wave is issued up front:
// the kernel is a handful of dependent L2 round trips, not bandwidth).
template <bool EMBED, int S>
__global__ __launch_bounds__(256) void k_resid_stats(ResidArgs a) {
    constexpr int KPW = 4;  // k-blocks per wave (host guarantees chunk length <= 4*KPW)
    __shared__ double red[4][64][2];
    const int c = blockIdx.x / a.MT, mt = blockIdx.x % a.MT;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kb0 = (int)((long long)c * a.KB / a.n_chunks), kb1 = (int)((long long)(c + 1) * a.KB / a.n_chunks);
    const int m = mt * 32 + (lane & 31), half = lane >> 5;
    const float* erow = nullptr;
    const float* prow = nullptr;
    if (EMBED) {
        int pos = *a.pos_dev;
        long long tk = (m < a.B) ? a.tok[(long long)m * a.tok_stride + (a.tok_use_pos ? pos : 0)] : 0;
        erow = a.tok_emb + tk * a.K;
        prow = a.pos_emb + (long long)pos * a.K;
    }
    const float* grow = nullptr;
    if (!EMBED && a.gate)
        grow = (a.gate_u && m >= a.gate_split) ? a.gate_u + (long long)(*a.pos_dev) * a.gate_stride
                                               : a.gate + (long long)m * a.gate_stride;
    float4 v[KPW], bb[KPW], sl[KPW][S > 0 ? S : 1];
    int kbs[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kb = kb0 + w + 4 * i;
        kbs[i] = kb < kb1 ? kb : -1;
        const int kk = kb < kb1 ? kb : kb0;  // in-bounds dummy for idle slots
        const long long idx = ((long long)kk * a.MT + mt) * 64 + lane;
        const int k = kk * 8 + 4 * half;
        if (EMBED) {
            v[i] = *(const float4*)(erow + k);
            bb[i] = *(const float4*)(prow + k);
        } else {
            v[i] = a.x[idx];
            bb[i] = *(const float4*)(a.bias + k);
#pragma unroll
            for (int sidx = 0; sidx < S; ++sidx) sl[i][sidx] = a.slabs[(long long)sidx * a.slab_stride + idx];
        }
    }
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        if (kbs[i] < 0) continue;
        float4 r;
        if (EMBED) {
            r = make_float4(v[i].x + bb[i].x, v[i].y + bb[i].y, v[i].z + bb[i].z, v[i].w + bb[i].w);
        } else {
            float4 acc = sl[i][0];
            const bool short_tile = a.n_hi > 0 && (kbs[i] >> 2) >= a.n_hi;     // this column tile has one slab fewer
#pragma unroll
            for (int sidx = 1; sidx < S; ++sidx) {
                if (sidx == S - 1 && short_tile) continue;
                acc.x += sl[i][sidx].x; acc.y += sl[i][sidx].y; acc.z += sl[i][sidx].z; acc.w += sl[i][sidx].w;
            }
            float4 gg = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.gate) gg = *(const float4*)(grow + kbs[i] * 8 + 4 * half);
            r = make_float4(v[i].x + gg.x * (bb[i].x + acc.x), v[i].y + gg.y * (bb[i].y + acc.y),
                            v[i].z + gg.z * (bb[i].z + acc.z), v[i].w + gg.w * (bb[i].w + acc.w));
        }
        a.x[((long long)kbs[i] * a.MT + mt) * 64 + lane] = r;
        s += (double)r.x + (double)r.y + (double)r.z + (double)r.w;
        ss += sq4_f64(r);         // never a v_fmac_f64 chain: common.h
    }
    red[w][lane][0] = s; red[w][lane][1] = ss;       // all 64 lanes, no shuffle: see k_qkvx_bx's keeper reduction
    __syncthreads();
    if (threadIdx.x < 32) {
        double ts = 0, tss = 0;
        for (int i = 0; i < 4; ++i) { ts += red[i][threadIdx.x][0] + red[i][threadIdx.x + 32][0]; tss += red[i][threadIdx.x][1] + red[i][threadIdx.x + 32][1]; }
        const int Mpad = a.MT * 32;
        double* o = a.stats + ((long long)c * Mpad + mt * 32 + threadIdx.x) * 2;
        o[0] = ts; o[1] = tss;
    }
}

template <int S>
static void launch_resid_s(const ResidArgs& r, int grid, hipStream_t st) {
    hipLaunchKernelGGL((k_resid_stats<false, S>), dim3(grid), dim3(256), 0, st, r);
}
static int launch_resid(const ResidArgs& r, int grid, hipStream_t st) {
    switch (r.S) {
        case 1: launch_resid_s<1>(r, grid, st); break;
        case 2: launch_resid_s<2>(r, grid, st); break;
        case 3: launch_resid_s<3>(r, grid, st); break;
        case 4: launch_resid_s<4>(r, grid, st); break;
        case 5: launch_resid_s<5>(r, grid, st); break;
        case 6: launch_resid_s<6>(r, grid, st); break;
        case 7: launch_resid_s<7>(r, grid, st); break;
        case 8: launch_resid_s<8>(r, grid, st); break;
        default: set_error("resid: bad slab count %d", r.S); return WMAR_EINVAL;
    }
    return launch_status("k_resid_stats");
}

// mean / rstd of row m from the per-chunk fp64 partial sums written by k_resid_stats.
// All loads of a group of 16 chunks are issued together (one L2 round trip, not one per chunk).
__device__ __forceinline__ void ln_row_stats(const double* __restrict__ stats, int n_chunks, int Mpad, int m, int K,
                                             float* mu, float* rstd) {
    double sm = 0, sq = 0;
    for (int c0 = 0; c0 < n_chunks; c0 += 16) {
        double2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = min(c0 + i, n_chunks - 1);
            v[i] = *(const double2*)(stats + ((long long)c * Mpad + m) * 2);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (c0 + i < n_chunks) { sm += v[i].x; sq += v[i].y; }
    }
    const double invK = inv_count_f64((double)K);
    const double mean = sm * invK;
    *mu = (float)mean;
    *rstd = rsqrtf((float)var_f64(sq * invK, mean) + 1e-5f);
}

// ------------------------------------------------------------------------- skinny GEMM
enum { EPI_PACKED = 0, EPI_GELU = 1, EPI_QKV = 2, EPI_LOGITS = 3 };

struct GemmArgs {
    const float4* Wp;          // [NT][KB][64]
    const float4* Xp;          // [KB][MT][64]
    const float* bias;         // [N] or null
    const float* c1;           // [N] row sums of the gamma-folded weights (LN epilogue), or null
    int KB, NT, MT, S;
    int n_hi;                  // the first n_hi column tiles are split S+1 ways, the rest S ways (fills the 256 CUs evenly; 0: uniform)
    // fused LayerNorm on the B operand
    const double* stats; int n_chunks; int K;
    // epilogues
    float4* out_packed; long long slab_stride;          // EPI_PACKED (slab s) / EPI_GELU
    u32x4* out_planes;                                  // EPI_GELU, nullable: the activation as bf16 pieces instead (a k_bx GEMM follows)
    float* qbuf; float* kcache; float* vcache;          // EPI_QKV
    const int* pos_dev; int D, H, hd, Tmax;
    float* logits; int V;                               // EPI_LOGITS
    int B;
    unsigned long long* trace;                          // dev only (WMAR_GEMM_TRACE): 4 timestamps per workgroup
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// grid = NT * (MT/MTW) * S workgroups of NW waves.  Each workgroup owns one 32-column
// tile of the output for MTW row tiles and one K slice; its NW waves split that K slice
// and reduce through LDS in a fixed order.
// NTW > 1 (round 4): NTW adjacent column tiles per workgroup -- every activation fragment then feeds NTW weight fragments, i.e. the
// workgroup pulls 1 / NTW as many activation bytes per weight byte through its CU (whole-K launches with thousands of column tiles --
// the vocabulary head, RAR's adaLN GEMM -- were bound by exactly that: t ~ bytes per CU / 33 GB/s).  Host: S == 1,
// n_hi == 0, NT % NTW == 0.
template <int MTW, int NW, int EPI, bool LN, int ABL = 0, int U = GEMM_STAGE, bool ROT = true, int NTW = 1>
__global__ __launch_bounds__(NW * 64) void k_gemm(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NW][NTW*MTW*16][64]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int MG = a.MT / MTW;
    int bid = blockIdx.x;
    int nt, mg, s, S_t = a.S;
    if (NTW > 1) {                             // groups of NTW tiles; no K split across workgroups
        const int NTG = a.NT / NTW;
        nt = (bid % NTG) * NTW; mg = bid / NTG; s = 0;
    } else if (bid < a.NT * MG * a.S) {
        nt = bid % a.NT; bid /= a.NT;
        mg = bid % MG;
        s = bid / MG;
        if (nt < a.n_hi) S_t = a.S + 1;
    } else {                                   // the extra K slice of the first n_hi tiles (host: only with MG == 1)
        nt = bid - a.NT * MG * a.S; mg = 0; s = a.S; S_t = a.S + 1;
    }
    const int mt0 = mg * MTW;
    const int slices = S_t * NW;
    const int sl = s * NW + w;
    // 32-bit index math (there is no integer divide instruction: 64-bit division is ~10x dearer)
    int kb0 = (int)((unsigned)sl * (unsigned)a.KB / (unsigned)slices);
    int kb1 = (int)((unsigned)(sl + 1) * (unsigned)a.KB / (unsigned)slices);
    if (a.n_hi > 0 && a.KB % (NW * 2 * U) == 0) {
        // non-uniform split: workgroup boundaries on whole pipeline rounds (NW waves x 2U k-blocks), so that no wave is left
        // with an unpipelined tail -- e.g. 24 rounds over 5 workgroups = 4,5,5,5,5
        const unsigned units = (unsigned)a.KB / (NW * 2 * U);
        const int u0 = (int)((unsigned)s * units / (unsigned)S_t), u1 = (int)((unsigned)(s + 1) * units / (unsigned)S_t);
        kb0 = (u0 * NW + w * (u1 - u0)) * 2 * U;
        kb1 = kb0 + (u1 - u0) * 2 * U;
    }
    const int half = lane >> 5;
    WMAR_ST_BEGIN
#ifdef WMAR_GEMM_TRACE
    unsigned long long tr0 = __builtin_amdgcn_s_memtime(), tr1 = 0, tr2 = 0;
#endif

    f32x16 acc[NTW][MTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;

    float mu[MTW], rstd[MTW];
    const float4* Wp = a.Wp + (long long)nt * a.KB * 64 + lane;
    const long long wtile = (long long)a.KB * 64;           // next column tile
    const float4* Xp = a.Xp + (long long)mt0 * 64 + lane;
    const long long xstep = (long long)a.MT * 64;

    // Register double buffer: while the MFMAs of one stage (U k-blocks = 4U instructions per
    // row tile) run, the loads of the next stage are in flight.
    float4 wA[U][NTW], wB[U][NTW], xA[U][MTW], xB[U][MTW];
#define WMAR_LOAD(WBUF, XBUF, KB0)                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                            \
        const int kk = (KB0) + u;                                                               \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                        \
            WBUF[u][t] = (ABL == 2) ? make_float4(1.f, 2.f, 3.f, (float)kk) : ld_nt(Wp + t * wtile + (long long)kk * 64); \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            XBUF[u][i] = (ABL == 1) ? make_float4(1.f, 2.f, 3.f, (float)kk) : Xp[(long long)kk * xstep + i * 64]; \
    }
#define WMAR_LN_PROLOGUE                                                                        \
    if (LN) {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            ln_row_stats(a.stats, a.n_chunks, a.MT * 32, (mt0 + i) * 32 + (lane & 31), a.K, &mu[i], &rstd[i]); \
    }
#define WMAR_LNX(XV)
#define WMAR_MMA1(WV, XV)                                                                       \
    {                                                                                           \
        if (ABL == 3) {                                                                         \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                    \
                acc[0][i][0] += WV[0].x * XV[i].x + WV[0].y * XV[i].y + WV[0].z * XV[i].z + WV[0].w * XV[i].w; \
        } else {                                                                                \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                        \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV[t].x, XV[i].x, acc[t][i], 0, 0, 0); \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                        \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV[t].y, XV[i].y, acc[t][i], 0, 0, 0); \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                        \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV[t].z, XV[i].z, acc[t][i], 0, 0, 0); \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                        \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV[t].w, XV[i].w, acc[t][i], 0, 0, 0); \
        }                                                                                       \
    }
// the normalisation of k-block u+1 is issued ahead of the MFMAs of k-block u (VALU under MFMA),
// and each k-block only waits for its own loads
#define WMAR_MMA(WBUF, XBUF)                                                                    \
    WMAR_LNX(XBUF[0])                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                            \
        if (u + 1 < U) { WMAR_LNX(XBUF[u + 1]) }                                                \
        WMAR_MMA1(WBUF[u], XBUF[u])                                                             \
    }
    int kb = kb0;
    const int nfull = (kb1 - kb0) / (2 * U);
    if (nfull > 0) {
        // Every workgroup of a launch walks the SAME activation rows.  Started in lockstep they
        // would all hit the same few L2 channels at once (measured: 4.7 TB/s aggregate instead of
        // >30), so each output tile starts its K walk at a different stage and wraps around.
        // The summation order per tile stays fixed (it depends on the tile index only).
        const int nst = 2 * nfull;
        const int rot = (ROT ? (nt * 5 + mg * 3) : 0) % nst;
#define WMAR_STAGE_KB(SI) (kb0 + (((SI) + rot) % nst) * U)
        // sched_barrier(0): hipcc otherwise sinks every load down to its first use (it minimises
        // registers), which serialises load -> wait -> 4 MFMAs.
        WMAR_LOAD(wA, xA, WMAR_STAGE_KB(0))
        __builtin_amdgcn_sched_barrier(0);
        // LayerNorm statistics are fetched AFTER the first operand loads are in flight
        WMAR_LN_PROLOGUE
        __builtin_amdgcn_sched_barrier(0);
        WMAR_ST_LANDED(2)
#ifdef WMAR_GEMM_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr1 = __builtin_amdgcn_s_memtime();
#endif
#if WMAR_GEMM_INTERLEAVE > 0
        // one load between every WMAR_GEMM_INTERLEAVE MFMAs: a wave issues in order, so a batch of loads in front of the MFMAs
        // keeps the matrix pipe idle for as long as the batch takes to issue (~30 cycles per 1 KiB load); one load fits in the
        // 64-cycle shadow of an MFMA
#define WMAR_INTERLEAVE()                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < U * (NTW + MTW); ++i_) {                          \
            __builtin_amdgcn_sched_group_barrier(0x008, WMAR_GEMM_INTERLEAVE, 0);                  \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                     \
        }                                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * U * MTW * NTW - WMAR_GEMM_INTERLEAVE * U * (NTW + MTW), 0);
        for (int it = 0; it < nfull; ++it) {
            WMAR_LOAD(wB, xB, WMAR_STAGE_KB(2 * it + 1))
            WMAR_MMA(wA, xA)
            WMAR_INTERLEAVE()
            __builtin_amdgcn_sched_barrier(0);
            WMAR_LOAD(wA, xA, WMAR_STAGE_KB((2 * it + 2) % nst))
            WMAR_MMA(wB, xB)
            WMAR_INTERLEAVE()
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WMAR_INTERLEAVE
#else
        for (int it = 0; it < nfull; ++it) {
            WMAR_LOAD(wB, xB, WMAR_STAGE_KB(2 * it + 1))
            __builtin_amdgcn_sched_barrier(0);
            WMAR_MMA(wA, xA)
            __builtin_amdgcn_sched_barrier(0);
            // last round: re-read an in-bounds stage instead of branching around the loads
            WMAR_LOAD(wA, xA, WMAR_STAGE_KB((2 * it + 2) % nst))
            __builtin_amdgcn_sched_barrier(0);
            WMAR_MMA(wB, xB)
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#undef WMAR_STAGE_KB
        kb = kb0 + nst * U;
    } else {
        WMAR_LN_PROLOGUE
    }
    // tail (slices that are not a multiple of 2U blocks, at most 2U-1 of them): every load is issued before the
    // first wait -- one k-block at a time would expose a full memory round trip per block
    if (kb < kb1) {
        const int nt_ = kb1 - kb;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < nt_) {             // wave-uniform: only the blocks that exist are fetched
                _Pragma("unroll") for (int t = 0; t < NTW; ++t) wA[u][t] = ld_nt(Wp + t * wtile + (long long)(kb + u) * 64);
#pragma unroll
                for (int i = 0; i < MTW; ++i) xA[u][i] = Xp[(long long)(kb + u) * xstep + i * 64];
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (U + u < nt_) {
                _Pragma("unroll") for (int t = 0; t < NTW; ++t) wB[u][t] = ld_nt(Wp + t * wtile + (long long)(kb + U + u) * 64);
#pragma unroll
                for (int i = 0; i < MTW; ++i) xB[u][i] = Xp[(long long)(kb + U + u) * xstep + i * 64];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < nt_) { WMAR_LNX(xA[u]) WMAR_MMA1(wA[u], xA[u]) }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (U + u < nt_) { WMAR_LNX(xB[u]) WMAR_MMA1(wB[u], xB[u]) }
    }
#undef WMAR_MMA1
#undef WMAR_LNX
#undef WMAR_LN_PROLOGUE
#undef WMAR_LOAD
#undef WMAR_MMA

    WMAR_ST(3)
#ifdef WMAR_GEMM_TRACE
    tr2 = __builtin_amdgcn_s_memtime();
#endif
    // in-workgroup K reduction (fixed order) + epilogue
    // 16-byte LDS accesses: a wave parks its accumulators as MTW*4 float4 rows (the 4 registers of one output group are
    // adjacent) and the reducing wave reads one float4 per partner -- a quarter of the LDS instructions of a dword layout
    float4* smem4 = reinterpret_cast<float4*>(smem);
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                smem4[((long long)w * (NTW * MTW * 4) + (t * MTW + i) * 4 + g4) * 64 + lane] =
                    make_float4(acc[t][i][g4 * 4 + 0], acc[t][i][g4 * 4 + 1], acc[t][i][g4 * 4 + 2], acc[t][i][g4 * 4 + 3]);
    __syncthreads();

    const int nt_first = nt;
    for (int grp = w; grp < NTW * MTW * 4; grp += NW) {
        const int t_ = grp / (MTW * 4), i = (grp >> 2) % MTW, g = grp & 3;
        nt = nt_first + t_;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float4 t = smem4[((long long)ww * (NTW * MTW * 4) + (t_ * MTW + i) * 4 + g) * 64 + lane];
            o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
        }
        const int mt = mt0 + i;
        const int n = nt * 32 + g * 8 + half * 4;       // first of 4 consecutive output columns
        const int m = mt * 32 + (lane & 31);
        if (LN) {
            // LN(x) W^T = rstd * (x W'^T - mean * rowsum(W')) + (bias + W beta):  the main loop ran on raw x
            const float4 cc = *(const float4*)(a.c1 + n);
            const float mm = mu[i], rs = rstd[i];
            o[0] = rs * (o[0] - mm * cc.x); o[1] = rs * (o[1] - mm * cc.y);
            o[2] = rs * (o[2] - mm * cc.z); o[3] = rs * (o[3] - mm * cc.w);
        }
        if (a.bias) {
            float4 bb = *(const float4*)(a.bias + n);
            o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w;
        }
        if (EPI == EPI_PACKED || EPI == EPI_GELU) {
            if (EPI == EPI_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = gelu_erf(o[j]);
            }
            // output column n is input feature k' = n of the next GEMM: kb' = n/8
            if (EPI == EPI_GELU && a.out_planes) {
                bx_store_planes4(a.out_planes, a.MT, nt * 4 + g, lane >> 5, mt, lane & 31, make_float4(o[0], o[1], o[2], o[3]));
            } else {
                float4* dst = a.out_packed + (long long)s * a.slab_stride + ((long long)(nt * 4 + g) * a.MT + mt) * 64 + lane;
                st_out(dst, make_float4(o[0], o[1], o[2], o[3]));
            }
        } else if (EPI == EPI_QKV) {
            if (m < a.B) {
                const int which = n / a.D, c = n % a.D;
                if (which == 0) {
                    *(float4*)(a.qbuf + (long long)m * a.D + c) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    const int hh = c / a.hd, d = c % a.hd;
                    const int pos = *a.pos_dev;
                    float* base = (which == 1) ? a.kcache : a.vcache;
                    *(float4*)(base + (((long long)m * a.H + hh) * a.Tmax + pos) * a.hd + d) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        } else {  // EPI_LOGITS
            if (m < a.B) *(float4*)(a.logits + (long long)m * a.V + n) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    WMAR_ST_END(a.trace, blockIdx.x)
#ifdef WMAR_GEMM_TRACE
    if (a.trace && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* t = a.trace + (long long)blockIdx.x * 4;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// --------------------------------------------------------- FC1 (+ LayerNorm algebra + GELU) on 24-column tiles
// The GELU needs complete sums, so FC1 cannot be split over K across workgroups; with 32-column tiles its 6144 columns make 192
// workgroups and a quarter of the CUs idle.  6144 = 256 x 24: a workgroup of this kernel owns 24 columns x all 64 rows and
// the chip is full.  24 columns are formed at the full fp32 MFMA rate from two multi-block instructions that take the SAME B
// register -- one component of the packed activation as it lies in memory (lanes 0..31: rows of a 32-row tile at k, lanes
// 32..63: the same rows at k+4), no lane shuffles:
//     v_mfma_f32_16x16x1_4b_f32  blocks {0,1} = rows 0-15 / 16-31 at k, blocks {2,3} = the same rows at k+4; A (16 columns) is
//                                broadcast within each block PAIR (cbsz 1): lanes 0-31 of the A register carry two k-pairs'
//                                "k" columns, lanes 32-63 their "k+4" columns, abid picks the pair                32 cycles
//     v_mfma_f32_4x4x1_16b_f32   blocks 0-7 = 4-row groups at k, 8-15 at k+4; A (4 columns) broadcast within each OCTET (cbsz 3):
//                                one A register holds all eight k-pairs of a 16-k unit, abid picks the pair       8 cycles, twice
// = 2 x 48 cycles per k-pair for 24 x 64 outputs against 2 x 64 with v_mfma_f32_32x32x2_f32 on 32 columns.  The "k" and "k+4"
// partial sums of a row sit in different blocks (16x16: different registers, 4x4: lanes l and l+32) and are added once, after
// the K loop.  Four waves split K and meet in LDS (fixed order), as in k_gemm.  (First version: operands turned into "row =
// lane" registers with v_permlane32_swap and one accumulator set -- each swap in front of its MFMA cost ~20 cycles of idle pipe.)
#ifndef FX_ABL
#define FX_ABL 0     // dev ablations: 1 = no loads in the main loop, 4 = no 4x4 MFMAs
#endif
struct Fc1xArgs {
    const float4* W16;         // [N/24][K/16][64] x 4 registers: register q, lane 16*blk + col: pair 2q + (blk&1), k or k+4 by blk>>1
    const float2* W8;          // [N/24][K/16][64] x 2 registers (columns 16-19 / 20-23): lane 4*blk + col: pair blk&7, k or k+4 by blk>>3
    const float4* Xp;          // packed [K/8][2][64]
    const float* bias;         // [N] bias + W beta (LN folded)
    const float* c1;           // [N] row sums of the gamma-folded weights
    const double* stats; int n_chunks; int K;
    float4* out;               // packed hidden activation [N/8][2][64]
    int KU;                    // K / 16
    unsigned long long* trace; // dev only (WMAR_FX_TRACE): 4 timestamps per workgroup
};

// pair p = 4*kb + j of a 16-k unit covers k = 8*kb + j ("lo") and k + 4 ("hi")
static __global__ void k_pack_fc1x(const float* __restrict__ W, const float* __restrict__ gamma, float4* __restrict__ W16,
                                   float2* __restrict__ W8, int N, int K) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over (N/24) * (K/16) * 64
    const int KU = K / 16;
    if (idx >= (long long)(N / 24) * KU * 64) return;
    const int lane = (int)(idx & 63);
    const long long r = idx >> 6;
    const int kk = (int)(r % KU), tile = (int)(r / KU);
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int blk = lane >> 4, pair = 2 * q + (blk & 1);
        const int n = tile * 24 + (lane & 15), k = kk * 16 + 8 * (pair >> 2) + (pair & 3) + 4 * (blk >> 1);
        v[q] = W[(long long)n * K + k] * (gamma ? gamma[k] : 1.f);
    }
    W16[idx] = make_float4(v[0], v[1], v[2], v[3]);
    float u[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int blk = lane >> 2, pair = blk & 7;
        const int n = tile * 24 + 16 + 4 * g + (lane & 3), k = kk * 16 + 8 * (pair >> 2) + (pair & 3) + 4 * (blk >> 3);
        u[g] = W[(long long)n * K + k] * (gamma ? gamma[k] : 1.f);
    }
    W8[idx] = make_float2(u[0], u[1]);
}

static __global__ __launch_bounds__(256) void k_fc1x(Fc1xArgs a) {
    __shared__ __attribute__((aligned(16))) float4 red[4][6][64];
    __shared__ __attribute__((aligned(16))) double2 st_red[4][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    // K units (16 k) of this wave; the walk starts at a tile-dependent unit and wraps (all workgroups read the SAME activation
    // rows: started in lockstep they would queue on the same L2 channels)
    const int per = a.KU >> 2;                 // host: KU % 16 == 0
    const int u0 = w * per;
    const int rot = (tile * 5) % per;
#define WMAR_FX_UNIT(I) (u0 + (((I) + rot) >= per ? (I) + rot - per : (I) + rot))

    f32x16 acc16[2];
    f32x4 acc4a[2], acc4b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc16[i][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc4a[i][r] = 0.f; acc4b[i][r] = 0.f; }
    }

    const float4* W16 = a.W16 + (long long)tile * a.KU * 64 + lane;
    const float2* W8 = a.W8 + (long long)tile * a.KU * 64 + lane;
    const float4* Xp = a.Xp + lane;
    // four-unit register ring: the loads of unit i+3 are issued (one per k-pair, each right behind a 32-cycle MFMA) while unit i
    // is multiplied -- three units (~1.5 us) of distance, above the loaded HBM latency
    float4 xA[2][2], xB[2][2], xC[2][2], xD[2][2], wqA, wqB, wqC, wqD;
    float2 w8A, w8B, w8C, w8D;
#define WMAR_FX_LOAD(XB, WQ, W8V, UNIT)                                                          \
    {                                                                                             \
        const int un = (UNIT);                                                                    \
        WQ = ld_nt(W16 + (long long)un * 64);                                                     \
        W8V = ld_nt2(W8 + (long long)un * 64);                                                    \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                          \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
                XB[kb][i] = Xp[((long long)(un * 2 + kb) * 2 + i) * 64];                          \
    }
    // (the block-select immediates need compile-time constants: chains of ifs over the unrolled indices fold away)
    auto fx16 = [&](f32x16& c16, float bv, float aq, int sel) {          // 32 cycles; sel = pair & 1
        if (sel == 0) c16 = __builtin_amdgcn_mfma_f32_16x16x1f32(aq, bv, c16, 1, 0, 0);
        if (sel == 1) c16 = __builtin_amdgcn_mfma_f32_16x16x1f32(aq, bv, c16, 1, 1, 0);
    };
    auto fx4 = [&](f32x4& c4a, f32x4& c4b, float bv, const float2& w8v, int pair) {     // 2 x 8 cycles
        if (FX_ABL & 4) return;
#define WMAR_FX_CASE(PV)                                                                          \
        if (pair == PV) {                                                                         \
            c4a = __builtin_amdgcn_mfma_f32_4x4x1f32(w8v.x, bv, c4a, 3, PV, 0);                   \
            c4b = __builtin_amdgcn_mfma_f32_4x4x1f32(w8v.y, bv, c4b, 3, PV, 0);                   \
        }
        WMAR_FX_CASE(0) WMAR_FX_CASE(1) WMAR_FX_CASE(2) WMAR_FX_CASE(3) WMAR_FX_CASE(4) WMAR_FX_CASE(5) WMAR_FX_CASE(6) WMAR_FX_CASE(7)
#undef WMAR_FX_CASE
    };
// One unit (16 k = 8 pairs x 6 MFMAs); the six loads of the unit three ahead ride one per pair behind a 16x16 MFMA: a wave issues
// in order, so anything in front of an MFMA that is ready idles the matrix pipe.
#define WMAR_FX_MMA(XC, WQ, W8V, XL, WQL, W8L, UNITL, LOADQ)                                     \
    {                                                                                             \
        const int un_ = (UNITL);                                                                  \
        _Pragma("unroll") for (int pair = 0; pair < 8; ++pair) {                                  \
            const int kb = pair >> 2, j = pair & 3, q = pair >> 1;                                \
            const float aq = q == 0 ? WQ.x : (q == 1 ? WQ.y : (q == 2 ? WQ.z : WQ.w));            \
            const float b0 = j == 0 ? XC[kb][0].x : (j == 1 ? XC[kb][0].y : (j == 2 ? XC[kb][0].z : XC[kb][0].w)); \
            const float b1 = j == 0 ? XC[kb][1].x : (j == 1 ? XC[kb][1].y : (j == 2 ? XC[kb][1].z : XC[kb][1].w)); \
            fx16(acc16[0], b0, aq, pair & 1);                                                     \
            if (!(FX_ABL & 1) && (LOADQ)) {                                                       \
                if (pair == 0) WQL = ld_nt(W16 + (long long)un_ * 64);                            \
                if (pair == 1) W8L = ld_nt2(W8 + (long long)un_ * 64);                            \
                if (pair >= 2 && pair < 6)                                                        \
                    XL[(pair - 2) >> 1][(pair - 2) & 1] = Xp[((long long)(un_ * 2 + ((pair - 2) >> 1)) * 2 + ((pair - 2) & 1)) * 64]; \
            }                                                                                     \
            __builtin_amdgcn_sched_barrier(0);   /* hipcc otherwise clusters the dependent MFMAs of one accumulator */ \
            fx4(acc4a[0], acc4b[0], b0, W8V, pair);                                               \
            fx16(acc16[1], b1, aq, pair & 1);                                                     \
            fx4(acc4a[1], acc4b[1], b1, W8V, pair);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                         \
    }

    WMAR_ST_BEGIN
#ifdef WMAR_FX_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
#endif
    WMAR_FX_LOAD(xA, wqA, w8A, WMAR_FX_UNIT(0))
    WMAR_FX_LOAD(xB, wqB, w8B, WMAR_FX_UNIT(1))
    WMAR_FX_LOAD(xC, wqC, w8C, WMAR_FX_UNIT(2))
    __builtin_amdgcn_sched_barrier(0);
    // LayerNorm statistics of row = lane, fetched behind the first operands.  Round 5: wave w fetches chunks w, w + 4, ... only (four
    // 1-KiB loads instead of sixteen per wave: 48 KiB less in the CU's first burst) and the four partial sums meet in LDS behind the
    // K loop, in wave order.  (host: n_chunks <= 16)
    double st_sm = 0, st_sq = 0;
    {
        double2 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(const double2*)(a.stats + ((long long)min(w + 4 * i, a.n_chunks - 1) * 64 + lane) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (w + 4 * i < a.n_chunks) { st_sm += v[i].x; st_sq += v[i].y; }
    }
    // the epilogue's column constants (folded-LN row sums, bias) of this wave's output groups gi = w and gi = w + 4 (waves 0, 1):
    // requested here (round 5) -- inside the epilogue they were a dependent L2 round trip per group with every store behind it
    // (in-loop stamps: 2.5 us from the last MFMA to the last store acknowledged)
    const int nA_ = tile * 24 + 4 * (lane >> 4), nB_ = tile * 24 + 16 + 4 * (w & 1);
    const float4 ccA_ = *(const float4*)(a.c1 + nA_), bbA_ = *(const float4*)(a.bias + nA_);
    const float4 ccB_ = *(const float4*)(a.c1 + nB_), bbB_ = *(const float4*)(a.bias + nB_);
    __builtin_amdgcn_sched_barrier(0);
    WMAR_ST_LANDED(2)
#ifdef WMAR_FX_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
#endif
    int it = 0;
    for (; it + 4 < per; it += 4) {
        WMAR_FX_MMA(xA, wqA, w8A, xD, wqD, w8D, WMAR_FX_UNIT(it + 3), true)
        WMAR_FX_MMA(xB, wqB, w8B, xA, wqA, w8A, WMAR_FX_UNIT(it + 4), true)
        WMAR_FX_MMA(xC, wqC, w8C, xB, wqB, w8B, WMAR_FX_UNIT(it + 5), true)
        WMAR_FX_MMA(xD, wqD, w8D, xC, wqC, w8C, WMAR_FX_UNIT(it + 6), true)
    }
    // the last four units: only the first still has a unit to request (nothing is read past the wave's K range)
    WMAR_FX_MMA(xA, wqA, w8A, xD, wqD, w8D, WMAR_FX_UNIT(it + 3), true)
    WMAR_FX_MMA(xB, wqB, w8B, xA, wqA, w8A, 0, false)
    WMAR_FX_MMA(xC, wqC, w8C, xB, wqB, w8B, 0, false)
    WMAR_FX_MMA(xD, wqD, w8D, xC, wqC, w8C, 0, false)
#undef WMAR_FX_LOAD
#undef WMAR_FX_MMA
#undef WMAR_FX_UNIT

    WMAR_ST(3)
#ifdef WMAR_FX_TRACE
    const unsigned long long tr2 = __builtin_amdgcn_s_memtime();
#endif
    // "k" + "k+4" partial sums, then the cross-wave K reduction in LDS (fixed order), LayerNorm algebra, bias, GELU, packed store.
    // red[w][g]: g = 2*i + b: 16x16 part, rows 32 i + 16 b + lane%16, columns 4*(lane/16)..+3;  g = 4 / 5: columns 16-19 / 20-23, row = lane
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            red[w][2 * i + b][lane] = make_float4(acc16[i][4 * b + 0] + acc16[i][4 * b + 8], acc16[i][4 * b + 1] + acc16[i][4 * b + 9],
                                                  acc16[i][4 * b + 2] + acc16[i][4 * b + 10], acc16[i][4 * b + 3] + acc16[i][4 * b + 11]);
    {
        float v4[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // lanes l and l+32 of a 4x4 accumulator hold the two partial sums of row 32 i + l: the swap lines them up as
            // [tile 0 | tile 1] rows in lane order
            const auto sa = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc4a[0][r]), __float_as_uint(acc4a[1][r]), false, false);
            const auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc4b[0][r]), __float_as_uint(acc4b[1][r]), false, false);
            v4[0][r] = __uint_as_float(sa[0]) + __uint_as_float(sa[1]);
            v4[1][r] = __uint_as_float(sb[0]) + __uint_as_float(sb[1]);
        }
        red[w][4][lane] = make_float4(v4[0][0], v4[0][1], v4[0][2], v4[0][3]);
        red[w][5][lane] = make_float4(v4[1][0], v4[1][1], v4[1][2], v4[1][3]);
    }
    st_red[w][lane] = make_double2(st_sm, st_sq);
    __syncthreads();
    float mu, rstd;
    {
        double sm = 0, sq = 0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { const double2 t = st_red[ww][lane]; sm += t.x; sq += t.y; }
        const double invK = inv_count_f64((double)a.K);
        const double mean = sm * invK;
        mu = (float)mean;
        rstd = rsqrtf((float)var_f64(sq * invK, mean) + 1e-5f);
    }
    for (int gi = w; gi < 6; gi += 4) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float4 t = red[ww][gi][lane];
            o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
        }
        const int m = gi < 4 ? 16 * gi + (lane & 15) : lane;
        const int n = tile * 24 + (gi < 4 ? 4 * (lane >> 4) : 16 + 4 * (gi - 4));
        const float mm = __shfl(mu, m), rs = __shfl(rstd, m);
        const float4 cc = gi < 4 ? ccA_ : ccB_;
        const float4 bb = gi < 4 ? bbA_ : bbB_;
        o[0] = gelu_erf(rs * (o[0] - mm * cc.x) + bb.x); o[1] = gelu_erf(rs * (o[1] - mm * cc.y) + bb.y);
        o[2] = gelu_erf(rs * (o[2] - mm * cc.z) + bb.z); o[3] = gelu_erf(rs * (o[3] - mm * cc.w) + bb.w);
        st_out(a.out + ((long long)(n >> 3) * 2 + (m >> 5)) * 64 + (m & 31) + 32 * ((n >> 2) & 1), make_float4(o[0], o[1], o[2], o[3]));
    }
    WMAR_ST_END(a.trace, blockIdx.x)
#ifdef WMAR_FX_TRACE
    if (a.trace && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* t = a.trace + (long long)blockIdx.x * 4;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// ------------------------------------------------- residual fold + LN1 statistics + QKV projection in ONE launch
// The QKV GEMM of layer l consumes the residual stream  x' = x + bias + sum_s slab[s]  of layer l-1's FC2 (k_resid_stats does
// that fold as a launch of its own for the other consumers).  Here the fold happens while the operand is staged:
//   * grid = G column groups (4 x 32 columns: one tile per wave, NO in-workgroup reduction) x S slices of K -- 36 x 7 = 252
//     workgroups at n_embd 1536 instead of 144 column tiles that leave 112 CUs idle;
//   * the 4 waves of a workgroup share the SAME K slice, so x' is formed ONCE per workgroup, chunk by chunk (4 k-blocks: one
//     per wave), written to LDS and read back by all four as the MFMA B operand: the (1 + S_in)-fold read of the fold costs
//     ~14 KiB per k-block against 4 KiB of weights, inside the CU's L1 fill rate;
//   * column group 0 also writes x' to the OTHER residual buffer (the other groups still read the old one) and the fp64 row
//     sums of its K slice: S partial statistics per row for the attention prologue's LayerNorm algebra;
//   * every wave stores its accumulators as one of S split-K pieces, which the attention prologue sums in slice order.
// Weights: a wave walks its own column tile, loads issued two chunks (2 x 2048 MFMA cycles) ahead of use.
constexpr int QX_CK = 4;      // k-blocks per chunk (one staged by each wave)
#ifndef QX_ABL
#define QX_ABL 0     // dev ablations (trace builds): 1 = no weight loads in the loop, 2 = no chunk barriers, 4 = no LDS reads in the loop
#endif
#define WMAR_QX_SYNC() if (!(QX_ABL & 2)) __syncthreads();
#ifndef QX_SPLIT
#define QX_SPLIT 2             // k-blocks multiplied before the chunk barrier; the next chunk's LDS reads land under the rest
#endif

struct QkvxArgs {
    const float4* Wp;          // [NT][KB][64] gamma-folded QKV weights
    const float4* x_in;        // packed [KB][MT][64]: residual stream before the fold
    float4* x_out;             // residual stream after the fold (a different buffer)
    const float4* slabs;       // [S_IN][KB*MT*64]: split-K partial sums of the previous FC2
    long long slab_stride;     // float4 units
    int n_hi;                  // > 0: slab S_IN-1 exists only for column tiles < n_hi (see ResidArgs)
    const float* bias;         // [K] bias of the previous FC2 (S_IN > 0)
    double* stats;             // [S][Mpad][2]: (sum, sum of squares) of x' over K slice s
    float4* out;               // [S][3D/8][MT][64] split-K pieces of the projection
    long long out_stride;      // float4 units
    int KB, NT, S, cap;        // cap: workgroup slots per XCD (grid = 8 * cap)
    int nkeep;                 // k_qkvx_bx: column groups per K slice that share the keeper duty (statistics chunks = S * nkeep)
    unsigned long long* trace; // dev only (WMAR_QX_TRACE): 4 timestamps per wave
    unsigned long long* trace_chunks;   // dev only: start of each chunk relative to the first, per wave
};

// MTW = row tiles (all of them: MT == MTW); S_IN = slabs folded into x (0: x is used as it is, layer 0).  Both are template
// parameters so that EVERY load is unconditional and counted at compile time: hipcc's s_waitcnt placement then waits for exactly
// the operands a step needs.  Out-of-range k-blocks of a short last chunk re-read an in-bounds block instead of branching.
//
// Eight waves, two roles (one wave of each role per SIMD):
//   * waves 4..7 STAGE: wave 4+i forms k-block i of every chunk -- x + (bias + slabs in slab order), the arithmetic of
//     k_resid_stats -- and writes it to LDS; its loads run two chunks ahead of the chunk it finishes (two register sets), so
//     the L2 / fabric latency of the slabs never reaches the matrix pipe.  In column group 0 they also publish x' and the
//     fp64 row sums of their slice.
//   * waves 0..3 MULTIPLY: wave i owns column tile 4g+i; weights stream through a 4-chunk register ring (requested three
//     chunks = ~6000 cycles ahead of use), the B operand comes from LDS.  The fragments of chunk c+1 are read under the last 8
//     MFMAs of chunk c, so the MFMA stream has no bubbles at chunk boundaries.
// One barrier per chunk: barrier(c) = "chunk c+1 is in LDS" and "chunk c has been read" (the reads of a chunk happen at its start,
// so the multiplying waves reach the barrier three quarters into the chunk and the stagers have a whole chunk of time).
template <int MTW, int S_IN>
__global__ __launch_bounds__(512) void k_qkvx(QkvxArgs a) {
    __shared__ __attribute__((aligned(16))) float4 xs[2][QX_CK][MTW][64];
    __shared__ double red[4][MTW][64][2];      // every lane publishes its own partial sums (see the note at the keeper's reduction)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    // workgroups with the same K slice share an XCD (block b runs on XCD b % 8): the slice of x and of the slabs is then
    // fetched into that XCD's L2 once.  Placement only changes speed, never results.
    const int G = a.NT >> 2;
    const int j = (int)(blockIdx.x & 7) * a.cap + (int)(blockIdx.x >> 3);
    if (j >= G * a.S) return;
    const int s = j / G, g = j - s * G;
    const int kb0 = (int)((unsigned)s * (unsigned)a.KB / (unsigned)a.S);
    const int kb1 = (int)((unsigned)(s + 1) * (unsigned)a.KB / (unsigned)a.S);
    const int nkb = kb1 - kb0, nch = (nkb + QX_CK - 1) / QX_CK;
    const bool keeper = g == 0;          // this workgroup also publishes x' and the row sums of its slice

    if (w >= 4) {
        // ------------------------------------------------------------------------------------------ staging waves
        const int sw = w - 4;
        double sum[MTW], sq[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) { sum[i] = 0.0; sq[i] = 0.0; }
        float4 xvA[MTW], bbA, slA[S_IN > 0 ? S_IN : 1][MTW];
        float4 xvB[MTW], bbB, slB[S_IN > 0 ? S_IN : 1][MTW];
        int skbA = 0, skbB = 0;
#define WMAR_QX_ISSUE(XV, BB, SL, SKB, C)                                                              \
    {                                                                                                   \
        SKB = kb0 + (C) * QX_CK + sw;                                                                   \
        const int kk = SKB < kb1 ? SKB : kb1 - 1;                                                       \
        const long long idx = (long long)kk * MTW * 64 + lane;                                          \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) XV[i] = a.x_in[idx + i * 64];                   \
        if (S_IN > 0) {                                                                                 \
            BB = *(const float4*)(a.bias + kk * 8 + 4 * half);                                          \
            _Pragma("unroll") for (int si = 0; si < S_IN; ++si)                                         \
                _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                         \
                    SL[si][i] = a.slabs[(long long)si * a.slab_stride + idx + i * 64];                  \
        }                                                                                               \
    }
#define WMAR_QX_FINISH(XV, BB, SL, SKB, BUF)                                                           \
    if (SKB < kb1) {                                                                                    \
        const bool short_tile = a.n_hi > 0 && (SKB >> 2) >= a.n_hi;                                     \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) {                                               \
            float4 r = XV[i];                                                                           \
            if (S_IN > 0) {                                                                             \
                float4 t = SL[0][i];                                                                    \
                _Pragma("unroll") for (int si = 1; si < S_IN; ++si)                                     \
                    if (!(si == S_IN - 1 && short_tile)) {                                              \
                        t.x += SL[si][i].x; t.y += SL[si][i].y; t.z += SL[si][i].z; t.w += SL[si][i].w; \
                    }                                                                                   \
                r = make_float4(r.x + (BB.x + t.x), r.y + (BB.y + t.y), r.z + (BB.z + t.z), r.w + (BB.w + t.w)); \
            }                                                                                           \
            xs[BUF][sw][i][lane] = r;                                                                   \
            if (keeper) {                                                                               \
                a.x_out[((long long)SKB * MTW + i) * 64 + lane] = r;                                    \
                sum[i] += (double)r.x + (double)r.y + (double)r.z + (double)r.w;                        \
                sq[i] += sq4_f64(r);            /* never a v_fmac_f64 chain: common.h */                \
            }                                                                                           \
        }                                                                                               \
    }
        WMAR_QX_ISSUE(xvA, bbA, slA, skbA, 0)
        WMAR_QX_ISSUE(xvB, bbB, slB, skbB, 1)
        __builtin_amdgcn_sched_barrier(0);
        // sched_barrier(0) after every block: hipcc otherwise sinks the loads of a set down to the FINISH that consumes them
        // (it minimises register lifetimes), which would expose one full memory round trip per chunk
        WMAR_QX_FINISH(xvA, bbA, slA, skbA, 0)
        __builtin_amdgcn_sched_barrier(0);
        WMAR_QX_ISSUE(xvA, bbA, slA, skbA, 2)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                   // barrier(-1): chunk 0 is in LDS
        for (int c = 0; c + 1 < nch; c += 2) {
            WMAR_QX_FINISH(xvB, bbB, slB, skbB, 1)         // chunk c+1
            __builtin_amdgcn_sched_barrier(0);
            WMAR_QX_ISSUE(xvB, bbB, slB, skbB, c + 3)
            __builtin_amdgcn_sched_barrier(0);
            WMAR_QX_SYNC()                                 // barrier(c)
            if (c + 2 >= nch) break;
            WMAR_QX_FINISH(xvA, bbA, slA, skbA, 0)         // chunk c+2
            __builtin_amdgcn_sched_barrier(0);
            WMAR_QX_ISSUE(xvA, bbA, slA, skbA, c + 4)
            __builtin_amdgcn_sched_barrier(0);
            WMAR_QX_SYNC()                                 // barrier(c+1)
        }
#undef WMAR_QX_ISSUE
#undef WMAR_QX_FINISH
        if (keeper) {
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                red[sw][i][lane][0] = sum[i]; red[sw][i][lane][1] = sq[i];
            }
            __syncthreads();                               // the multiplying waves meet it after their stores
            const int t = threadIdx.x - 256;
            if (t < 32 * MTW) {
                const int i = t >> 5, r = t & 31;
                double ts = 0, tss = 0;
                for (int ww = 0; ww < 4; ++ww) { ts += red[ww][i][r][0] + red[ww][i][r + 32][0]; tss += red[ww][i][r][1] + red[ww][i][r + 32][1]; }
                double* o = a.stats + ((long long)s * (MTW * 32) + i * 32 + r) * 2;
                o[0] = ts; o[1] = tss;
            }
        }
        return;
    }

    // ------------------------------------------------------------------------------------------ multiplying waves
#ifdef QX_PRIO
    __builtin_amdgcn_s_setprio(QX_PRIO);
#endif
    const int nt = g * 4 + w;
    f32x16 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float4* Wp = a.Wp + ((long long)nt * a.KB + kb0) * 64 + lane;
    float4 w0[QX_CK], w1[QX_CK], w2[QX_CK], w3[QX_CK];
    float4 xfA[QX_CK][MTW], xfB[QX_CK][MTW];
#define WMAR_QX_W(WBUF, C)                                                                             \
    _Pragma("unroll") for (int u = 0; u < QX_CK; ++u) {                                                 \
        const int kl = (C) * QX_CK + u;                                                                 \
        WBUF[u] = ld_nt(Wp + (long long)(kl < nkb ? kl : nkb - 1) * 64);                                \
    }
#define WMAR_QX_READ(XF, BUF)                                                                          \
    _Pragma("unroll") for (int u = 0; u < QX_CK; ++u)                                                   \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) XF[u][i] = xs[BUF][u][i][lane];
#define WMAR_QX_MMA(WBUF, XF, C, U0, U1)                                                               \
    _Pragma("unroll") for (int u = U0; u < U1; ++u)                                                     \
        if ((C) * QX_CK + u < nkb) {                                                                    \
            const float4 wv = WBUF[u];                                                                  \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                             \
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, XF[u][i].x, acc[i], 0, 0, 0);       \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                             \
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, XF[u][i].y, acc[i], 0, 0, 0);       \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                             \
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, XF[u][i].z, acc[i], 0, 0, 0);       \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                             \
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, XF[u][i].w, acc[i], 0, 0, 0);       \
        }
// one chunk: request the weights three chunks ahead, multiply three k-blocks, pass the chunk barrier and read the next chunk's
// fragments, multiply the last k-block
#ifdef WMAR_QX_TRACE
#define WMAR_QX_STAMP(C) if ((C) < 8) trc[(C)] = __builtin_amdgcn_s_memtime();
#else
#define WMAR_QX_STAMP(C)
#endif
#define WMAR_QX_STEP(C, WCUR, WFAR, XCUR, XNEXT, BUFNEXT)                                              \
    WMAR_QX_STAMP(C)                                                                                    \
    if (!(QX_ABL & 1)) { WMAR_QX_W(WFAR, (C) + 3) }                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    WMAR_QX_MMA(WCUR, XCUR, C, 0, QX_SPLIT)                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if ((C) + 1 < nch) {                                                                                \
        WMAR_QX_SYNC()                                                                                  \
        if (!(QX_ABL & 4)) { WMAR_QX_READ(XNEXT, BUFNEXT) }                                             \
    }                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    WMAR_QX_MMA(WCUR, XCUR, C, QX_SPLIT, QX_CK)                                                         \
    __builtin_amdgcn_sched_barrier(0);

#ifdef WMAR_QX_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
    unsigned long long trc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    WMAR_QX_W(w0, 0)
    WMAR_QX_W(w1, 1)
    WMAR_QX_W(w2, 2)
    __syncthreads();                                       // barrier(-1)
#ifdef WMAR_QX_TRACE
    const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
#endif
    WMAR_QX_READ(xfA, 0)
    for (int c = 0; c < nch; c += 4) {
        WMAR_QX_STEP(c, w0, w3, xfA, xfB, 1)
        if (c + 1 >= nch) break;
        WMAR_QX_STEP(c + 1, w1, w0, xfB, xfA, 0)
        if (c + 2 >= nch) break;
        WMAR_QX_STEP(c + 2, w2, w1, xfA, xfB, 1)
        if (c + 3 >= nch) break;
        WMAR_QX_STEP(c + 3, w3, w2, xfB, xfA, 0)
    }
#undef WMAR_QX_W
#undef WMAR_QX_READ
#undef WMAR_QX_MMA
#undef WMAR_QX_STEP

#ifdef WMAR_QX_TRACE
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long tr2 = __builtin_amdgcn_s_memtime();
#endif
    // one split-K piece per wave, straight from the accumulators (already the packed layout of the consumer)
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            st_out(a.out + (long long)s * a.out_stride + ((long long)(nt * 4 + q4) * MTW + i) * 64 + lane,
                   make_float4(acc[i][q4 * 4 + 0], acc[i][q4 * 4 + 1], acc[i][q4 * 4 + 2], acc[i][q4 * 4 + 3]));
    if (keeper) __syncthreads();
#ifdef WMAR_QX_TRACE
    if (a.trace && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* t = a.trace + ((long long)j * 4 + w) * 4;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memtime();
        if (a.trace_chunks) for (int i = 0; i < 8; ++i) a.trace_chunks[((long long)j * 4 + w) * 8 + i] = trc[i] ? trc[i] - tr1 : 0;
    }
#endif
}

template <int MTW>
static int launch_qkvx_mt(const QkvxArgs& q, int S_in, hipStream_t st) {
    const dim3 grid((unsigned)(8 * q.cap));
    switch (S_in) {
#define WMAR_QX_CASE(N) case N: hipLaunchKernelGGL((k_qkvx<MTW, N>), grid, dim3(512), 0, st, q); break;
        WMAR_QX_CASE(0) WMAR_QX_CASE(1) WMAR_QX_CASE(2) WMAR_QX_CASE(3) WMAR_QX_CASE(4)
        WMAR_QX_CASE(5) WMAR_QX_CASE(6) WMAR_QX_CASE(7) WMAR_QX_CASE(8)
#undef WMAR_QX_CASE
        default: set_error("qkvx: bad slab count %d", S_in); return WMAR_EINVAL;
    }
    return launch_status("k_qkvx");
}
static int launch_qkvx(const QkvxArgs& q, int MT, int S_in, hipStream_t st) {
    if (MT == 1) return launch_qkvx_mt<1>(q, S_in, st);
    if (MT == 2) return launch_qkvx_mt<2>(q, S_in, st);
    set_error("qkvx: %d row tiles unsupported", MT);
    return WMAR_EINVAL;
}

// ------------------------------------------------- the same launch on the bf16 matrix pipe (64 rows)
// k_qkvx with the multiplying waves on v_mfma_f32_32x32x16_bf16 (bx_split.h: six bf16 piece products per fp32 product, fp32
// accuracy): a chunk of 32 k is 2 x 12 MFMAs of 32 cycles instead of 32 of 64.  The staging waves split x' into its three bf16
// pieces once per workgroup and write them to LDS in the B-operand layout [step][row tile][piece][lane] (8-byte stores); the
// multiplying waves read 16-byte operands, keep their fp32 weights (Wq: k_pack_bx layout, no extra HBM bytes) in the same
// 4-chunk ring and split them in registers between the MFMAs.  K slices are cut on 16-k steps.
static __global__ void k_pack_bx(const float* __restrict__ W, const float* __restrict__ gamma, float4* __restrict__ Wq, int N, int K,
                                      int tile_off) {
    // Wq[tile][ku][half][lane] float4: lane holds W[n = 32 tile + lane % 32][k = 16 ku + 8 (lane / 32) + 4 half + 0..3] * gamma[k]
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KU = K / 16;
    if (idx >= (long long)(N / 32) * KU * 128) return;
    const int lane = (int)(idx & 63), hf = (int)((idx >> 6) & 1);
    const long long r = idx >> 7;
    const int ku = (int)(r % KU), tile = (int)(r / KU);
    const int k = ku * 16 + 8 * (lane >> 5) + 4 * hf;
    const float* p = W + (long long)(tile * 32 + (lane & 31)) * K + k;
    float4 v = make_float4(p[0], p[1], p[2], p[3]);
    if (gamma) { v.x *= gamma[k]; v.y *= gamma[k + 1]; v.z *= gamma[k + 2]; v.w *= gamma[k + 3]; }
    Wq[idx + (long long)tile_off * KU * 128] = v;
}

constexpr int QX_TR_STRIDE = 36;      // floats per row of the epilogue's transpose tile (32 + 4: 16-byte rows on distinct banks)
template <int S_IN>
__global__ __launch_bounds__(512) void k_qkvx_bx(QkvxArgs a) {
    constexpr int MTW = 2;
    __shared__ __attribute__((aligned(16))) u32x4 xq[2][2][MTW][3][64];      // [buffer][step][row tile][piece][lane]
    __shared__ __attribute__((aligned(16))) float tr_s[4][32 * QX_TR_STRIDE];   // epilogue: one 32 x 32 tile per multiplying wave
    __shared__ double red[4][MTW][64][2];      // every lane publishes its own partial sums (see the note at the keeper's reduction)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int G = a.NT >> 2;
    const int j = (int)(blockIdx.x & 7) * a.cap + (int)(blockIdx.x >> 3);
    if (j >= G * a.S) return;
    const int s = j / G, g = j - s * G;
    const int KU = a.KB >> 1;                                       // 16-k steps; slices are cut on steps
    const int kb0 = 2 * (int)((unsigned)s * (unsigned)KU / (unsigned)a.S);
    const int kb1 = 2 * (int)((unsigned)(s + 1) * (unsigned)KU / (unsigned)a.S);
    const int nkb = kb1 - kb0, nch = (nkb + QX_CK - 1) / QX_CK;
    // Keeper duty -- publishing x' and the fp64 row sums of the K slice -- is shared by the first `nkeep` column groups of a slice
    // (round 5; chunk c belongs to group c % nkeep): with ONE keeper per slice its stagers' extra stores and fp64 sums made those 7
    // workgroups the launch's stragglers (in-loop stamps: last workgroup out 2.2 us after the mean).  Statistics chunk = s * nkeep + g.
    const bool keeper = g < a.nkeep;

    if (w >= 4) {
        // ------------------------------------------------------------------------------------------ staging waves
        const int sw = w - 4;
        double sum[MTW], sq[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) { sum[i] = 0.0; sq[i] = 0.0; }
        float4 xvA[MTW], bbA, slA[S_IN > 0 ? S_IN : 1][MTW];
        float4 xvB[MTW], bbB, slB[S_IN > 0 ? S_IN : 1][MTW];
        int skbA = 0, skbB = 0;
#define WMAR_QX_ISSUE(XV, BB, SL, SKB, C)                                                              \
    {                                                                                                   \
        SKB = kb0 + (C) * QX_CK + sw;                                                                   \
        const int kk = SKB < kb1 ? SKB : kb1 - 1;                                                       \
        const long long idx = (long long)kk * MTW * 64 + lane;                                          \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) XV[i] = a.x_in[idx + i * 64];                   \
        if (S_IN > 0) {                                                                                 \
            BB = *(const float4*)(a.bias + kk * 8 + 4 * half);                                          \
            _Pragma("unroll") for (int si = 0; si < S_IN; ++si)                                         \
                _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                         \
                    SL[si][i] = a.slabs[(long long)si * a.slab_stride + idx + i * 64];                  \
        }                                                                                               \
    }
// wave sw stages k-block sw of the chunk: step sw / 2, operand lanes m + 32 (sw % 2), this lane's 8 bytes = half
#define WMAR_QX_FINISH(XV, BB, SL, SKB, BUF, CIDX)                                                     \
    if (SKB < kb1) {                                                                                    \
        const bool short_tile = a.n_hi > 0 && (SKB >> 2) >= a.n_hi;                                     \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) {                                               \
            float4 r = XV[i];                                                                           \
            if (S_IN > 0) {                                                                             \
                float4 t = SL[0][i];                                                                    \
                _Pragma("unroll") for (int si = 1; si < S_IN; ++si)                                     \
                    if (!(si == S_IN - 1 && short_tile)) {                                              \
                        t.x += SL[si][i].x; t.y += SL[si][i].y; t.z += SL[si][i].z; t.w += SL[si][i].w; \
                    }                                                                                   \
                r = make_float4(r.x + (BB.x + t.x), r.y + (BB.y + t.y), r.z + (BB.z + t.z), r.w + (BB.w + t.w)); \
            }                                                                                           \
            unsigned h0, m0, l0, h1, m1, l1;                                                            \
            bx_split2(r.x, r.y, h0, m0, l0);                                                            \
            bx_split2(r.z, r.w, h1, m1, l1);                                                            \
            u32x2* d = (u32x2*)&xq[BUF][sw >> 1][i][0][(lane & 31) + 32 * (sw & 1)] + half;             \
            d[0] = u32x2{h0, h1}; d[128] = u32x2{m0, m1}; d[256] = u32x2{l0, l1};                       \
            if (keeper && (CIDX) % a.nkeep == g) {                                                      \
                a.x_out[((long long)SKB * MTW + i) * 64 + lane] = r;                                    \
                sum[i] += (double)r.x + (double)r.y + (double)r.z + (double)r.w;                        \
                sq[i] += sq4_f64(r);        /* never a v_fmac_f64 chain: common.h */                \
            }                                                                                           \
        }                                                                                               \
    }
        WMAR_QX_ISSUE(xvA, bbA, slA, skbA, 0)
        WMAR_QX_ISSUE(xvB, bbB, slB, skbB, 1)
        __builtin_amdgcn_sched_barrier(0);
        WMAR_QX_FINISH(xvA, bbA, slA, skbA, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
        WMAR_QX_ISSUE(xvA, bbA, slA, skbA, 2)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                   // barrier(-1): chunk 0 is in LDS
        for (int c = 0; c + 1 < nch; c += 2) {
            WMAR_QX_FINISH(xvB, bbB, slB, skbB, 1, c + 1)  // chunk c+1
            __builtin_amdgcn_sched_barrier(0);
            WMAR_QX_ISSUE(xvB, bbB, slB, skbB, c + 3)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                               // barrier(c)
            if (c + 2 >= nch) break;
            WMAR_QX_FINISH(xvA, bbA, slA, skbA, 0, c + 2)  // chunk c+2
            __builtin_amdgcn_sched_barrier(0);
            WMAR_QX_ISSUE(xvA, bbA, slA, skbA, c + 4)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                               // barrier(c+1)
        }
#undef WMAR_QX_ISSUE
#undef WMAR_QX_FINISH
        if (keeper) {
            // every lane stores its own (sum, sum of squares); the thread that owns row r adds lanes r and r + 32 of the four waves in a
            // fixed order (the same additions the 64-bit __shfl_xor(., 32) of earlier rounds made)
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                red[sw][i][lane][0] = sum[i]; red[sw][i][lane][1] = sq[i];
            }
            __syncthreads();                               // the multiplying waves meet it after their stores
            const int t = threadIdx.x - 256;
            if (t < 32 * MTW) {
                const int i = t >> 5, r = t & 31;
                double ts = 0, tss = 0;
                for (int ww = 0; ww < 4; ++ww) { ts += red[ww][i][r][0] + red[ww][i][r + 32][0]; tss += red[ww][i][r][1] + red[ww][i][r + 32][1]; }
                double* o = a.stats + ((long long)(s * a.nkeep + g) * (MTW * 32) + i * 32 + r) * 2;
                o[0] = ts; o[1] = tss;
            }
        }
        return;
    }

    // ------------------------------------------------------------------------------------------ multiplying waves
    WMAR_ST_BEGIN
    const int nt = g * 4 + w;
    f32x16 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // chunk c, slot u: step u / 2, half u % 2 of the weight tile's k-step (kb0 + 4 c) / 2 + u / 2
    const float4* Wq = a.Wp + ((long long)nt * KU + (kb0 >> 1)) * 128 + lane;
    float4 w0[QX_CK], w1[QX_CK], w2[QX_CK], w3[QX_CK];
    u32x4 xfA[2][MTW][3], xfB[2][MTW][3];
#define WMAR_QX_W(WBUF, C)                                                                             \
    _Pragma("unroll") for (int u = 0; u < QX_CK; ++u) {                                                 \
        const int kl = (C) * QX_CK + u;                                                                 \
        WBUF[u] = ld_nt(Wq + (long long)(kl < nkb ? kl : nkb - 1) * 64);                                \
    }
#define WMAR_QX_READ(XF, BUF)                                                                          \
    _Pragma("unroll") for (int st = 0; st < 2; ++st)                                                    \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                 \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) XF[st][i][p] = xq[BUF][st][i][p][lane];
#define WMAR_QX_BF(V) __builtin_bit_cast(bf16x8, V)
#define WMAR_QX_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
// one 16-k step: split the step's weights (VALU), twelve MFMAs, the small products first
#define WMAR_QX_MMA(WBUF, XF, C, ST)                                                                   \
    if ((C) * QX_CK + 2 * (ST) < nkb) {                                                                 \
        unsigned h_[4], m_[4], l_[4];                                                                   \
        bx_split2(WBUF[2 * (ST)].x, WBUF[2 * (ST)].y, h_[0], m_[0], l_[0]);                             \
        bx_split2(WBUF[2 * (ST)].z, WBUF[2 * (ST)].w, h_[1], m_[1], l_[1]);                             \
        bx_split2(WBUF[2 * (ST) + 1].x, WBUF[2 * (ST) + 1].y, h_[2], m_[2], l_[2]);                     \
        bx_split2(WBUF[2 * (ST) + 1].z, WBUF[2 * (ST) + 1].w, h_[3], m_[3], l_[3]);                     \
        const bf16x8 wh = WMAR_QX_BF((u32x4{h_[0], h_[1], h_[2], h_[3]})), wm = WMAR_QX_BF((u32x4{m_[0], m_[1], m_[2], m_[3]})), \
                     wl = WMAR_QX_BF((u32x4{l_[0], l_[1], l_[2], l_[3]}));                              \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) WMAR_QX_MFMA(wl, WMAR_QX_BF(XF[ST][i][0]), acc[i]); \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) WMAR_QX_MFMA(wh, WMAR_QX_BF(XF[ST][i][2]), acc[i]); \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) WMAR_QX_MFMA(wm, WMAR_QX_BF(XF[ST][i][1]), acc[i]); \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) WMAR_QX_MFMA(wm, WMAR_QX_BF(XF[ST][i][0]), acc[i]); \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) WMAR_QX_MFMA(wh, WMAR_QX_BF(XF[ST][i][1]), acc[i]); \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) WMAR_QX_MFMA(wh, WMAR_QX_BF(XF[ST][i][0]), acc[i]); \
    }
// one chunk: request the weights three chunks ahead, multiply the first step, pass the chunk barrier and read the next chunk's
// operands, multiply the second step
#define WMAR_QX_STEP(C, WCUR, WFAR, XCUR, XNEXT, BUFNEXT)                                              \
    WMAR_QX_W(WFAR, (C) + 3)                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    WMAR_QX_MMA(WCUR, XCUR, C, 0)                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if ((C) + 1 < nch) {                                                                                \
        __syncthreads();                                                                                \
        WMAR_QX_READ(XNEXT, BUFNEXT)                                                                    \
    }                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    WMAR_QX_MMA(WCUR, XCUR, C, 1)                                                                       \
    __builtin_amdgcn_sched_barrier(0);

    WMAR_QX_W(w0, 0)
    WMAR_QX_W(w1, 1)
    WMAR_QX_W(w2, 2)
    __syncthreads();                                       // barrier(-1)
    WMAR_ST_LANDED(2)                                      // chunk 0 staged (the stagers' first loads) and this wave's first weights
    WMAR_QX_READ(xfA, 0)
    for (int c = 0; c < nch; c += 4) {
        WMAR_QX_STEP(c, w0, w3, xfA, xfB, 1)
        if (c + 1 >= nch) break;
        WMAR_QX_STEP(c + 1, w1, w0, xfB, xfA, 0)
        if (c + 2 >= nch) break;
        WMAR_QX_STEP(c + 2, w2, w1, xfA, xfB, 1)
        if (c + 3 >= nch) break;
        WMAR_QX_STEP(c + 3, w3, w2, xfB, xfA, 0)
    }
    WMAR_ST(3)
#undef WMAR_QX_W
#undef WMAR_QX_READ
#undef WMAR_QX_MMA
#undef WMAR_QX_MFMA
#undef WMAR_QX_BF
#undef WMAR_QX_STEP
    // One split-K piece per wave, ROW-MAJOR (round 5): piece s is [64 rows][3 D] floats, so the attention workgroup of (sequence,
    // head) finds its q / k / v columns of a piece as 256 contiguous bytes.  In the packed operand layout the same 64 floats lay in
    // 16 different 128-byte lines shared with 31 other sequences: the attention prologue touched 384 lines (49 KB) per wave for
    // 5.4 KB of payload -- 75 MB of L2 -> L1 traffic per launch beside the 100 MB of K/V it streams.  The accumulators (row = lane % 32,
    // columns 8 q4 + 4 (lane / 32) ..+3 of the wave's 32-column tile) are turned through a wave-private LDS tile: a store instruction
    // then covers eight rows x 128 contiguous bytes.
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        float* T = &tr_s[w][0];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *(float4*)(T + (lane & 31) * QX_TR_STRIDE + 8 * q4 + 4 * half) =
                make_float4(acc[i][q4 * 4 + 0], acc[i][q4 * 4 + 1], acc[i][q4 * 4 + 2], acc[i][q4 * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = (lane >> 3) + 8 * it, cg = lane & 7;
            const float4 v = *(const float4*)(T + row * QX_TR_STRIDE + 4 * cg);
            // piece s starts at out + s * out_stride (float4 units; the same size as a packed piece), row m at m * 3 D floats
            st_out(a.out + (long long)s * a.out_stride + ((long long)(32 * i + row) * (a.NT * 32) + nt * 32 + 4 * cg) / 4, v);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    WMAR_ST_END(a.trace, j)
    if (keeper) __syncthreads();
}

static int launch_qkvx_bx(const QkvxArgs& q, int S_in, hipStream_t st) {
    const dim3 grid((unsigned)(8 * q.cap));
    switch (S_in) {
#define WMAR_QX_CASE(N) case N: hipLaunchKernelGGL((k_qkvx_bx<N>), grid, dim3(512), 0, st, q); break;
        WMAR_QX_CASE(0) WMAR_QX_CASE(1) WMAR_QX_CASE(2) WMAR_QX_CASE(3) WMAR_QX_CASE(4)
        WMAR_QX_CASE(5) WMAR_QX_CASE(6) WMAR_QX_CASE(7) WMAR_QX_CASE(8)
#undef WMAR_QX_CASE
        default: set_error("qkvx_bx: bad slab count %d", S_in); return WMAR_EINVAL;
    }
    return launch_status("k_qkvx_bx");
}


// ------------------------------------------------- plain split-K GEMM on the bf16 matrix pipe
// out[32 MTW x N] = X W^T with X given as bf16 pieces (planes [K/16][MTW][3][64], written by the producer: bx_store_planes4) and W
// in fp32 (k_pack_bx order), split in registers.  Workgroup = NT column tiles of 32 x one K slice of 64 PER; its four waves take
// 16 PER k each for all rows, and meet in LDS in a fixed order.  Slab s = raw partial sums of slice s in the packed fp32 layout --
// what k_resid_stats and the attention prologue consume -- or, with the whole K in one slice, gelu(sum + bias) as bf16 pieces for the
// next k_bx.  Users: the Taming output projection (64 rows, 32 x 384 tiles), RAR's QKV / proj / FC1 / FC2 at 128 rows.
// A wave keeps ONE step of weights and of activation pieces in flight beyond the one it computes on: more requests per CU than that
// queue at the memory pipeline and block the wave's in-order issue, MFMAs included (scripts/stream_profile.hip, scripts/bx6_bench.hip).
struct BxArgs {
    const float4* Wq;          // k_pack_bx layout
    const u32x4* Xq;           // activation planes [K/16][MTW][3][64]
    float4* out;               // slab s at out + s * slab_stride, packed [N/8][MTW][64]
    long long slab_stride;     // float4 units
    int KU;                    // K / 16
    int S;                     // K slices: KU == 4 * PER * S
    // GELU variant (S == 1): out is not written; gelu(sum + bias) goes out as bf16 pieces for the next k_bx GEMM
    const float* bias;         // [N]; without GELU: nullable, added to the slab (S == 1: finished values)
    u32x4* outq;               // planes [N/16][MTW][3][64]
    int rowmajor;              // (NW == 4, no GELU) slab s as a ROW-MAJOR piece [32 MTW rows][N] instead of the packed layout: what the
    int N;                     // attention prologue reads per (sequence, head) is then contiguous (k_qkvx_bx's epilogue, round 5)
};

#define WMAR_BX_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)

// NW = waves per workgroup (4; 8 for RAR's FC1, round 5): the waves split the K slice, NW x PER x 16 k per workgroup.  A wave keeps one
// step in flight beyond the one it computes, so a launch whose steps are bound by the latency of their loads takes PER x that latency:
// twice the waves, half the steps.
// XF (round 5): the activation arrives as PACKED FP32 rows [K/8][MTW][64] instead of bf16 planes and is split in registers like the
// weights -- 2 x 16 bytes per lane, row tile and step instead of 3 x 16 (a launch whose steps are bound by the bytes its CU pulls: RAR's
// FC1, 983 KB of planes per workgroup); lane (row, k half) of the B operand fetches both halves of k-block 2 U + k half.  The pieces are
// the ones bx_store_planes4 would have stored (same bx_split2 per pair): bit-identical results.
template <int NT, int PER, int MTW = 2, bool GELU = false, int NW = 4, bool XF = false>
__global__ __launch_bounds__(NW * 64) void k_bx(BxArgs a) {
    // MTW row tiles of 32 (2: 64 rows, 4: 128 rows -- RAR under guidance); planes [K/16][MTW][3][64], slabs packed [N/8][MTW][64]
    constexpr int XR = XF ? 2 * MTW : 3 * MTW;             // 16-byte operand loads of a step
    constexpr int ROWS = NT * 4 * MTW;      // float4 rows (tile, row tile, register group) of the workgroup's output
    __shared__ __attribute__((aligned(16))) float4 red[NW][ROWS][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // slice = id % S: with S a multiple of 4 every XCD (id % 8) works on one or two K slices and keeps only those in its L2
    const int grp = (int)blockIdx.x / a.S, ks = (int)blockIdx.x % a.S;
    const int u0 = (ks * NW + w) * PER;
    f32x16 acc[NT][MTW];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
    const float4* wp = a.Wq + ((long long)grp * NT * a.KU + u0) * 128 + lane;
    const long long wt = (long long)a.KU * 128;     // next column tile
    const u32x4* xp = XF ? a.Xq + ((long long)(2 * u0 + (lane >> 5)) * MTW) * 64 + (lane & 31)      // k-block 2 U + lane / 32, row lane % 32
                         : a.Xq + (long long)u0 * XR * 64 + lane;
    float4 wr[NT][2];
    u32x4 xr[XR], xr2[XF ? XR : 1];      // XF: the activation runs TWO steps ahead (even steps in xr, odd steps in xr2)
#define WMAR_BX_LOADW(U)                                                                           \
    { _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                             \
        wr[t][0] = ld_nt(wp + t * wt + (long long)(U) * 128);                                      \
        wr[t][1] = ld_nt(wp + t * wt + (long long)(U) * 128 + 64); } }
#define WMAR_BX_LOADXF(XB, U)                                                                      \
    { _Pragma("unroll") for (int i = 0; i < MTW; ++i) {                                            \
        XB[2 * i] = xp[((long long)(2 * (U)) * MTW + i) * 64];                                     \
        XB[2 * i + 1] = xp[((long long)(2 * (U)) * MTW + i) * 64 + 32]; } }
#define WMAR_BX_LOADX(U)                                                                           \
    { _Pragma("unroll") for (int q = 0; q < XR; ++q) xr[q] = xp[(long long)(U) * (XR * 64) + q * 64]; }
    WMAR_BX_LOADW(0);
    if (XF) { WMAR_BX_LOADXF(xr, 0) if (1 < PER) WMAR_BX_LOADXF(xr2, 1) } else WMAR_BX_LOADX(0);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 ph[2][NT], pm[2][NT], pl[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bx_split8(wr[t][0], wr[t][1], ph[0][t], pm[0][t], pl[0][t]);
    if (1 < PER) WMAR_BX_LOADW(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = j & 1, n = c ^ 1;
        // the NEXT step's weights are split (VALU) between this step's MFMAs; their registers are refilled at once
        if (j + 1 < PER) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bx_split8(wr[t][0], wr[t][1], ph[n][t], pm[n][t], pl[n][t]);
            if (j + 2 < PER) WMAR_BX_LOADW(j + 2);
        }
        bf16x8 x[3 * MTW];
        if (XF) {
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const u32x4 lo = c ? xr2[(2 * i) % (XF ? XR : 1)] : xr[2 * i], hi = c ? xr2[(2 * i + 1) % (XF ? XR : 1)] : xr[2 * i + 1];
                bx_split8(__builtin_bit_cast(float4, lo), __builtin_bit_cast(float4, hi), x[3 * i], x[3 * i + 1], x[3 * i + 2]);
            }
            // the buffer just split is free: step j + 2 is requested before this step's MFMAs
            if (j + 2 < PER) { if (c) WMAR_BX_LOADXF(xr2, j + 2) else WMAR_BX_LOADXF(xr, j + 2) }
        } else {
#pragma unroll
            for (int q = 0; q < 3 * MTW; ++q) x[q] = __builtin_bit_cast(bf16x8, xr[XF ? 0 : q]);
        }
        // x[3 mt + piece]; the small products first
#define WMAR_BX_ROUND(WP, XP)                                                                      \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                             \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i) WMAR_BX_MFMA(WP[c][t], x[3 * i + XP], acc[t][i]);
        WMAR_BX_ROUND(pl, 0) WMAR_BX_ROUND(ph, 2) WMAR_BX_ROUND(pm, 1) WMAR_BX_ROUND(pm, 0) WMAR_BX_ROUND(ph, 1) WMAR_BX_ROUND(ph, 0)
#undef WMAR_BX_ROUND
        if (!XF && j + 1 < PER) WMAR_BX_LOADX(j + 1);
#pragma unroll
        for (int i = 0; i < 6 * MTW * NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            if (i % 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef WMAR_BX_LOADW
#undef WMAR_BX_LOADX
#undef WMAR_BX_LOADXF
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                red[w][(t * MTW + i) * 4 + g][lane] = make_float4(acc[t][i][4 * g], acc[t][i][4 * g + 1], acc[t][i][4 * g + 2], acc[t][i][4 * g + 3]);
    __syncthreads();
    // wave w sums rows w, w + 4, ... of the (tile, row tile, register group) rows over the four K quarters, in fixed order
    float4* out = a.out + (long long)ks * a.slab_stride;
    if (!GELU && NW == 4 && a.rowmajor) {
        // Row-major piece: with four waves, wave w holds register group g = w of every (tile, row tile) pair -- columns 8 w + 4 (lane / 32)
        // .. + 3 of the 32-column tile, row lane % 32.  The pairs are turned through LDS (the reduction buffer, free behind a barrier):
        // a store instruction then covers eight rows x 128 contiguous bytes.
        constexpr int NP = NT * MTW, TS = 36;            // pairs; floats per row of a transposed tile (32 + 4: 16-byte rows on distinct banks)
        static_assert(NP * 32 * TS * 4 <= (int)sizeof(red), "k_bx: transposed tiles do not fit the reduction buffer");
        float4 vv[NP];
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const int row = r * NW + w;
            float4 v = red[0][row][lane];
#pragma unroll
            for (int o = 1; o < NW; ++o) { const float4 q = red[o][row][lane]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            if (a.bias) {
                const float4 bb = *(const float4*)(a.bias + ((grp * NT + r / MTW) * 4 + w) * 8 + 4 * (lane >> 5));
                v = make_float4(v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w);
            }
            vv[r] = v;
        }
        __syncthreads();                                 // every wave has read its rows of `red`
        float* T = reinterpret_cast<float*>(&red[0][0][0]);
#pragma unroll
        for (int r = 0; r < NP; ++r) *(float4*)(T + (r * 32 + (lane & 31)) * TS + 8 * w + 4 * (lane >> 5)) = vv[r];
        __syncthreads();
#pragma unroll
        for (int r = w; r < NP; r += NW) {
            const int t = r / MTW, i = r % MTW;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = (lane >> 3) + 8 * it, cg = lane & 7;
                const float4 v = *(const float4*)(T + (r * 32 + row) * TS + 4 * cg);
                st_out(out + ((long long)(32 * i + row) * a.N + (grp * NT + t) * 32 + 4 * cg) / 4, v);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < (ROWS + NW - 1) / NW; ++r) {
        const int row = r * NW + w;
        if (ROWS % NW != 0 && row >= ROWS) break;
        const int t = row / (4 * MTW), i = (row >> 2) % MTW, g = row & 3;
        float4 v = red[0][row][lane];
#pragma unroll
        for (int o = 1; o < NW; ++o) { const float4 q = red[o][row][lane]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        if (GELU) {
            const int kb = (grp * NT + t) * 4 + g, hf = lane >> 5;          // output features 8 kb + 4 hf .. + 3 = the next GEMM's k
            const float4 bb = *(const float4*)(a.bias + kb * 8 + 4 * hf);
            v = make_float4(gelu_erf(v.x + bb.x), gelu_erf(v.y + bb.y), gelu_erf(v.z + bb.z), gelu_erf(v.w + bb.w));
            bx_store_planes4(a.outq, MTW, kb, hf, i, lane & 31, v);
        } else {
            if (a.bias) {       // (round 5: RAR's adaLN GEMM -- the whole K in one slice, finished values)
                const float4 bb = *(const float4*)(a.bias + ((grp * NT + t) * 4 + g) * 8 + 4 * (lane >> 5));
                v = make_float4(v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w);
            }
            st_out(out + ((long long)((grp * NT + t) * 4 + g) * MTW + i) * 64 + lane, v);
        }
    }
}

constexpr int BX_PER = 6;              // 16-k steps per wave: K slice = 4 waves x 6 x 16 = 384
constexpr int BX_KSLICE = 4 * BX_PER * 16;
// N must be a multiple of 32 * NT, K = S * 64 * PER; 32 * MTW rows
template <int NT, int PER = BX_PER, int MTW = 2, bool GELU = false, int NW = 4, bool XF = false>
static int launch_bx(const BxArgs& a, int N, hipStream_t st) {
    hipLaunchKernelGGL((k_bx<NT, PER, MTW, GELU, NW, XF>), dim3((unsigned)(N / (32 * NT) * a.S)), dim3(NW * 64), 0, st, a);
    return launch_status("k_bx");
}



// ------------------------------------------------- output projection + residual fold + LN2 statistics in ONE launch (round 4)
// k_bx<1, PER> for the attention output projection, with what k_resid_stats did as a launch of its own behind it (5 us x 48 layers)
// moved INSIDE the launch behind an XCD-local barrier.
//   * Block b runs on an XCD that depends on b % 8 only (measured on gfx950: XCC id = (b + r) % 8, r rotating with the launches
//     before it; 24 of the 192 blocks on each XCD every time).  XCD group x = b % 8 owns column tiles 6x .. 6x+5 for ALL K slices: the
//     S partial tiles of a column tile are then written and read through ONE L2.
//   * phase 1 = k_bx: workgroup (tile, K slice) writes its partial tile to its slab with plain stores; s_waitcnt vmcnt(0) = the L2
//     has them.
//   * XCD-local barrier (scripts/xg_kernel.h, profiles/r04_xg_barrier_trace.log: 1.8 us): arrival counter and generation word are L2
//     atomics WITHOUT sc1 -- they never leave the XCD -- polled with a returning L2 atomic.  The device-wide barrier of round 1 cost 17 us.
//   * phase 2: the first four workgroups of the group (row tile x half of the group's 24 k-blocks) fold: x += bias + slab 0 + .. +
//     slab S-1 in slab order (the arithmetic and order of k_resid_stats), and leave one (sum, sum of squares) per row and half for
//     LayerNorm 2: statistics chunk 2 x + half of 16.  Loads bypass the L1 (nt): the slabs were written by other CUs of this XCD.
// Safety: the engine probes the block -> XCD grouping at creation and otherwise keeps the two-launch path; in the launch, block c == 0
// of a group publishes its XCC id and every block compares after the barrier (fail[0]); a wait that does not complete in 2^22 polls
// raises fail[1] and goes on (wmar_gpt_check / the next call report it: the results are then invalid).  All 192 workgroups are
// co-resident (one per CU).  Results never depend on timing: every sum has a fixed order.
struct BxrArgs {
    BxArgs bx;
    float4* x;                 // packed residual stream [KB][2][64], updated in place
    const float* bias;         // [N]
    double* stats;             // [16][64][2]: chunk = 2 x XCD group + half
    unsigned* sync;            // [8][64] words: word 0 arrivals, word 16 the group's XCC id, word 32 generation
    unsigned* fail;            // [0] placement mismatch, [1] barrier timeout
    int tiles_per_group;       // column tiles per XCD group (N / 32 / 8)
    unsigned long long* trace; // dev only (WMAR_STAMPS): 8 stamps per workgroup
};

// the current value of a word as THIS XCD's L2 holds it (a compiler-level fetch_or(0) folds into a load that may hit the L1)
__device__ __forceinline__ unsigned l2_read_u32(unsigned* p) {
    unsigned v; const unsigned z = 0;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(z) : "memory");
    return v;
}
// returning add / swap performed by THIS XCD's L2 (sc0 = return the old value; no sc1: the request stops at the L2)
__device__ __forceinline__ unsigned l2_add_u32(unsigned* p, unsigned v) {
    unsigned o;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(o) : "v"(p), "v"(v) : "memory");
    return o;
}
// fire-and-forget forms (no return value, nothing to wait for)
__device__ __forceinline__ void l2_add_u32_noret(unsigned* p, unsigned v) {
    asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void l2_swap_u32_noret(unsigned* p, unsigned v) {
    asm volatile("global_atomic_swap %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
constexpr int XR_MAX_POLLS = 1 << 15;      // ~20 ms of polling against a 1.8 us barrier
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
static __global__ void k_xcc_probe(unsigned* o) { if (threadIdx.x == 0) o[blockIdx.x] = xcc_id(); }

// NW waves split the K slice in phase 1 (NW x PER x 16 k); phase 2 is the work of waves 0..3
template <int PER, int S, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_bx_xr(BxrArgs q) {
    constexpr int MTW = 2, XR = 3 * MTW, ROWS = 4 * MTW;
    const BxArgs& a = q.bx;
    __shared__ __attribute__((aligned(16))) float4 red[NW][ROWS][64];
    __shared__ double sred[4][64][2];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = (int)blockIdx.x & 7, c = (int)blockIdx.x >> 3;          // XCD group, member
    const int tile = grp * q.tiles_per_group + c / S, ks = c % S;
    unsigned gen0 = 0, xid = 0;
    WMAR_ST_BEGIN
    if (threadIdx.x == 0) {
        xid = xcc_id();
        if (c == 0) l2_swap_u32_noret(q.sync + grp * 64 + 16, xid);
        gen0 = __hip_atomic_load(q.sync + grp * 64 + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // global_load sc1: served by the L2
    }
    // ---------------------------------------------------------------------------------------------------- phase 1 (k_bx<1, PER>)
    {
        const int u0 = (ks * NW + w) * PER;
        f32x16 acc[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float4* wp = a.Wq + ((long long)tile * a.KU + u0) * 128 + lane;
        const u32x4* xp = a.Xq + (long long)u0 * XR * 64 + lane;
        float4 wr[2];
        u32x4 xr[XR];
#define WMAR_BXR_LOADW(U) { wr[0] = ld_nt(wp + (long long)(U) * 128); wr[1] = ld_nt(wp + (long long)(U) * 128 + 64); }
#define WMAR_BXR_LOADX(U) { _Pragma("unroll") for (int e = 0; e < XR; ++e) xr[e] = xp[(long long)(U) * (XR * 64) + e * 64]; }
        WMAR_BXR_LOADW(0);
        WMAR_BXR_LOADX(0);
        __builtin_amdgcn_sched_barrier(0);
        WMAR_ST_LANDED(2)
        bf16x8 ph[2], pm[2], pl[2];
        bx_split8(wr[0], wr[1], ph[0], pm[0], pl[0]);
        if (1 < PER) WMAR_BXR_LOADW(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int cu = j & 1, n = cu ^ 1;
            if (j + 1 < PER) {
                bx_split8(wr[0], wr[1], ph[n], pm[n], pl[n]);
                if (j + 2 < PER) WMAR_BXR_LOADW(j + 2);
            }
            bf16x8 x[XR];
#pragma unroll
            for (int e = 0; e < XR; ++e) x[e] = __builtin_bit_cast(bf16x8, xr[e]);
#define WMAR_BXR_ROUND(WP, XP) _Pragma("unroll") for (int i = 0; i < MTW; ++i) WMAR_BX_MFMA(WP[cu], x[3 * i + XP], acc[i]);
            WMAR_BXR_ROUND(pl, 0) WMAR_BXR_ROUND(ph, 2) WMAR_BXR_ROUND(pm, 1) WMAR_BXR_ROUND(pm, 0) WMAR_BXR_ROUND(ph, 1) WMAR_BXR_ROUND(ph, 0)
#undef WMAR_BXR_ROUND
            if (j + 1 < PER) WMAR_BXR_LOADX(j + 1);
#pragma unroll
            for (int i = 0; i < 6 * MTW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                if (i % 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WMAR_BXR_LOADW
#undef WMAR_BXR_LOADX
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                red[w][i * 4 + g][lane] = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        __syncthreads();
        float4* out = a.out + (long long)ks * a.slab_stride;
#pragma unroll
        for (int r = 0; r < ROWS / NW; ++r) {
            const int row = r * NW + w, i = row >> 2, g = row & 3;
            float4 v = red[0][row][lane];
#pragma unroll
            for (int o = 1; o < NW; ++o) { const float4 z = red[o][row][lane]; v.x += z.x; v.y += z.y; v.z += z.z; v.w += z.w; }
            out[((long long)(tile * 4 + g) * MTW + i) * 64 + lane] = v;
        }
    }
    // ---------------------------------------------------------------------------------------------------- XCD-local barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's slab stores are in the L2
    WMAR_ST(3)
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* cnt = q.sync + grp * 64;
        unsigned* gen = cnt + 32;
        const unsigned members = (unsigned)(q.tiles_per_group * S);
        // The arrive / reset / release atomics are spelled out (no sc1: performed by THIS XCD's L2, they never leave the XCD), so the
        // cache policy does not depend on how the compiler lowers a memory scope.
        const unsigned old = l2_add_u32(cnt, 1u);
        if (old == members - 1u) {
            l2_swap_u32_noret(cnt, 0u);
            l2_add_u32_noret(gen, 1u);
        } else {
            // Bounded wait: the barrier takes 1.8 us on a healthy device (polls of ~0.5 us).  A launch whose workgroups are not all
            // resident (a CU mask, a partition mode, another process on the device) would otherwise spin for seconds in each of the
            // 48 x 256 launches of a captured loop: the first wait that gives up raises fail[1], every later wait -- in this launch
            // and in the launches behind it -- sees the flag and leaves at once, and the host (gpt_verify) re-runs the call on the
            // two-launch path.
            int spins = 0;
            bool dead = __hip_atomic_load(q.fail + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            while (!dead && l2_read_u32(gen) == gen0) {
                __builtin_amdgcn_s_sleep(2);
                ++spins;
                if ((spins & 255) == 0) dead = __hip_atomic_load(q.fail + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (spins > XR_MAX_POLLS) { __hip_atomic_store(q.fail + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        if (l2_read_u32(q.sync + grp * 64 + 16) != xid) __hip_atomic_store(q.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    WMAR_ST(6)
    // ---------------------------------------------------------------------------------------------------- phase 2
    if (c >= 2 * MTW) { WMAR_ST_END(q.trace, blockIdx.x) return; }
    {
        // four workgroups fold: row tile mt = c & 1, half hp = c >> 1 of the group's k-blocks (12 of 24 at n_embd 1536: three per
        // wave, ONE batch of 3 x (1 + S) loads) -> statistics chunk 2 grp + hp of 16
        const int mt = c & 1, hp = c >> 1;
        const int nkb = q.tiles_per_group * 2;                   // k-blocks (8 columns) of this half
        const int kb_first = (grp * 2 + hp) * nkb;
        const int half = lane >> 5;
        double s = 0.0, ss = 0.0;
        for (int k0 = w; k0 < nkb && w < 4; k0 += 12) {
            float4 v[3], sl[3][S], bb[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int kb = kb_first + min(k0 + 4 * e, nkb - 4 + w);      // clamped (re-read; masked below)
                const long long idx = ((long long)kb * MTW + mt) * 64 + lane;
                v[e] = q.x[idx];
                bb[e] = *(const float4*)(q.bias + kb * 8 + 4 * half);
#pragma unroll
                for (int si = 0; si < S; ++si) sl[e][si] = ld_nt(a.out + (long long)si * a.slab_stride + idx);
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                if (k0 + 4 * e >= nkb) continue;
                const int kb = kb_first + k0 + 4 * e;
                float4 t = sl[e][0];
#pragma unroll
                for (int si = 1; si < S; ++si) { t.x += sl[e][si].x; t.y += sl[e][si].y; t.z += sl[e][si].z; t.w += sl[e][si].w; }
                // k_resid_stats: x + gate * (bias + sum), gate == 1
                const float4 r = make_float4(v[e].x + (bb[e].x + t.x), v[e].y + (bb[e].y + t.y), v[e].z + (bb[e].z + t.z), v[e].w + (bb[e].w + t.w));
                q.x[((long long)kb * MTW + mt) * 64 + lane] = r;
                s += (double)r.x + (double)r.y + (double)r.z + (double)r.w;
                ss += sq4_f64(r);         // never a v_fmac_f64 chain: common.h
            }
        }
        if (w < 4) { sred[w][lane][0] = s; sred[w][lane][1] = ss; }
        __syncthreads();
        if (threadIdx.x < 32) {
            double ts = 0, tss = 0;
            for (int i = 0; i < 4; ++i) { ts += sred[i][threadIdx.x][0] + sred[i][threadIdx.x + 32][0]; tss += sred[i][threadIdx.x][1] + sred[i][threadIdx.x + 32][1]; }
            double* o = q.stats + ((long long)(grp * 2 + hp) * (MTW * 32) + mt * 32 + threadIdx.x) * 2;
            o[0] = ts; o[1] = tss;
        }
    }
    WMAR_ST_END(q.trace, blockIdx.x)
}

template <int PER, int S, int NW = 4>
static int launch_bx_xr(const BxrArgs& q, int N, hipStream_t st) {
    hipLaunchKernelGGL((k_bx_xr<PER, S, NW>), dim3((unsigned)(N / 32 * S)), dim3(NW * 64), 0, st, q);
    return launch_status("k_bx_xr");
}

// --------------------------------------------------------------------- decode attention
// K/V rows are streamed once per step and the cache (9.7 GB at batch 64) is far larger than the L2s and the memory-side cache: they
// are loaded NON-TEMPORALLY -- round 4, same-box A/B over the 256-step loop: 4.112 -> 3.98 ms per step (Taming), 3.948 -> 3.899 (RAR-XL):
// with plain loads 100 MB of K/V per layer swept the slabs, pieces and activations of the neighbouring launches out of the L2s.
// -DWMAR_ATT_PLAIN_KV restores plain loads.
// 1-KiB loads of K (and of V) per chunk of k_attn_decode.  Round 5: 2 (8 cached rows of head_dim 64 per chunk, two chunks = 8 KiB in
// flight per wave) instead of 8.  With 32 KiB in flight per wave the 1536 waves of a 64-row launch had 50 MB queued at the memory
// system -- four times what 6 TB/s x its latency can use -- so every dependent round of a wave waited ~8 us, some waves far longer
// (in-loop stamps: streaming phase 9.6 us mean / 14.8 max, last workgroup out 4 us after the mean).  Same-box round-robin A/B over
// the 256-step loop (scripts/ab_loop.py, 12 runs each): CH 8 / 4 / 3 / 2 / 1 = 3.894 / 3.851 / 3.840 / 3.798 / 3.829 ms per step.
#ifndef WMAR_ATT_CH
#define WMAR_ATT_CH 2
#endif
#ifndef WMAR_A80_NU
#define WMAR_A80_NU 2           // k_attn_decode80: 1-KiB main loads of K (and of V) per chunk, 4 cached rows each (round 5: 2 instead of
                                // 4 -- RAR-XL 4.04 -> 3.98 ms per step, same box; 1: 3.97)
#endif
// Order of the first requests of a one-wave workgroup (LATE in k_attn_decode): 0 = first chunk requested together with the prologue's
// operands (round 4), 1 = both chunks once the operands are here, 2 = first chunk once they are here, the second behind the q / k / v
// algebra.  At CH 2 the three are within noise (3.798 / 3.814 / 3.798); at CH 8: 3.915 / 3.893 / 3.894.
#ifndef WMAR_ATT_LATE_KV
#define WMAR_ATT_LATE_KV 2
#endif
#ifndef WMAR_ATT_PLAIN_KV
#define WMAR_KV_LD(P) ld_nt(P)
#else
#define WMAR_KV_LD(P) (*(P))
#endif
// One wave per (sequence, head).  K/V rows are hd floats; LPR = hd/4 lanes cover a row with
// float4s and RPI = 64/LPR rows are read per wave-wide load (1 KiB, coalesced).
struct AttnArgs {
    // The QKV projection arrives as S split-K partial slabs in packed layout (columns
    // [q | k | v], 3*D wide).  The attention wave of (sequence, head) finishes its own
    // 3 x hd columns -- LayerNorm algebra, bias -- appends k and v to the cache and goes on.
    const float4* qkv_slabs;   // [S][(3D/8)][MT][64]
    long long slab_stride;     // float4 units
    int S;
    const double* stats; int n_chunks; int K;   // LN1 row statistics (see k_resid_stats)
    double invK;               // 1 / K (host-computed: a double division costs the prologue ~100 cycles)
    const float* c1;           // [3D] row sums of the gamma-folded QKV weights (mode 0)
    const float* bias;         // [3D] bias (+ W beta in mode 0)
    int rowmajor;              // the pieces are row-major [rows][3 D] (written by k_qkvx_bx) instead of the packed operand layout
    int mode;                  // 0: minGPT (LN1 folded, finish with LN algebra); 1: RAR (plain bias, then per-head
                               //    LayerNorm of q and k -- Attention.q_norm / k_norm, rar.py:76-94)
    const float *qn_w, *qn_b, *kn_w, *kn_b;   // [hd] (mode 1)
    float* kcache;             // [B][H][Tmax][hd] (this layer)
    float* vcache;
    float4* y;                 // packed [KB][MT][64]
    u32x4* yq;                 // nullable: the same rows as bf16 pieces for a k_bx output projection (then y is not written)
    const int* pos_dev;
    int D, H, Tmax, MT;
    float scale;
    int dbg;                   // dev builds only (-DWMAR_DEV_KNOBS, WMAR_ATT_DBG): 2 = skip K/V streaming
    unsigned long long* trace; // dev only (WMAR_ATT_TRACE): 5 timestamps per workgroup
};

// One workgroup of NWA waves per (sequence, head).  The cached rows are cut into chunks of
// CH 1-KiB loads (CH*RPI rows); wave w takes chunks w, w+NWA, ...  and keeps a running
// (max, sum, weighted V sum) in registers -- K and V of a chunk are requested together, the next
// chunk is in flight while the current one is reduced.  The NWA partial results meet in LDS.
// MODE (AttnArgs::mode) and RM (AttnArgs::rowmajor) are template parameters (round 5): as run-time branches around the prologue's
// loads they left merge points behind which hipcc's s_waitcnt pass waited for EVERY outstanding load (vmcnt(0)) -- the cache chunks
// requested in front of the q / k / v algebra had to land before it could even begin.
template <int HD, int NWA, bool PF2 = false, int MODE = 0, bool RM = false>
__global__ __launch_bounds__(NWA * 64) void k_attn_decode(AttnArgs a) {
    // LPRA lanes (one float4 each) cover a cache row; rows are laid on LPR = next power of two lanes so
    // that the row reductions are xor-shuffles (hd = 80: 20 of 32 lanes active, 2 rows per load).
    constexpr int LPRA = HD / 4;
    constexpr int LPR = LPRA <= 8 ? 8 : (LPRA <= 16 ? 16 : 32);
    constexpr int RPI = 64 / LPR;
    constexpr int CH = WMAR_ATT_CH;
    constexpr int ROWS = CH * RPI;
    static_assert(HD % 4 == 0 && LPRA <= 32, "head_dim must be a multiple of 4, at most 128");
    __shared__ __attribute__((aligned(16))) float part[NWA][HD + 4];    // per wave: weighted V sum [HD], max, sum (16-B rows)
    __shared__ __attribute__((aligned(16))) float qkv_s[3][HD];
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int lane = threadIdx.x & 63;
    const int w = NWA == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = *a.pos_dev + 1;
    const int subr = lane % LPR, rsel = lane / LPR;
    const bool lane_on = subr < LPRA;            // idle lanes of a padded row read lane 0's data and are masked
    const int sub = lane_on ? subr : 0;
    float* Kc = a.kcache + ((long long)b * a.H + h) * a.Tmax * HD + sub * 4;
    float* Vc = a.vcache + ((long long)b * a.H + h) * a.Tmax * HD + sub * 4;
#ifdef WMAR_DEV_KNOBS
    const int nchunk = (a.dbg & 2) ? 0 : (T + ROWS - 1) / ROWS;
#else
    const int nchunk = (T + ROWS - 1) / ROWS;
#endif

    // rows past T-1 are clamped to T-1 and replaced from registers / masked below
#define WMAR_ATT_LOAD(KB, VB, C0)                                                        \
    _Pragma("unroll") for (int u = 0; u < CH; ++u) {                                     \
        const int t = min((C0) * ROWS + u * RPI + rsel, T - 1);                          \
        KB[u] = WMAR_KV_LD((const float4*)(Kc + (long long)t * HD));                     \
        VB[u] = WMAR_KV_LD((const float4*)(Vc + (long long)t * HD));                     \
    }
    float4 kA[CH], vA[CH], kB[CH], vB[CH];
    float4 q, knew, vnew;
    WMAR_ST_BEGIN
#ifdef WMAR_ATT_TRACE
    unsigned long long tr[5];
    tr[0] = __builtin_amdgcn_s_memtime();
#endif
    // PF2: the wave's first TWO chunks are requested before the prologue (32 KiB in flight per wave: with 1 / 2 / 4 waves the
    // whole cache up to 64 / 128 / 256 rows streams while q/k/v are finished); otherwise one, the second from inside the loop.
    // (Measured dead end: clamping the first chunk to the cache's capacity instead of its fill, so that its loads need not wait
    // for the scalar load of the position, saves 1.3 us at 64 rows and costs 3 us below 32 rows -- stale rows are then streamed.)
    // The prologue's loads go FIRST: loads return in issue order, so behind 16 KiB of cache rows per wave the QKV pieces would
    // arrive only after the chip-wide burst of first chunks has drained (~4 us); ahead of it they are back in ~1.5 us and q, k, v
    // are finished while the first chunk is still in flight.
    // every load of the prologue is issued before the first wait: LN partial sums (one chunk per
    // lane), the QKV slab(s), the folded-LN row sums and the bias (16 lanes each)
    const int Mpad = a.MT * 32;
    const int mt = b >> 5;
    double sm = 0, sq = 0;
    // Every prologue load is UNCONDITIONAL per lane (indices clamped, unused values masked where they are summed): a load under a
    // per-lane condition (`cond ? load : 0`) compiles to an exec-masked branch whose result is merged right behind it, i.e. one
    // `s_waitcnt vmcnt(0)` per load -- a dozen serialized L2 round trips instead of one (round 3: the ISA showed exactly that).
    double2 st0 = make_double2(0.0, 0.0);
    if (w == 0) st0 = *(const double2*)(a.stats + ((long long)min(lane, a.n_chunks - 1) * Mpad + b) * 2);
    // The S split-K pieces of this head's 3 x hd columns are spread over the wave's RPI row groups (piece p is fetched by
    // group p % RPI), so that a lane holds at most PMAX pieces: all loads are still in flight together, without 3 x 8
    // float4 registers per lane (the kernel's occupancy is set by its registers).  The partial sums meet in a fixed
    // xor-butterfly over the groups: the summation order depends on S only.
    constexpr int PMAX = (QKV_SLABS_MAX + RPI - 1) / RPI;
    float4 sl[3][PMAX], cc[3], bb[3];
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const int n = which * a.D + h * HD + sub * 4;          // first of this lane's 4 columns
        // packed operand layout, or row-major pieces [rows][3 D] (k_qkvx_bx, round 5): this lane's float4 of row b
        const long long idx = RM ? ((long long)b * (3 * a.D) + n) >> 2
                                         : ((long long)(n >> 3) * a.MT + mt) * 64 + (b & 31) + 32 * ((n >> 2) & 1);
        if (w == 0) {           // wave-uniform
#pragma unroll
            for (int pi = 0; pi < PMAX; ++pi) {
                const int pc = min(rsel + pi * RPI, a.S - 1);
                sl[which][pi] = a.qkv_slabs[(long long)pc * a.slab_stride + idx];
            }
            cc[which] = *(const float4*)((MODE == 0 ? a.c1 : a.bias) + n);     // (mode 1 has no c1: any valid address, value unused)
            bb[which] = *(const float4*)(a.bias + n);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // LATE (round 5, one wave per (sequence, head)): the cache chunks are requested only once the prologue's operands are HERE.
    // In-loop stamps (profiles/r05_stamp_table_*): with the first chunk requested up front, the six waves of a CU queued 96 KiB of K/V
    // in front of each other's 5 KiB of prologue operands -- they landed 4.2 us after entry on average, 12.3 at worst, and the
    // late waves were the launch's stragglers (last workgroup out 4.2 us after the mean).  Now every wave gets its operands at
    // L2 latency, requests TWO chunks, and finishes q / k / v while they fly.
    constexpr bool LATE = WMAR_ATT_LATE_KV && NWA == 1;
    constexpr bool LATE2 = LATE && WMAR_ATT_LATE_KV == 1;      // both chunks in front of the q / k / v algebra (2: the second one behind it)
    if (LATE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef WMAR_STAMPS
        st_[2] = __builtin_amdgcn_s_memtime();
#endif
    }
    WMAR_ATT_LOAD(kA, vA, w)                    // unconditional (clamped rows): see the refills below
    if (PF2 || LATE2) { WMAR_ATT_LOAD(kB, vB, w + NWA) }
    __builtin_amdgcn_sched_barrier(0);
    if (w == 0) {
#ifdef WMAR_STAMPS
        if (!LATE) {
            // (the prologue's operands only: the cache chunk behind them stays in flight -- 16 K/V loads were issued after them)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            st_[2] = __builtin_amdgcn_s_memtime();
        }
#endif
#ifdef WMAR_ATT_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[1] = __builtin_amdgcn_s_memtime();
#endif
        sm = lane < a.n_chunks ? st0.x : 0.0; sq = lane < a.n_chunks ? st0.y : 0.0;
        // (n_chunks <= STAT_CHUNKS_MAX = 64: one chunk per lane; a loop over further chunks here would put a load in a loop and make
        // hipcc wait for ALL outstanding loads, the first cache chunk included)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
        const double invK = a.invK;
        const double mean = sm * invK;
        const float mu = (float)mean;
        const float rstd = rsqrtf((float)var_f64(sq * invK, mean) + 1e-5f);
        float4 accs[3];
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int pi = 0; pi < PMAX; ++pi)
                if (rsel + pi * RPI < a.S) {
                    acc.x += sl[which][pi].x; acc.y += sl[which][pi].y;
                    acc.z += sl[which][pi].z; acc.w += sl[which][pi].w;
                }
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
                acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
            }
            accs[which] = acc;
        }
        if (rsel == 0) {
            float4 r[3];
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                const float4 acc = accs[which];
                if (MODE == 0) {
                    r[which] = make_float4(rstd * (acc.x - mu * cc[which].x) + bb[which].x,
                                           rstd * (acc.y - mu * cc[which].y) + bb[which].y,
                                           rstd * (acc.z - mu * cc[which].z) + bb[which].z,
                                           rstd * (acc.w - mu * cc[which].w) + bb[which].w);
                } else {
                    r[which] = make_float4(acc.x + bb[which].x, acc.y + bb[which].y, acc.z + bb[which].z, acc.w + bb[which].w);
                }
            }
            if (MODE == 1) {
                // q_norm / k_norm: LayerNorm over the hd values of this head (eps 1e-6, affine)
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    float4 v4 = lane_on ? r[which] : make_float4(0.f, 0.f, 0.f, 0.f);
                    float sm1 = v4.x + v4.y + v4.z + v4.w;
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) sm1 += __shfl_xor(sm1, o);
                    const float mean = sm1 * (1.0f / HD);
                    float4 dv = make_float4(v4.x - mean, v4.y - mean, v4.z - mean, v4.w - mean);
                    float sq1 = lane_on ? dv.x * dv.x + dv.y * dv.y + dv.z * dv.z + dv.w * dv.w : 0.f;
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) sq1 += __shfl_xor(sq1, o);
                    const float rs = rsqrtf(sq1 * (1.0f / HD) + 1e-6f);
                    const float4 gw = *(const float4*)((which == 0 ? a.qn_w : a.kn_w) + sub * 4);
                    const float4 gb = *(const float4*)((which == 0 ? a.qn_b : a.kn_b) + sub * 4);
                    r[which] = make_float4(dv.x * rs * gw.x + gb.x, dv.y * rs * gw.y + gb.y, dv.z * rs * gw.z + gb.z,
                                           dv.w * rs * gw.w + gb.w);
                }
            }
            if (lane_on) {
#pragma unroll
                for (int which = 0; which < 3; ++which) *(float4*)(&qkv_s[which][sub * 4]) = r[which];
                // present = (k, v) of this step -> cache row T-1 (mingpt.py:77 / rar.py:96-107)
                *(float4*)(Kc + (long long)(T - 1) * HD) = r[1];
                *(float4*)(Vc + (long long)(T - 1) * HD) = r[2];
            }
        }
    }
    // One wave per workgroup: the hand-over of q / k / v through LDS needs no s_barrier -- and hipcc's __syncthreads() drains EVERY
    // outstanding load first (s_waitcnt vmcnt(0)): the cache chunks requested above would have to land before the streaming loop may
    // even start (stamps: "finish q/k/v" 6.6 us with two chunks in flight).  LDS operations of one wave execute in order.
    if (NWA > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    WMAR_ST(6)
#ifdef WMAR_ATT_TRACE
    tr[2] = __builtin_amdgcn_s_memtime();
#endif
    q = lane_on ? *(const float4*)(&qkv_s[0][sub * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
    knew = *(const float4*)(&qkv_s[1][sub * 4]);
    vnew = *(const float4*)(&qkv_s[2][sub * 4]);
    __builtin_amdgcn_sched_barrier(0);

    float m = -INFINITY, l = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define WMAR_ATT_CHUNK(KB, VB, C0)                                                       \
    {                                                                                    \
        float sc[CH];                                                                    \
        float cm = -INFINITY;                                                            \
        _Pragma("unroll") for (int u = 0; u < CH; ++u) {                                 \
            const int t = (C0) * ROWS + u * RPI + rsel;                                  \
            if (t >= T - 1) { KB[u] = knew; VB[u] = vnew; }                              \
            float p = KB[u].x * q.x + KB[u].y * q.y + KB[u].z * q.z + KB[u].w * q.w;     \
            _Pragma("unroll") for (int o = 1; o < LPR; o <<= 1) p += __shfl_xor(p, o);   \
            p = (t < T) ? p * a.scale : -INFINITY;                                       \
            sc[u] = p;                                                                   \
            cm = fmaxf(cm, p);                                                           \
        }                                                                                \
        _Pragma("unroll") for (int o = LPR; o < 64; o <<= 1) cm = fmaxf(cm, __shfl_xor(cm, o)); \
        const float mn = fmaxf(m, cm);                                                   \
        const float rs = __expf(m - mn);       /* 0 on the first chunk (m = -inf) */     \
        l *= rs; acc.x *= rs; acc.y *= rs; acc.z *= rs; acc.w *= rs;                     \
        _Pragma("unroll") for (int u = 0; u < CH; ++u) {                                 \
            const float e = __expf(sc[u] - mn);                                          \
            l += e;                                                                      \
            acc.x += e * VB[u].x; acc.y += e * VB[u].y; acc.z += e * VB[u].z; acc.w += e * VB[u].w; \
        }                                                                                \
        m = mn;                                                                          \
    }
    // The refills inside the loop are UNCONDITIONAL (rows past the cache clamp to row T-1: L1 hits): a load under a run-time branch
    // makes hipcc's s_waitcnt pass take the smaller outstanding count of the two paths at the merge, i.e. every use of chunk c
    // then waits for chunk c+1's loads as well and the double buffer degenerates to one chunk in flight.
    if (!PF2 && !LATE2) { WMAR_ATT_LOAD(kB, vB, w + NWA) }
    __builtin_amdgcn_sched_barrier(0);
    for (int c = w; c < nchunk; c += 2 * NWA) {
        WMAR_ATT_CHUNK(kA, vA, c)
        __builtin_amdgcn_sched_barrier(0);
        WMAR_ATT_LOAD(kA, vA, c + 2 * NWA)
        __builtin_amdgcn_sched_barrier(0);
        if (c + NWA < nchunk) { WMAR_ATT_CHUNK(kB, vB, c + NWA) }
        __builtin_amdgcn_sched_barrier(0);
        WMAR_ATT_LOAD(kB, vB, c + 3 * NWA)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef WMAR_ATT_LOAD
#undef WMAR_ATT_CHUNK
    WMAR_ST(3)
#ifdef WMAR_ATT_TRACE
    tr[3] = __builtin_amdgcn_s_memtime();
#endif
    // fold the RPI row groups of this wave (l and acc are per-lane partials over the lane's rows)
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        l += __shfl_xor(l, o);
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
        acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (rsel == 0 && lane_on) {
        *(float4*)(&part[w][sub * 4]) = acc;
        if (sub == 0) { part[w][HD] = m; part[w][HD + 1] = l; }
    }
    if (NWA > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    if (w == 0 && rsel == 0 && lane_on) {
        float M = part[0][HD];
#pragma unroll
        for (int i = 1; i < NWA; ++i) M = fmaxf(M, part[i][HD]);
        float L = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NWA; ++i) {
            const float f = __expf(part[i][HD] - M);   // waves without rows: exp(-inf) = 0
            const float4 pa = *(const float4*)(&part[i][sub * 4]);
            L += part[i][HD + 1] * f;
            o.x += pa.x * f; o.y += pa.y * f; o.z += pa.z * f; o.w += pa.w * f;
        }
        const float inv = 1.0f / L;
        // y[b][h*HD + sub*4 .. +3] into the packed activation layout
        const int k = h * HD + sub * 4;
        const int kb = k >> 3, hf = (k >> 2) & 1;
        const int mt = b >> 5;
        const float4 yv = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
        if (a.yq) bx_store_planes4(a.yq, a.MT, kb, hf, mt, b & 31, yv);
        else a.y[((long long)kb * a.MT + mt) * 64 + (b & 31) + 32 * hf] = yv;
    }
    WMAR_ST_END(a.trace, blockIdx.x)
#ifdef WMAR_ATT_TRACE
    if (a.trace && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[4] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 5; ++i) a.trace[(long long)blockIdx.x * 5 + i] = tr[i];
    }
#endif
}

// ------------------------------------------------- decode attention at head_dim 80 (RAR-XL), every lane busy (round 4)
// k_attn_decode lays a cache row on the next power of two of lanes: at head_dim 80 that is 32 lanes of which 20 carry data -- 3/8 of
// every load instruction, dot product and shuffle idle, 40 us x 32 blocks = 31 % of the RAR-XL step.  Here a row is cut into a MAIN
// part of 64 floats (16 lanes x float4: 4 rows per 1-KiB load) and a TAIL of 16 floats (4 lanes x float4: 16 rows per load): 16 cached
// rows are 4 + 1 loads of K and 4 + 1 of V with all 64 lanes carrying data (10 KiB per 16 rows instead of 16 KiB of load slots).
// A lane plays both roles.  Main load u (0..3) holds rows 4u + g in lane group g = lane / 16; the tail load holds row
// 4 ((lane / 4) % 4) + lane / 16 in lane quad lane / 4 -- i.e. a tail lane's row is the row of ITS OWN lane group for u = (lane / 4) % 4,
// so it picks the main partial score out of its own registers (3 selects, no cross-lane traffic), and a main lane fetches the tail
// partial of row (u, g) from lane 16 g + 4 u of its own 16-lane row (one __shfl per main load).  The prologue (QKV pieces, bias,
// q / k LayerNorm over the 80 values, cache append) is k_attn_decode's.
template <int NWA, int MODE = 1>     // MODE: AttnArgs::mode at compile time (see k_attn_decode); RAR is the only head_dim-80 user
__global__ __launch_bounds__(NWA * 64) void k_attn_decode80(AttnArgs a) {
    // NU main loads per chunk (4 rows each): chunks of 4 NU rows.  The tail load always spans 16 rows' worth of lanes; with NU < 4 the
    // lane quads of the rows past the chunk re-read a clamped row and contribute nothing (their weight is 0).
    constexpr int HD = 80, LPRA = 20, LPR = 32, RPI = 2, NU = WMAR_A80_NU, ROWS = 4 * NU;
    __shared__ __attribute__((aligned(16))) float part[NWA][HD + 4];
    __shared__ __attribute__((aligned(16))) float qkv_s[3][HD];
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int lane = threadIdx.x & 63;
    const int w = NWA == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = *a.pos_dev + 1;
    // prologue roles (k_attn_decode's layout: 32 lanes per row, 20 active)
    const int subr = lane % LPR, rsel = lane / LPR;
    const bool lane_on = subr < LPRA;
    const int sub = lane_on ? subr : 0;
    float* Kc = a.kcache + ((long long)b * a.H + h) * a.Tmax * HD;
    float* Vc = a.vcache + ((long long)b * a.H + h) * a.Tmax * HD;
    const int nchunk = (T + ROWS - 1) / ROWS;
    // streaming roles
    const int g16 = lane >> 4, m16 = lane & 15;              // main: row 4u + g16, floats 4 m16 .. + 3
    const int ut = (lane >> 2) & 3, t4 = lane & 3;           // tail: row 4 ut + g16, floats 64 + 4 t4 .. + 3
    const float* Kmain = Kc + m16 * 4;
    const float* Vmain = Vc + m16 * 4;
    const float* Ktail = Kc + 64 + t4 * 4;
    const float* Vtail = Vc + 64 + t4 * 4;
    // rows past T-1 are clamped to T-1 and replaced from registers / masked below
#define WMAR_A80_LOAD(KM, KT, VM, VT, C0)                                                \
    _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                     \
        const int t = min((C0) * ROWS + 4 * u + g16, T - 1);                             \
        KM[u] = WMAR_KV_LD((const float4*)(Kmain + (long long)t * HD));                  \
        VM[u] = WMAR_KV_LD((const float4*)(Vmain + (long long)t * HD));                  \
    }                                                                                    \
    { const int t = min((C0) * ROWS + 4 * min(ut, NU - 1) + g16, T - 1);                 \
      KT = WMAR_KV_LD((const float4*)(Ktail + (long long)t * HD));                       \
      VT = WMAR_KV_LD((const float4*)(Vtail + (long long)t * HD)); }
    float4 kmA[NU], vmA[NU], ktA, vtA, kmB[NU], vmB[NU], ktB, vtB;
    // PF2: the wave's first TWO chunks are requested before the prologue (32 KiB in flight per wave: with 1 / 2 / 4 waves the
    // whole cache up to 64 / 128 / 256 rows streams while q/k/v are finished); otherwise one, the second from inside the loop.
    // (Measured dead end: clamping the first chunk to the cache's capacity instead of its fill, so that its loads need not wait
    // for the scalar load of the position, saves 1.3 us at 64 rows and costs 3 us below 32 rows -- stale rows are then streamed.)
    // The prologue's loads go FIRST: loads return in issue order, so behind 16 KiB of cache rows per wave the QKV pieces would
    // arrive only after the chip-wide burst of first chunks has drained (~4 us); ahead of it they are back in ~1.5 us and q, k, v
    // are finished while the first chunk is still in flight.
    // every load of the prologue is issued before the first wait: LN partial sums (one chunk per
    // lane), the QKV slab(s), the folded-LN row sums and the bias (16 lanes each)
    const int Mpad = a.MT * 32;
    const int mt = b >> 5;
    double sm = 0, sq = 0;
    // Every prologue load is UNCONDITIONAL per lane (indices clamped, unused values masked where they are summed): a load under a
    // per-lane condition (`cond ? load : 0`) compiles to an exec-masked branch whose result is merged right behind it, i.e. one
    // `s_waitcnt vmcnt(0)` per load -- a dozen serialized L2 round trips instead of one (round 3: the ISA showed exactly that).
    double2 st0 = make_double2(0.0, 0.0);
    if (w == 0) st0 = *(const double2*)(a.stats + ((long long)min(lane, a.n_chunks - 1) * Mpad + b) * 2);
    // The S split-K pieces of this head's 3 x hd columns are spread over the wave's RPI row groups (piece p is fetched by
    // group p % RPI), so that a lane holds at most PMAX pieces: all loads are still in flight together, without 3 x 8
    // float4 registers per lane (the kernel's occupancy is set by its registers).  The partial sums meet in a fixed
    // xor-butterfly over the groups: the summation order depends on S only.
    constexpr int PMAX = (QKV_SLABS_MAX + RPI - 1) / RPI;
    float4 sl[3][PMAX], cc[3], bb[3];
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const int n = which * a.D + h * HD + sub * 4;          // first of this lane's 4 columns
        // packed operand layout, or row-major pieces [rows][3 D] (k_qkvx_bx, round 5): this lane's float4 of row b
        const long long idx = a.rowmajor ? ((long long)b * (3 * a.D) + n) >> 2
                                         : ((long long)(n >> 3) * a.MT + mt) * 64 + (b & 31) + 32 * ((n >> 2) & 1);
        if (w == 0) {           // wave-uniform
#pragma unroll
            for (int pi = 0; pi < PMAX; ++pi) {
                const int pc = min(rsel + pi * RPI, a.S - 1);
                sl[which][pi] = a.qkv_slabs[(long long)pc * a.slab_stride + idx];
            }
            cc[which] = *(const float4*)((MODE == 0 ? a.c1 : a.bias) + n);     // (mode 1 has no c1: any valid address, value unused)
            bb[which] = *(const float4*)(a.bias + n);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    WMAR_A80_LOAD(kmA, ktA, vmA, vtA, w)
    __builtin_amdgcn_sched_barrier(0);
    if (w == 0) {
        sm = lane < a.n_chunks ? st0.x : 0.0; sq = lane < a.n_chunks ? st0.y : 0.0;
        // (n_chunks <= STAT_CHUNKS_MAX = 64: one chunk per lane; a loop over further chunks here would put a load in a loop and make
        // hipcc wait for ALL outstanding loads, the first cache chunk included)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
        const double invK = a.invK;
        const double mean = sm * invK;
        const float mu = (float)mean;
        const float rstd = rsqrtf((float)var_f64(sq * invK, mean) + 1e-5f);
        float4 accs[3];
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int pi = 0; pi < PMAX; ++pi)
                if (rsel + pi * RPI < a.S) {
                    acc.x += sl[which][pi].x; acc.y += sl[which][pi].y;
                    acc.z += sl[which][pi].z; acc.w += sl[which][pi].w;
                }
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
                acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
            }
            accs[which] = acc;
        }
        if (rsel == 0) {
            float4 r[3];
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                const float4 acc = accs[which];
                if (MODE == 0) {
                    r[which] = make_float4(rstd * (acc.x - mu * cc[which].x) + bb[which].x,
                                           rstd * (acc.y - mu * cc[which].y) + bb[which].y,
                                           rstd * (acc.z - mu * cc[which].z) + bb[which].z,
                                           rstd * (acc.w - mu * cc[which].w) + bb[which].w);
                } else {
                    r[which] = make_float4(acc.x + bb[which].x, acc.y + bb[which].y, acc.z + bb[which].z, acc.w + bb[which].w);
                }
            }
            if (MODE == 1) {
                // q_norm / k_norm: LayerNorm over the hd values of this head (eps 1e-6, affine)
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    float4 v4 = lane_on ? r[which] : make_float4(0.f, 0.f, 0.f, 0.f);
                    float sm1 = v4.x + v4.y + v4.z + v4.w;
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) sm1 += __shfl_xor(sm1, o);
                    const float mean = sm1 * (1.0f / HD);
                    float4 dv = make_float4(v4.x - mean, v4.y - mean, v4.z - mean, v4.w - mean);
                    float sq1 = lane_on ? dv.x * dv.x + dv.y * dv.y + dv.z * dv.z + dv.w * dv.w : 0.f;
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) sq1 += __shfl_xor(sq1, o);
                    const float rs = rsqrtf(sq1 * (1.0f / HD) + 1e-6f);
                    const float4 gw = *(const float4*)((which == 0 ? a.qn_w : a.kn_w) + sub * 4);
                    const float4 gb = *(const float4*)((which == 0 ? a.qn_b : a.kn_b) + sub * 4);
                    r[which] = make_float4(dv.x * rs * gw.x + gb.x, dv.y * rs * gw.y + gb.y, dv.z * rs * gw.z + gb.z,
                                           dv.w * rs * gw.w + gb.w);
                }
            }
            if (lane_on) {
#pragma unroll
                for (int which = 0; which < 3; ++which) *(float4*)(&qkv_s[which][sub * 4]) = r[which];
                // present = (k, v) of this step -> cache row T-1 (mingpt.py:77 / rar.py:96-107)
                *(float4*)(Kc + (long long)(T - 1) * HD + sub * 4) = r[1];
                *(float4*)(Vc + (long long)(T - 1) * HD + sub * 4) = r[2];
            }
        }
    }
    if (NWA > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();     // one wave: no s_barrier, no vmcnt(0) drain (k_attn_decode)
    const float4 qM = *(const float4*)(&qkv_s[0][m16 * 4]), qT = *(const float4*)(&qkv_s[0][64 + t4 * 4]);
    const float4 knM = *(const float4*)(&qkv_s[1][m16 * 4]), knT = *(const float4*)(&qkv_s[1][64 + t4 * 4]);
    const float4 vnM = *(const float4*)(&qkv_s[2][m16 * 4]), vnT = *(const float4*)(&qkv_s[2][64 + t4 * 4]);
    __builtin_amdgcn_sched_barrier(0);

    float m = -INFINITY, l = 0.f;
    float4 accM = make_float4(0.f, 0.f, 0.f, 0.f), accT = accM;
#define WMAR_A80_CHUNK(KM, KT, VM, VT, C0)                                               \
    {                                                                                    \
        const int tt = (C0) * ROWS + 4 * min(ut, NU - 1) + g16;                          \
        if (tt >= T - 1) { KT = knT; VT = vnT; }                                         \
        float pt = KT.x * qT.x + KT.y * qT.y + KT.z * qT.z + KT.w * qT.w;                \
        pt += __shfl_xor(pt, 1); pt += __shfl_xor(pt, 2);     /* the row's 16 tail floats */ \
        float sc[NU];                                                                    \
        float cm = -INFINITY;                                                            \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                 \
            const int t = (C0) * ROWS + 4 * u + g16;                                     \
            if (t >= T - 1) { KM[u] = knM; VM[u] = vnM; }                                \
            float p = KM[u].x * qM.x + KM[u].y * qM.y + KM[u].z * qM.z + KM[u].w * qM.w; \
            p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4); p += __shfl_xor(p, 8); \
            p += __shfl(pt, (lane & 48) + 4 * u);             /* tail partial of row 4u + g16: quad u of this 16-lane row */ \
            p = (t < T) ? p * a.scale : -INFINITY;                                       \
            sc[u] = p;                                                                   \
            cm = fmaxf(cm, p);                                                           \
        }                                                                                \
        cm = fmaxf(cm, __shfl_xor(cm, 16)); cm = fmaxf(cm, __shfl_xor(cm, 32));          \
        const float mn = fmaxf(m, cm);                                                   \
        const float rs = __expf(m - mn);       /* 0 on the first chunk (m = -inf) */     \
        l *= rs; accM.x *= rs; accM.y *= rs; accM.z *= rs; accM.w *= rs;                 \
        accT.x *= rs; accT.y *= rs; accT.z *= rs; accT.w *= rs;                          \
        float e4[4] = {0.f, 0.f, 0.f, 0.f};                                              \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                 \
            const float e = __expf(sc[u] - mn);                                          \
            e4[u] = e;                                                                   \
            l += e;                                                                      \
            accM.x += e * VM[u].x; accM.y += e * VM[u].y; accM.z += e * VM[u].z; accM.w += e * VM[u].w; \
        }                                                                                \
        /* the tail lane's row is row 4 ut + g16: its weight is this lane's own e4[ut] */ \
        const float et = ut == 0 ? e4[0] : (ut == 1 ? e4[1] : (ut == 2 ? e4[2] : e4[3])); \
        accT.x += et * VT.x; accT.y += et * VT.y; accT.z += et * VT.z; accT.w += et * VT.w; \
        m = mn;                                                                          \
    }
    // refills are UNCONDITIONAL (clamped rows): see k_attn_decode
    WMAR_A80_LOAD(kmB, ktB, vmB, vtB, w + NWA)
    __builtin_amdgcn_sched_barrier(0);
    for (int c = w; c < nchunk; c += 2 * NWA) {
        WMAR_A80_CHUNK(kmA, ktA, vmA, vtA, c)
        __builtin_amdgcn_sched_barrier(0);
        WMAR_A80_LOAD(kmA, ktA, vmA, vtA, c + 2 * NWA)
        __builtin_amdgcn_sched_barrier(0);
        if (c + NWA < nchunk) { WMAR_A80_CHUNK(kmB, ktB, vmB, vtB, c + NWA) }
        __builtin_amdgcn_sched_barrier(0);
        WMAR_A80_LOAD(kmB, ktB, vmB, vtB, c + 3 * NWA)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef WMAR_A80_LOAD
#undef WMAR_A80_CHUNK
    // l counts every row once per lane of its group: fold the four row groups; accM likewise; accT over the 16 quads
    l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        accM.x += __shfl_xor(accM.x, o); accM.y += __shfl_xor(accM.y, o); accM.z += __shfl_xor(accM.z, o); accM.w += __shfl_xor(accM.w, o);
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
        accT.x += __shfl_xor(accT.x, o); accT.y += __shfl_xor(accT.y, o); accT.z += __shfl_xor(accT.z, o); accT.w += __shfl_xor(accT.w, o);
    }
    if (lane < 16) *(float4*)(&part[w][lane * 4]) = accM;
    if (lane < 4) *(float4*)(&part[w][64 + lane * 4]) = accT;
    if (lane == 0) { part[w][HD] = m; part[w][HD + 1] = l; }
    if (NWA > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();     // one wave: no s_barrier, no vmcnt(0) drain (k_attn_decode)
    if (w == 0 && lane < LPRA) {
        float M = part[0][HD];
#pragma unroll
        for (int i = 1; i < NWA; ++i) M = fmaxf(M, part[i][HD]);
        float L = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NWA; ++i) {
            const float f = __expf(part[i][HD] - M);   // waves without rows: exp(-inf) = 0
            const float4 pa = *(const float4*)(&part[i][lane * 4]);
            L += part[i][HD + 1] * f;
            o.x += pa.x * f; o.y += pa.y * f; o.z += pa.z * f; o.w += pa.w * f;
        }
        const float inv = 1.0f / L;
        const int k = h * HD + lane * 4;
        const int kb = k >> 3, hf = (k >> 2) & 1;
        const int mt = b >> 5;
        const float4 yv = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
        if (a.yq) bx_store_planes4(a.yq, a.MT, kb, hf, mt, b & 31, yv);
        else a.y[((long long)kb * a.MT + mt) * 64 + (b & 31) + 32 * hf] = yv;
    }
}

}  // namespace wmar
