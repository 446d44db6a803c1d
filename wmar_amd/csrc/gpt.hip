// minGPT decode engine for gfx950 (fp32, exact-f32 MFMA).
//
// Reference: deps/taming/modules/transformer/mingpt.py:42-214 (CausalSelfAttention, Block,
// GPT.forward_with_past) and :326-368 (sample_with_past).
//
// Data layout in HBM (everything fp32):
//   * Linear weights W[N][K] are repacked once into MFMA-fragment order
//         Wp[nt][kb][lane][4]   nt = n/32, kb = k/8, lane = (n%32) + 32*((k%8)/4), i = k%4
//     so that ONE wave-wide float4 load (1 KiB contiguous) feeds four
//     v_mfma_f32_32x32x2_f32 instructions as the A operand.
//   * Activations X[M][K] (M = batch rows, padded to 32) live in the SAME fragment order
//         Xp[kb][mt][lane][4]   mt = m/32, lane = (m%32) + 32*((k%8)/4)
//     and feed the B operand.  With A = W and B = X^T the MFMA result registers
//     (n = 8*(reg/4) + 4*(lane/32) + reg%4, m = lane%32) ARE the packed layout of the next
//     layer's activation: every GEMM epilogue stores whole float4s, 1 KiB per wave.
//   * KV cache [L][B][H][Tmax][hd]; one wave per (sequence, head) streams its K and V rows
//     with 1 KiB coalesced loads.
//   * The step position lives in device memory so one captured hipGraph replays for all steps.
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "sampler.h"

namespace wmar {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// streamed-once data (weights): non-temporal 16-byte load
__device__ __forceinline__ float4 ld_nt(const float4* p) {
    f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int MAX_SLABS = 8;
constexpr int QKV_SLABS_MAX = 4;   // the QKV projection is split at most 4 ways (1 by default)
constexpr int STAT_CHUNKS_MAX = 64;   // n_embd <= 8192
constexpr int GEMM_STAGE = 4;   // k-blocks (of 8) per register stage of the skinny GEMM

// ------------------------------------------------------------------------ weight packing
// gamma (nullable): the LayerNorm scale of the layer that feeds this Linear, folded into
// the weights ( LN(x) W^T = xhat (W*gamma)^T + W beta ), so the GEMM's inner loop only
// has to form xhat = (x - mean) * rstd.
__global__ void k_pack_linear(const float* __restrict__ W, float4* __restrict__ Wp, int N, int K, int nt_off,
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
__global__ __launch_bounds__(64) void k_fold_bias(const float* __restrict__ W, const float* __restrict__ bias,
                                                  const float* __restrict__ beta, float* __restrict__ out, int K) {
    const int n = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    for (int k = lane; k < K; k += 64) acc += (double)W[(long long)n * K + k] * (double)beta[k];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[n] = (float)((bias ? (double)bias[n] : 0.0) + acc);
}

__global__ void k_set_int(int* p, int v) { *p = v; }
__global__ void k_advance3(int* p) { p[0] += 1; p[1] += 1; p[2] += 1; }  // pos, step, len

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
    __shared__ double red[4][32][2];
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
#pragma unroll
            for (int sidx = 1; sidx < S; ++sidx) {
                acc.x += sl[i][sidx].x; acc.y += sl[i][sidx].y; acc.z += sl[i][sidx].z; acc.w += sl[i][sidx].w;
            }
            r = make_float4(v[i].x + (bb[i].x + acc.x), v[i].y + (bb[i].y + acc.y), v[i].z + (bb[i].z + acc.z),
                            v[i].w + (bb[i].w + acc.w));
        }
        a.x[((long long)kbs[i] * a.MT + mt) * 64 + lane] = r;
        s += (double)r.x + (double)r.y + (double)r.z + (double)r.w;
        ss += (double)r.x * r.x + (double)r.y * r.y + (double)r.z * r.z + (double)r.w * r.w;
    }
    s += __shfl_xor(s, 32);
    ss += __shfl_xor(ss, 32);
    if (lane < 32) { red[w][lane][0] = s; red[w][lane][1] = ss; }
    __syncthreads();
    if (threadIdx.x < 32) {
        double ts = 0, tss = 0;
        for (int i = 0; i < 4; ++i) { ts += red[i][threadIdx.x][0]; tss += red[i][threadIdx.x][1]; }
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
    const double invK = 1.0 / (double)K;
    const double mean = sm * invK;
    *mu = (float)mean;
    *rstd = rsqrtf((float)(sq * invK - mean * mean) + 1e-5f);
}

// ------------------------------------------------------------------------- skinny GEMM
enum { EPI_PACKED = 0, EPI_GELU = 1, EPI_QKV = 2, EPI_LOGITS = 3 };

struct GemmArgs {
    const float4* Wp;          // [NT][KB][64]
    const float4* Xp;          // [KB][MT][64]
    const float* bias;         // [N] or null
    const float* c1;           // [N] row sums of the gamma-folded weights (LN epilogue), or null
    int KB, NT, MT, S;
    // fused LayerNorm on the B operand
    const double* stats; int n_chunks; int K;
    // epilogues
    float4* out_packed; long long slab_stride;          // EPI_PACKED (slab s) / EPI_GELU
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
template <int MTW, int NW, int EPI, bool LN, int ABL = 0, int U = GEMM_STAGE, bool ROT = true>
__global__ __launch_bounds__(NW * 64) void k_gemm(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NW][MTW*16][64]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int MG = a.MT / MTW;
    int bid = blockIdx.x;
    const int nt = bid % a.NT; bid /= a.NT;
    const int mg = bid % MG;
    const int s = bid / MG;
    const int mt0 = mg * MTW;
    const int slices = a.S * NW;
    const int sl = s * NW + w;
    // 32-bit index math (there is no integer divide instruction: 64-bit division is ~10x dearer)
    const int kb0 = (int)((unsigned)sl * (unsigned)a.KB / (unsigned)slices);
    const int kb1 = (int)((unsigned)(sl + 1) * (unsigned)a.KB / (unsigned)slices);
    const int half = lane >> 5;
#ifdef WMAR_GEMM_TRACE
    unsigned long long tr0 = __builtin_amdgcn_s_memtime(), tr1 = 0, tr2 = 0;
#endif

    f32x16 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float mu[MTW], rstd[MTW];
    const float4* Wp = a.Wp + (long long)nt * a.KB * 64 + lane;
    const float4* Xp = a.Xp + (long long)mt0 * 64 + lane;
    const long long xstep = (long long)a.MT * 64;

    // Register double buffer: while the MFMAs of one stage (U k-blocks = 4U instructions per
    // row tile) run, the loads of the next stage are in flight.
    float4 wA[U], wB[U], xA[U][MTW], xB[U][MTW];
#define WMAR_LOAD(WBUF, XBUF, KB0)                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                            \
        const int kk = (KB0) + u;                                                               \
        WBUF[u] = (ABL == 2) ? make_float4(1.f, 2.f, 3.f, (float)kk) : ld_nt(Wp + (long long)kk * 64); \
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
                acc[i][0] += WV.x * XV[i].x + WV.y * XV[i].y + WV.z * XV[i].z + WV.w * XV[i].w; \
        } else {                                                                                \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV.x, XV[i].x, acc[i], 0, 0, 0);      \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV.y, XV[i].y, acc[i], 0, 0, 0);      \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV.z, XV[i].z, acc[i], 0, 0, 0);      \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV.w, XV[i].w, acc[i], 0, 0, 0);      \
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
#ifdef WMAR_GEMM_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr1 = __builtin_amdgcn_s_memtime();
#endif
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
#undef WMAR_STAGE_KB
        kb = kb0 + nst * U;
    } else {
        WMAR_LN_PROLOGUE
    }
    for (; kb < kb1; ++kb) {  // tail (slices that are not a multiple of 2U blocks)
        float4 wv = ld_nt(Wp + (long long)kb * 64);
        float4 xt[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) xt[i] = Xp[(long long)kb * xstep + i * 64];
        WMAR_LNX(xt)
        WMAR_MMA1(wv, xt)
    }
#undef WMAR_MMA1
#undef WMAR_LNX
#undef WMAR_LN_PROLOGUE
#undef WMAR_LOAD
#undef WMAR_MMA

#ifdef WMAR_GEMM_TRACE
    tr2 = __builtin_amdgcn_s_memtime();
#endif
    // in-workgroup K reduction (fixed order) + epilogue
    float* my = smem + (long long)w * (MTW * 16) * 64;
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) my[(i * 16 + r) * 64 + lane] = acc[i][r];
    __syncthreads();

    for (int grp = w; grp < MTW * 4; grp += NW) {
        const int i = grp >> 2, g = grp & 3;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = 0.f;
            for (int ww = 0; ww < NW; ++ww) t += smem[((long long)ww * (MTW * 16) + i * 16 + g * 4 + j) * 64 + lane];
            o[j] = t;
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
            float4* dst = a.out_packed + (long long)s * a.slab_stride + ((long long)(nt * 4 + g) * a.MT + mt) * 64 + lane;
            *dst = make_float4(o[0], o[1], o[2], o[3]);
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
#ifdef WMAR_GEMM_TRACE
    if (a.trace && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* t = a.trace + (long long)blockIdx.x * 4;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// --------------------------------------------------------------------- decode attention
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
    const float* c1;           // [3D] row sums of the gamma-folded QKV weights
    const float* bias;         // [3D] bias + W beta
    float* kcache;             // [B][H][Tmax][hd] (this layer)
    float* vcache;
    float4* y;                 // packed [KB][MT][64]
    const int* pos_dev;
    int D, H, Tmax, MT;
    float scale;
    int dbg;                   // dev ablation (WMAR_ATT_DBG): 2 = skip K/V streaming
};

// One workgroup of NWA waves per (sequence, head).  The cached rows are cut into chunks of
// CH 1-KiB loads (CH*RPI rows); wave w takes chunks w, w+NWA, ...  and keeps a running
// (max, sum, weighted V sum) in registers -- K and V of a chunk are requested together, the next
// chunk is in flight while the current one is reduced.  The NWA partial results meet in LDS.
template <int HD, int NWA>
__global__ __launch_bounds__(NWA * 64) void k_attn_decode(AttnArgs a) {
    constexpr int LPR = HD / 4, RPI = 64 / LPR;
    constexpr int CH = 8;
    constexpr int ROWS = CH * RPI;
    __shared__ float part[NWA][HD + 2];    // per wave: weighted V sum [HD], max, sum
    __shared__ __attribute__((aligned(16))) float qkv_s[3][HD];
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = *a.pos_dev + 1;
    const int sub = lane % LPR, rsel = lane / LPR;
    float* Kc = a.kcache + ((long long)b * a.H + h) * a.Tmax * HD + sub * 4;
    float* Vc = a.vcache + ((long long)b * a.H + h) * a.Tmax * HD + sub * 4;
    const int nchunk = (a.dbg & 2) ? 0 : (T + ROWS - 1) / ROWS;

    // rows past T-1 are clamped to T-1 and replaced from registers / masked below
#define WMAR_ATT_LOAD(KB, VB, C0)                                                        \
    _Pragma("unroll") for (int u = 0; u < CH; ++u) {                                     \
        const int t = min((C0) * ROWS + u * RPI + rsel, T - 1);                          \
        KB[u] = *(const float4*)(Kc + (long long)t * HD);                                \
        VB[u] = *(const float4*)(Vc + (long long)t * HD);                                \
    }
    float4 kA[CH], vA[CH], kB[CH], vB[CH];
    float4 q, knew, vnew;
    if (w < nchunk) { WMAR_ATT_LOAD(kA, vA, w) }   // in flight while q/k/v are finished
    __builtin_amdgcn_sched_barrier(0);
    // ---- wave 0 finishes this head's q, k, v for the new token from the QKV slab(s) and hands
    // them to the other waves through LDS.  Loads are issued by as few lanes as possible: a
    // wave-wide load of one shared address still costs the address path a full 64-lane pass.
    if (w == 0) {
        // every load of the prologue is issued before the first wait: LN partial sums (one chunk per
        // lane), the QKV slab(s), the folded-LN row sums and the bias (16 lanes each)
        const int Mpad = a.MT * 32;
        const int mt = b >> 5;
        double sm = 0, sq = 0;
        double2 st0 = make_double2(0.0, 0.0);
        if (lane < a.n_chunks) st0 = *(const double2*)(a.stats + ((long long)lane * Mpad + b) * 2);
        float4 sl[3][QKV_SLABS_MAX], cc[3], bb[3];
        if (rsel == 0) {
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                const int n = which * a.D + h * HD + sub * 4;          // first of this lane's 4 columns
                const long long idx = ((long long)(n >> 3) * a.MT + mt) * 64 + (b & 31) + 32 * ((n >> 2) & 1);
#pragma unroll
                for (int sidx = 0; sidx < QKV_SLABS_MAX; ++sidx)
                    sl[which][sidx] = a.qkv_slabs[(long long)min(sidx, a.S - 1) * a.slab_stride + idx];
                cc[which] = *(const float4*)(a.c1 + n);
                bb[which] = *(const float4*)(a.bias + n);
            }
        }
        sm = st0.x; sq = st0.y;
        for (int c = lane + 64; c < a.n_chunks; c += 64) {     // n_embd > 8192 only
            const double2 v = *(const double2*)(a.stats + ((long long)c * Mpad + b) * 2);
            sm += v.x; sq += v.y;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
        const double invK = 1.0 / (double)a.K;
        const double mean = sm * invK;
        const float mu = (float)mean;
        const float rstd = rsqrtf((float)(sq * invK - mean * mean) + 1e-5f);
        if (rsel == 0) {
            float4 r[3];
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                float4 acc = sl[which][0];
#pragma unroll
                for (int sidx = 1; sidx < QKV_SLABS_MAX; ++sidx)
                    if (sidx < a.S) {
                        acc.x += sl[which][sidx].x; acc.y += sl[which][sidx].y;
                        acc.z += sl[which][sidx].z; acc.w += sl[which][sidx].w;
                    }
                r[which] = make_float4(rstd * (acc.x - mu * cc[which].x) + bb[which].x,
                                       rstd * (acc.y - mu * cc[which].y) + bb[which].y,
                                       rstd * (acc.z - mu * cc[which].z) + bb[which].z,
                                       rstd * (acc.w - mu * cc[which].w) + bb[which].w);
                *(float4*)(&qkv_s[which][sub * 4]) = r[which];
            }
            // present = (k, v) of this step -> cache row T-1 (mingpt.py:77)
            *(float4*)(Kc + (long long)(T - 1) * HD) = r[1];
            *(float4*)(Vc + (long long)(T - 1) * HD) = r[2];
        }
    }
    __syncthreads();
    q = *(const float4*)(&qkv_s[0][sub * 4]);
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
    for (int c = w; c < nchunk; c += 2 * NWA) {
        if (c + NWA < nchunk) { WMAR_ATT_LOAD(kB, vB, c + NWA) }
        __builtin_amdgcn_sched_barrier(0);
        WMAR_ATT_CHUNK(kA, vA, c)
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 * NWA < nchunk) { WMAR_ATT_LOAD(kA, vA, c + 2 * NWA) }
        __builtin_amdgcn_sched_barrier(0);
        if (c + NWA < nchunk) { WMAR_ATT_CHUNK(kB, vB, c + NWA) }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef WMAR_ATT_LOAD
#undef WMAR_ATT_CHUNK
    // fold the RPI row groups of this wave (l and acc are per-lane partials over the lane's rows)
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        l += __shfl_xor(l, o);
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
        acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (rsel == 0) {
        *(float4*)(&part[w][sub * 4]) = acc;
        if (sub == 0) { part[w][HD] = m; part[w][HD + 1] = l; }
    }
    __syncthreads();
    if (w == 0 && rsel == 0) {
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
        a.y[((long long)kb * a.MT + mt) * 64 + (b & 31) + 32 * hf] = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
    }
}

}  // namespace wmar

using namespace wmar;

// ------------------------------------------------------------------------------ engine
struct LayerW {
    float4 *wqkv, *wproj, *wfc1, *wfc2;
    float *bqkv, *bproj, *bfc1, *bfc2, *cqkv, *cfc1;
};

struct wmar_gpt {
    wmar_gpt_config cfg{};
    int D = 0, H = 0, hd = 0, V = 0, L = 0, Tmax = 0, Bmax = 0, MTmax = 0;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    std::vector<LayerW> layers;
    float *tok_emb = nullptr, *pos_emb = nullptr, *bhead = nullptr, *chead = nullptr;
    float4* whead = nullptr;
    // workspaces
    float4 *x = nullptr, *y = nullptr, *hbuf = nullptr, *slabs = nullptr, *qkv_slabs = nullptr;
    float* qbuf = nullptr;
    double* stats = nullptr;
    float *kcache = nullptr, *vcache = nullptr;
    float *logits = nullptr, *scratch = nullptr;
    long long* past = nullptr;  // [Bmax][Tmax+1]
    int *pos_dev = nullptr, *step_dev = nullptr;
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    unsigned long long graph_key[12] = {0};
    bool pending = false;          // replays of `exec` may still be running (ev1 marks their end)
    void drop_graph() {
        if (pending && ev1) (void)hipEventSynchronize(ev1);
        pending = false;
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr;
    }
    int timing = 0;
    int force_s[3] = {0, 0, 0};   // tuning knobs: WMAR_S_QKV / WMAR_S_PROJ / WMAR_S_FC2
    double step_ms = 0.0;
    // per-launch event timing (eager mode only)
    struct Span { int cls; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    bool span_on = false;
    double cls_us[WMAR_T_NCLASS] = {0};
    int64_t cls_calls[WMAR_T_NCLASS] = {0};
    hipEvent_t next_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }
    void span_begin(int cls, hipStream_t st) {
        if (!span_on) return;
        Span s{cls, next_event(), next_event()};
        (void)hipEventRecord(s.a, st);
        spans.push_back(s);
    }
    void span_end(hipStream_t st) {
        if (!span_on) return;
        (void)hipEventRecord(spans.back().b, st);
    }
    void span_collect() {
        for (int i = 0; i < WMAR_T_NCLASS; ++i) { cls_us[i] = 0; cls_calls[i] = 0; }
        for (auto& s : spans) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { cls_us[s.cls] += ms * 1000.0; cls_calls[s.cls] += 1; }
        }
        spans.clear();
        ev_used = 0;
    }

    template <typename T>
    int alloc(T** p, size_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, n * sizeof(T));
        if (e != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
            return WMAR_ENOMEM;
        }
        allocs.push_back(q);
        bytes += (int64_t)(n * sizeof(T));
        *p = (T*)q;
        return WMAR_OK;
    }
    ~wmar_gpt() {
        drop_graph();
        for (void* p : allocs) (void)hipFree(p);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    }
};

namespace {

int mt_for(int64_t B) { return B <= 32 ? 1 : (B <= 64 ? 2 : 4); }

struct TensorMap {
    std::map<std::string, const void*> m;
    const float* get(const std::string& k) const {
        auto it = m.find(k);
        return it == m.end() ? nullptr : (const float*)it->second;
    }
};

int pack(wmar_gpt* g, const float* W, float4* Wp, int N, int K, int nt_off, hipStream_t st,
         const float* gamma = nullptr) {
    long long total = (long long)(N / 32) * (K / 8) * 64;
    hipLaunchKernelGGL(k_pack_linear, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, Wp, N, K, nt_off, K / 8,
                       gamma);
    return launch_status("k_pack_linear");
}

// dst[0..N) = bias + W beta
int fold_bias(const float* W, const float* bias, const float* beta, float* dst, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL(k_fold_bias, dim3((unsigned)N), dim3(64), 0, st, W, bias, beta, dst, K);
    return launch_status("k_fold_bias");
}

int copy_vec(wmar_gpt* g, float** dst, const float* src, size_t n, hipStream_t st) {
    if (int rc = g->alloc(dst, n)) return rc;
    WMAR_HIP_CHECK(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return WMAR_OK;
}

template <int MTW, int NW, int EPI, bool LN, int ABL = 0, int U = GEMM_STAGE, bool ROT = true>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
    const int grid = a.NT * (a.MT / MTW) * a.S;
    const size_t lds = (size_t)NW * MTW * 16 * 64 * sizeof(float);
    hipLaunchKernelGGL((k_gemm<MTW, NW, EPI, LN, ABL, U, ROT>), dim3((unsigned)grid), dim3(NW * 64), lds, st, a);
    return launch_status("k_gemm");
}

// Split-K factor for a GEMM whose partial slabs are folded by a later kernel: pick the S that
// fills the 256 CUs most evenly (whole "rounds" of workgroups), keeping >= 8 k-blocks per wave.
int pick_split(int tiles, int KB, int NW) {
    int best = 1;
    double best_eff = 0.0;
    for (int S = 1; S <= MAX_SLABS; ++S) {
        if (KB / (S * NW) < 8 && S > 1) break;
        const int wgs = tiles * S;
        const int rounds = (wgs + 255) / 256;
        const double eff = (double)wgs / (rounds * 256.0);
        if (eff > best_eff + 0.02) { best_eff = eff; best = S; }
    }
    return best;
}

// Row tiles per workgroup: two 32-row tiles share every weight fragment (half the operand
// traffic per MFMA); a single tile when the batch has only one.
template <int EPI, bool LN>
int gemm_dispatch(GemmArgs a, bool allow_split, hipStream_t st) {
    constexpr int NW = 4;
    if (a.MT % 2 == 0) {
        a.S = allow_split ? pick_split(a.NT * (a.MT / 2), a.KB, NW) : 1;
        return launch_gemm<2, NW, EPI, LN>(a, st);
    }
    a.S = allow_split ? pick_split(a.NT * a.MT, a.KB, NW) : 1;
    return launch_gemm<1, NW, EPI, LN>(a, st);
}

// split-K GEMM writing partial slabs; reports the S it used
int gemm_split(GemmArgs a, int* S_out, hipStream_t st, int force_S = 0) {
    constexpr int NW = 4;
    if (a.MT % 2 == 0) {
        a.S = force_S > 0 ? force_S : pick_split(a.NT * (a.MT / 2), a.KB, NW);
        *S_out = a.S;
        return launch_gemm<2, NW, EPI_PACKED, false>(a, st);
    }
    a.S = force_S > 0 ? force_S : pick_split(a.NT * a.MT, a.KB, NW);
    *S_out = a.S;
    return launch_gemm<1, NW, EPI_PACKED, false>(a, st);
}

// chunks of <= 16 k-blocks: one k_resid_stats workgroup (4 waves x 4 blocks) per chunk and row tile
int stat_chunks(int KB) { return (KB + 15) / 16; }

struct StepIO {
    const long long* tok;  // row m's token: tok[m*stride + (use_pos ? pos : 0)]
    long long tok_stride;
    int tok_use_pos;
    float* logits;
};

// One decode step = the launches below, in order.  Each role is its own function so that
// bench.py can also replay a single role back to back (wmar_gpt_profile_role).
struct StepPlan {
    wmar_gpt* g;
    int64_t B;
    StepIO io;
    hipStream_t st;
    int MT, D, KBD, KBF, nch;
    long long act;
    ResidArgs r{};
    int S_qkv = 1, S_proj = 1, S_fc2 = 1;

    StepPlan(wmar_gpt* g_, int64_t B_, const StepIO& io_, hipStream_t st_) : g(g_), B(B_), io(io_), st(st_) {
        MT = mt_for(B); D = g->D; KBD = D / 8; KBF = 4 * D / 8; nch = stat_chunks(KBD);
        act = (long long)KBD * MT * 64;  // float4 units of one [M][D] packed activation
        r.x = g->x; r.stats = g->stats; r.KB = KBD; r.MT = MT; r.n_chunks = nch;
        r.tok_emb = g->tok_emb; r.pos_emb = g->pos_emb; r.tok = io.tok; r.tok_stride = io.tok_stride;
        r.tok_use_pos = io.tok_use_pos; r.pos_dev = g->pos_dev; r.B = (int)B; r.K = D;
        r.slabs = g->slabs; r.slab_stride = act;
        // split factors (fixed per shape so that a role can be replayed on its own)
        S_qkv = g->force_s[0] > 0 ? (g->force_s[0] > QKV_SLABS_MAX ? QKV_SLABS_MAX : g->force_s[0]) : 1;
        S_proj = g->force_s[1] > 0 ? g->force_s[1] : split_for(D / 32, KBD);
        S_fc2 = g->force_s[2] > 0 ? g->force_s[2] : ((MT % 2 == 0 && KBF >= 128) ? 4 : split_for(D / 32, KBF));
    }
    int split_for(int NT, int KB) const { return pick_split(MT % 2 == 0 ? NT * (MT / 2) : NT * MT, KB, 4); }
    GemmArgs base() const {
        GemmArgs a{};
        a.MT = MT; a.B = (int)B; a.stats = g->stats; a.n_chunks = nch; a.K = D;
        a.pos_dev = g->pos_dev; a.D = D; a.H = g->H; a.hd = g->hd; a.Tmax = g->Tmax;
        return a;
    }
    int embed() {
        g->span_begin(WMAR_T_EMBED, st);
        hipLaunchKernelGGL((k_resid_stats<true, 0>), dim3(nch * MT), dim3(256), 0, st, r);
        g->span_end(st);
        return launch_status("k_resid_stats<embed>");
    }
    // x += bias + sum of the S partial slabs; LN statistics of the new rows
    int resid(const float* bias, int S) {
        r.S = S; r.bias = bias;
        g->span_begin(WMAR_T_RESID, st);
        int rc = launch_resid(r, nch * MT, st);
        g->span_end(st);
        return rc;
    }
    // QKV projection of the RAW residual stream into slabs; LN1, bias and the KV-cache append are
    // finished per (sequence, head) in the attention kernel's prologue.  Measured: one slab (no K
    // split across workgroups) beats every split for this shape -- more, thinner workgroups per CU
    // contend for the same L1 fill path.
    int qkv(int l) {
        GemmArgs a = base();
        const LayerW& w = g->layers[l];
        a.Wp = w.wqkv; a.Xp = g->x; a.KB = KBD; a.NT = 3 * D / 32;
        a.out_packed = g->qkv_slabs; a.slab_stride = 3 * act;
        int S = 1;
        g->span_begin(WMAR_T_QKV, st);
        int rc = gemm_split(a, &S, st, S_qkv);
        g->span_end(st);
        return rc;
    }
    int attn(int l) {
        const LayerW& w = g->layers[l];
        const long long lstride = (long long)g->Bmax * g->H * g->Tmax * g->hd;
        AttnArgs t{};
        t.qkv_slabs = g->qkv_slabs; t.slab_stride = 3 * act; t.S = S_qkv; t.stats = g->stats; t.n_chunks = nch; t.K = D;
        t.c1 = w.cqkv; t.bias = w.bqkv;
        t.kcache = g->kcache + l * lstride; t.vcache = g->vcache + l * lstride; t.y = g->y; t.pos_dev = g->pos_dev;
        t.D = D; t.H = g->H; t.Tmax = g->Tmax; t.MT = MT; t.scale = 1.0f / sqrtf((float)g->hd);
        { const char* e = getenv("WMAR_ATT_DBG"); t.dbg = e ? atoi(e) : 0; }
        int nwa = 2;   // measured best at the average cache length (kv=128)
        { const char* e = getenv("WMAR_ATT_NW"); if (e) nwa = atoi(e); }
        const dim3 grid((unsigned)(B * g->H));
        g->span_begin(WMAR_T_ATTN, st);
#define WMAR_ATT_LAUNCH(HDV)                                                                              \
        if (nwa == 1) hipLaunchKernelGGL((k_attn_decode<HDV, 1>), grid, dim3(64), 0, st, t);                   \
        else if (nwa == 2) hipLaunchKernelGGL((k_attn_decode<HDV, 2>), grid, dim3(128), 0, st, t);             \
        else hipLaunchKernelGGL((k_attn_decode<HDV, 4>), grid, dim3(256), 0, st, t);
        if (g->hd == 64) { WMAR_ATT_LAUNCH(64) }
        else if (g->hd == 32) { WMAR_ATT_LAUNCH(32) }
        else { WMAR_ATT_LAUNCH(128) }
#undef WMAR_ATT_LAUNCH
        g->span_end(st);
        return launch_status("k_attn_decode");
    }
    // attention output projection (split-K slabs; bias + residual folded by the next resid())
    int proj(int l) {
        GemmArgs p = base();
        p.Wp = g->layers[l].wproj; p.Xp = g->y; p.KB = KBD; p.NT = D / 32;
        p.out_packed = g->slabs; p.slab_stride = act;
        int S = 1;
        g->span_begin(WMAR_T_PROJ, st);
        int rc = gemm_split(p, &S, st, S_proj);
        g->span_end(st);
        return rc;
    }
    // LN2 -> FC1 (+bias, GELU) -> packed hidden
    int fc1(int l) {
        GemmArgs f = base();
        const LayerW& w = g->layers[l];
        f.Wp = w.wfc1; f.Xp = g->x; f.bias = w.bfc1; f.c1 = w.cfc1; f.KB = KBD; f.NT = 4 * D / 32;
        f.out_packed = g->hbuf; f.slab_stride = 0;
        g->span_begin(WMAR_T_FC1, st);
        int rc = gemm_dispatch<EPI_GELU, true>(f, false, st);
        g->span_end(st);
        return rc;
    }
    int fc2(int l) {
        GemmArgs q = base();
        q.Wp = g->layers[l].wfc2; q.Xp = g->hbuf; q.KB = KBF; q.NT = D / 32;
        q.out_packed = g->slabs; q.slab_stride = act;
        int S = 1;
        g->span_begin(WMAR_T_FC2, st);
        int rc = gemm_split(q, &S, st, S_fc2);
        g->span_end(st);
        return rc;
    }
    // ln_f -> vocabulary head
    int head() {
        GemmArgs h = base();
        h.Wp = g->whead; h.Xp = g->x; h.KB = KBD; h.NT = g->V / 32;
        h.bias = g->bhead; h.c1 = g->chead; h.logits = io.logits; h.V = g->V;
        g->span_begin(WMAR_T_HEAD, st);
        int rc = gemm_dispatch<EPI_LOGITS, true>(h, false, st);
        g->span_end(st);
        return rc;
    }
};

int enqueue_step(wmar_gpt* g, int64_t B, const StepIO& io, hipStream_t st) {
    StepPlan p(g, B, io, st);
    int rc;
    if ((rc = p.embed())) return rc;
    for (int l = 0; l < g->L; ++l) {
        if (l > 0 && (rc = p.resid(g->layers[l - 1].bfc2, p.S_fc2))) return rc;
        if ((rc = p.qkv(l))) return rc;
        if ((rc = p.attn(l))) return rc;
        if ((rc = p.proj(l))) return rc;
        if ((rc = p.resid(g->layers[l].bproj, p.S_proj))) return rc;
        if ((rc = p.fc1(l))) return rc;
        if ((rc = p.fc2(l))) return rc;
    }
    if ((rc = p.resid(g->layers[g->L - 1].bfc2, p.S_fc2))) return rc;
    return p.head();
}

}  // namespace

extern "C" {

int wmar_gpt_create(const wmar_gpt_config* cfg, const char* const* names, const void* const* tensors_dev,
                    int32_t n_tensors, void* stream, wmar_gpt** out) {
    WMAR_REQUIRE(cfg && names && tensors_dev && out, "gpt_create: null argument");
    WMAR_REQUIRE(cfg->n_embd % cfg->n_head == 0, "n_embd %% n_head != 0");
    const int D = cfg->n_embd, H = cfg->n_head, hd = D / H, V = cfg->vocab_size, L = cfg->n_layer;
    WMAR_REQUIRE(D % 32 == 0 && V % 32 == 0 && D <= 8192, "n_embd (<= 8192) and vocab_size must be multiples of 32 (got %d, %d)", D, V);
    WMAR_REQUIRE(hd == 32 || hd == 64 || hd == 128, "head_dim %d unsupported (32, 64, 128)", hd);
    WMAR_REQUIRE(cfg->max_batch >= 1 && cfg->max_batch <= 128, "max_batch must be in 1..128");
    WMAR_REQUIRE(cfg->block_size >= 1 && L >= 1, "bad block_size / n_layer");
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) tm.m[names[i]] = tensors_dev[i];
    hipStream_t st = (hipStream_t)stream;
    auto* g = new wmar_gpt();
    g->cfg = *cfg; g->D = D; g->H = H; g->hd = hd; g->V = V; g->L = L; g->Tmax = cfg->block_size; g->Bmax = cfg->max_batch;
    g->MTmax = mt_for(cfg->max_batch);
    {
        const char* names_s[3] = {"WMAR_S_QKV", "WMAR_S_PROJ", "WMAR_S_FC2"};
        for (int i = 0; i < 3; ++i) {
            const char* e = getenv(names_s[i]);
            if (e) g->force_s[i] = atoi(e) > MAX_SLABS ? MAX_SLABS : atoi(e);
        }
    }
    int rc = WMAR_OK;
    auto need = [&](const std::string& k) -> const float* {
        const float* p = tm.get(k);
        if (!p && rc == WMAR_OK) { set_error("checkpoint tensor '%s' is missing", k.c_str()); rc = WMAR_EMISSING; }
        return p;
    };
#define TRY(x) do { if (rc == WMAR_OK) rc = (x); } while (0)
    const float* te = need("tok_emb.weight");
    const float* pe = need("pos_emb");
    const float* hw = need("head.weight");
    const float *lfw = need("ln_f.weight"), *lfb = need("ln_f.bias");
    if (rc == WMAR_OK) {
        TRY(copy_vec(g, &g->tok_emb, te, (size_t)V * D, st));
        TRY(copy_vec(g, &g->pos_emb, pe, (size_t)cfg->block_size * D, st));
        TRY(g->alloc(&g->whead, (size_t)V * D / 4));
        TRY(pack(g, hw, g->whead, V, D, 0, st, lfw));
        TRY(g->alloc(&g->bhead, (size_t)V));
        TRY(fold_bias(hw, nullptr, lfb, g->bhead, V, D, st));
        TRY(g->alloc(&g->chead, (size_t)V));
        TRY(fold_bias(hw, nullptr, lfw, g->chead, V, D, st));
    }
    g->layers.resize(L);
    for (int l = 0; l < L && rc == WMAR_OK; ++l) {
        std::string p = "blocks." + std::to_string(l) + ".";
        LayerW& w = g->layers[l];
        const float *qw = need(p + "attn.query.weight"), *kw = need(p + "attn.key.weight"), *vw = need(p + "attn.value.weight");
        const float *qb = need(p + "attn.query.bias"), *kb = need(p + "attn.key.bias"), *vb = need(p + "attn.value.bias");
        const float *pw = need(p + "attn.proj.weight"), *pb = need(p + "attn.proj.bias");
        const float *f1w = need(p + "mlp.0.weight"), *f1b = need(p + "mlp.0.bias");
        const float *f2w = need(p + "mlp.2.weight"), *f2b = need(p + "mlp.2.bias");
        const float *l1w = need(p + "ln1.weight"), *l1b = need(p + "ln1.bias"), *l2w = need(p + "ln2.weight"), *l2b = need(p + "ln2.bias");
        if (rc != WMAR_OK) break;
        TRY(g->alloc(&w.wqkv, (size_t)3 * D * D / 4));
        TRY(pack(g, qw, w.wqkv, D, D, 0, st, l1w));
        TRY(pack(g, kw, w.wqkv, D, D, D / 32, st, l1w));
        TRY(pack(g, vw, w.wqkv, D, D, 2 * D / 32, st, l1w));
        TRY(g->alloc(&w.bqkv, (size_t)3 * D));
        TRY(fold_bias(qw, qb, l1b, w.bqkv, D, D, st));
        TRY(fold_bias(kw, kb, l1b, w.bqkv + D, D, D, st));
        TRY(fold_bias(vw, vb, l1b, w.bqkv + 2 * D, D, D, st));
        TRY(g->alloc(&w.cqkv, (size_t)3 * D));
        TRY(fold_bias(qw, nullptr, l1w, w.cqkv, D, D, st));
        TRY(fold_bias(kw, nullptr, l1w, w.cqkv + D, D, D, st));
        TRY(fold_bias(vw, nullptr, l1w, w.cqkv + 2 * D, D, D, st));
        TRY(g->alloc(&w.wproj, (size_t)D * D / 4));
        TRY(pack(g, pw, w.wproj, D, D, 0, st));
        TRY(copy_vec(g, &w.bproj, pb, D, st));
        TRY(g->alloc(&w.wfc1, (size_t)4 * D * D / 4));
        TRY(pack(g, f1w, w.wfc1, 4 * D, D, 0, st, l2w));
        TRY(g->alloc(&w.bfc1, (size_t)4 * D));
        TRY(fold_bias(f1w, f1b, l2b, w.bfc1, 4 * D, D, st));
        TRY(g->alloc(&w.cfc1, (size_t)4 * D));
        TRY(fold_bias(f1w, nullptr, l2w, w.cfc1, 4 * D, D, st));
        TRY(g->alloc(&w.wfc2, (size_t)4 * D * D / 4));
        TRY(pack(g, f2w, w.wfc2, D, 4 * D, 0, st));
        TRY(copy_vec(g, &w.bfc2, f2b, D, st));
    }
    const size_t Mpad = (size_t)g->MTmax * 32;
    TRY(g->alloc(&g->x, Mpad * D / 4));
    TRY(g->alloc(&g->y, Mpad * D / 4));
    TRY(g->alloc(&g->hbuf, Mpad * 4 * D / 4));
    TRY(g->alloc(&g->slabs, (size_t)MAX_SLABS * Mpad * D / 4));
    TRY(g->alloc(&g->qkv_slabs, (size_t)MAX_SLABS * Mpad * 3 * D / 4));
    TRY(g->alloc(&g->qbuf, Mpad * D));
    TRY(g->alloc(&g->stats, (size_t)STAT_CHUNKS_MAX * Mpad * 2));
    const size_t kv = (size_t)L * g->Bmax * H * g->Tmax * hd;
    TRY(g->alloc(&g->kcache, kv));
    TRY(g->alloc(&g->vcache, kv));
    TRY(g->alloc(&g->logits, (size_t)g->Bmax * V));
    TRY(g->alloc(&g->scratch, (size_t)g->Bmax * V));
    TRY(g->alloc(&g->past, (size_t)g->Bmax * (g->Tmax + 1)));
    TRY(g->alloc(&g->pos_dev, 4));
    g->step_dev = g->pos_dev + 1;
    if (rc == WMAR_OK) {
        // padded rows of the packed buffers must hold finite numbers
        hipError_t e = hipMemsetAsync(g->x, 0, Mpad * D * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->kcache, 0, kv * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->vcache, 0, kv * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->hbuf, 0, Mpad * 4 * D * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->past, 0, (size_t)g->Bmax * (g->Tmax + 1) * 8, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->y, 0, Mpad * D * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->qbuf, 0, Mpad * D * 4, st);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->cap_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&g->ev0);
        if (e == hipSuccess) e = hipEventCreate(&g->ev1);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("gpt_create: %s", hipGetErrorString(e)); rc = WMAR_EHIP; }
    }
#undef TRY
    if (rc != WMAR_OK) { delete g; return rc; }
    *out = g;
    return WMAR_OK;
}

void wmar_gpt_destroy(wmar_gpt* g) { delete g; }
int64_t wmar_gpt_device_bytes(const wmar_gpt* g) { return g ? g->bytes : 0; }

int wmar_gpt_decode_step(wmar_gpt* g, const int64_t* tok_dev, int64_t B, int32_t pos, float* logits_dev, void* stream) {
    WMAR_REQUIRE(g && tok_dev && logits_dev, "decode_step: null argument");
    WMAR_REQUIRE(B >= 1 && B <= g->Bmax, "decode_step: batch %lld outside 1..%d", (long long)B, g->Bmax);
    WMAR_REQUIRE(pos >= 0 && pos < g->Tmax, "decode_step: position %d outside the block size %d", pos, g->Tmax);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, g->pos_dev, (int)pos);
    StepIO io{(const long long*)tok_dev, 1, 0, logits_dev};
    return enqueue_step(g, B, io, st);
}

int wmar_gpt_profile_role(wmar_gpt* g, int32_t role, int64_t B, int32_t kv_len, int32_t iters, void* stream,
                          double* avg_us) {
    WMAR_REQUIRE(g && avg_us && iters > 0, "profile_role: bad argument");
    WMAR_REQUIRE(B >= 1 && B <= g->Bmax, "profile_role: batch outside 1..%d", g->Bmax);
    WMAR_REQUIRE(kv_len >= 1 && kv_len <= g->Tmax, "profile_role: kv_len outside 1..%d", g->Tmax);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, g->pos_dev, (int)kv_len - 1);
    StepIO io{g->past, (long long)g->Tmax + 1, 0, g->logits};
    StepPlan p(g, B, io, st);
    auto one = [&](int it) -> int {
        const int l = it % g->L;
        switch (role) {
            case WMAR_T_EMBED: return p.embed();
            case WMAR_T_QKV: return p.qkv(l);
            case WMAR_T_ATTN: return p.attn(l);
            case WMAR_T_PROJ: return p.proj(l);
            case WMAR_T_RESID: return p.resid(g->layers[l].bproj, p.S_proj);
            case WMAR_T_FC1: return p.fc1(l);
            case WMAR_T_FC2: return p.fc2(l);
            case WMAR_T_HEAD: return p.head();
            default: set_error("profile_role: unknown role %d", role); return WMAR_EINVAL;
        }
    };
    const bool was = g->span_on;
    g->span_on = false;
    int rc = WMAR_OK;
    for (int i = 0; i < 3 && rc == WMAR_OK; ++i) rc = one(i);           // warm-up
    if (rc == WMAR_OK && hipEventRecord(g->ev0, st) != hipSuccess) rc = WMAR_EHIP;
    for (int i = 0; i < iters && rc == WMAR_OK; ++i) rc = one(i + 3);
    if (rc == WMAR_OK && hipEventRecord(g->ev1, st) != hipSuccess) rc = WMAR_EHIP;
    g->span_on = was;
    if (rc) return rc;
    WMAR_HIP_CHECK(hipEventSynchronize(g->ev1));
    float ms = 0.f;
    WMAR_HIP_CHECK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    *avg_us = (double)ms * 1000.0 / iters;
    return WMAR_OK;
}

int wmar_gpt_set_timing(wmar_gpt* g, int32_t enabled) { if (!g) return WMAR_EINVAL; g->timing = enabled; return WMAR_OK; }
int wmar_gpt_get_timing(wmar_gpt* g, double* total_us, int64_t* calls, double* step_ms) {
    if (!g) return WMAR_EINVAL;
    for (int i = 0; i < WMAR_T_NCLASS; ++i) {
        if (total_us) total_us[i] = g->cls_us[i];
        if (calls) calls[i] = g->cls_calls[i];
    }
    if (step_ms) *step_ms = g->step_ms;
    return WMAR_OK;
}

int wmar_gpt_generate(wmar_gpt* g, const wmar_wm_ctx* wm, const wmar_sample_params* sp, const int64_t* cond_dev,
                      int64_t B, int32_t steps, const float* q_dev, int64_t* tokens_out_dev,
                      float* logits_trace_dev, void* stream) {
    WMAR_REQUIRE(g && sp && cond_dev && q_dev && tokens_out_dev, "generate: null argument");
    WMAR_REQUIRE(B >= 1 && B <= g->Bmax, "generate: batch %lld outside 1..%d", (long long)B, g->Bmax);
    WMAR_REQUIRE(steps >= 1 && steps <= g->Tmax, "generate: steps %d outside 1..block_size %d", steps, g->Tmax);
    WMAR_REQUIRE(!(sp->top_p >= 0) || sp->top_p <= 1.0, "`top_p` has to be a float > 0 and < 1, but is %f", sp->top_p);
    if (wm) {
        WMAR_REQUIRE(wm->table_dev && wm->vocab_size == g->V, "generate: watermark vocab mismatch");
        WMAR_REQUIRE(wm->context_size >= 0 && wm->context_size <= 3, "context size unsupported");
    }
    hipStream_t st = (hipStream_t)stream;
    const long long pstride = g->Tmax + 1;
    // past[b][0] = conditioning token; position and step counters to 0
    WMAR_HIP_CHECK(hipMemcpy2DAsync(g->past, pstride * 8, cond_dev, 8, 8, (size_t)B, hipMemcpyDeviceToDevice, st));
    WMAR_HIP_CHECK(hipMemsetAsync(g->pos_dev, 0, 8, st));

    SampArgs a{};
    if (wm) {
        a.wm.table = wm->table_dev; a.wm.n_rows = wm->n_rows; a.wm.row_words = (wm->vocab_size + 31) / 32;
        a.wm.seed_mode = wm->seed_strategy; a.wm.h = wm->context_size; a.wm.S = wm->spatial_dim;
        a.wm.delta = wm->delta; a.wm.enabled = 1;
    }
    a.logits = g->logits; a.V = g->V; a.past = g->past; a.past_stride = pstride;
    a.t_dev = g->step_dev;      // filled below: current length = step + 1
    a.temperature = sp->temperature; a.top_k = sp->top_k; a.use_top_p = sp->top_p >= 0;
    a.top_p_thr = (float)(1.0 - sp->top_p);
    a.q = q_dev; a.q_step_stride = (long long)B * g->V; a.step_dev = g->step_dev;
    a.scratch = g->scratch; a.tok_out = (long long*)tokens_out_dev; a.tok_out_stride = steps;
    a.past_append = g->past; a.trace = logits_trace_dev; a.B = B;
    // sampler reads its length from pos_dev+... : length of past at step n is n+1 == pos+1.
    // k_sample_fused takes t from *t_dev, so point it at a dedicated counter kept at pos+1.
    int* len_dev = g->pos_dev + 2;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, len_dev, 1);
    a.t_dev = len_dev;

    StepIO io{g->past, pstride, 1, g->logits};
    auto one_step = [&](hipStream_t s) -> int {
        int rc = enqueue_step(g, B, io, s);
        if (rc) return rc;
        g->span_begin(WMAR_T_SAMPLE, s);
        rc = launch_sample_fused(a, s);
        g->span_end(s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_advance3, dim3(1), dim3(1), 0, s, g->pos_dev);
        return launch_status("k_advance3");
    };

    if (g->timing) WMAR_HIP_CHECK(hipEventRecord(g->ev0, st));
    if (sp->use_graph) {
        // The captured step only depends on pointers and scalars: reuse the instantiated graph
        // when a call repeats them (bench / harness loops), re-capture otherwise.
        unsigned long long key[12] = {(unsigned long long)B, (unsigned long long)steps, (unsigned long long)(uintptr_t)q_dev,
                                      (unsigned long long)(uintptr_t)tokens_out_dev, (unsigned long long)(uintptr_t)logits_trace_dev,
                                      wm ? (unsigned long long)(uintptr_t)wm->table_dev : 0ull,
                                      wm ? (unsigned long long)wm->n_rows : 0ull,
                                      wm ? ((unsigned long long)wm->seed_strategy << 40) | ((unsigned long long)wm->context_size << 20) | (unsigned long long)wm->spatial_dim : 0ull,
                                      0ull, 0ull, (unsigned long long)sp->top_k, 0ull};
        float fd = wm ? wm->delta : 0.f, ft = sp->temperature;
        memcpy(&key[8], &fd, 4); memcpy(&key[9], &ft, 4); memcpy(&key[11], &sp->top_p, 8);
        if (g->exec && memcmp(key, g->graph_key, sizeof(key)) != 0) g->drop_graph();
        if (!g->exec) {
            WMAR_HIP_CHECK(hipStreamBeginCapture(g->cap_stream, hipStreamCaptureModeThreadLocal));
            int rc = one_step(g->cap_stream);
            hipError_t e = hipStreamEndCapture(g->cap_stream, &g->graph);
            if (rc) { g->drop_graph(); return rc; }
            if (e != hipSuccess) { g->drop_graph(); set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return WMAR_EHIP; }
            e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
            if (e != hipSuccess) { g->drop_graph(); set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return WMAR_EHIP; }
            memcpy(g->graph_key, key, sizeof(key));
        }
        hipError_t e = hipSuccess;
        for (int n = 0; n < steps && e == hipSuccess; ++n) e = hipGraphLaunch(g->exec, st);
        if (e == hipSuccess) e = hipEventRecord(g->ev1, st);   // also marks "replays finished" for drop_graph()
        if (e == hipSuccess) { g->pending = true; }
        if (e != hipSuccess) { set_error("graph replay failed: %s", hipGetErrorString(e)); return WMAR_EHIP; }
        // asynchronous: the caller's stream orders everything after the replays
        if (g->timing) WMAR_HIP_CHECK(hipEventSynchronize(g->ev1));
    } else {
        g->span_on = g->timing != 0;
        int rc = WMAR_OK;
        for (int n = 0; n < steps && rc == WMAR_OK; ++n) rc = one_step(st);
        g->span_on = false;
        if (rc) return rc;
        if (g->timing) {
            WMAR_HIP_CHECK(hipEventRecord(g->ev1, st));
            WMAR_HIP_CHECK(hipStreamSynchronize(st));
            g->span_collect();
        }
    }
    if (g->timing) {
        float ms = 0;
        WMAR_HIP_CHECK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
        g->step_ms = ms / steps;
    }
    return WMAR_OK;
}

}  // extern "C"
