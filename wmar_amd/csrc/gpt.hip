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
constexpr int STAT_CHUNKS_MAX = 16;
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

// out[n] = (bias ? bias[n] : 0) + sum_k W[n][k] * beta[k]   (one wave per output row)
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

template <bool EMBED>
__global__ __launch_bounds__(256) void k_resid_stats(ResidArgs a) {
    __shared__ double red[4][32][2];
    const int c = blockIdx.x / a.MT, mt = blockIdx.x % a.MT;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kb0 = (int)((long long)c * a.KB / a.n_chunks), kb1 = (int)((long long)(c + 1) * a.KB / a.n_chunks);
    const int m = mt * 32 + (lane & 31), half = lane >> 5;
    double s = 0.0, ss = 0.0;
    const float* erow = nullptr;
    const float* prow = nullptr;
    if (EMBED) {
        int pos = *a.pos_dev;
        long long tk = (m < a.B) ? a.tok[(long long)m * a.tok_stride + (a.tok_use_pos ? pos : 0)] : 0;
        erow = a.tok_emb + tk * a.K;
        prow = a.pos_emb + (long long)pos * a.K;
    }
    for (int kb = kb0 + w; kb < kb1; kb += 4) {
        const long long idx = ((long long)kb * a.MT + mt) * 64 + lane;
        float4 v;
        if (EMBED) {
            const int k = kb * 8 + 4 * half;
            float4 e = *(const float4*)(erow + k);
            float4 p = *(const float4*)(prow + k);
            v = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
        } else {
            v = a.x[idx];
            float4 bb = *(const float4*)(a.bias + kb * 8 + 4 * half);
            float4 acc = a.slabs[idx];
            for (int sidx = 1; sidx < a.S; ++sidx) {
                float4 t = a.slabs[(long long)sidx * a.slab_stride + idx];
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
            v.x += bb.x + acc.x; v.y += bb.y + acc.y; v.z += bb.z + acc.z; v.w += bb.w + acc.w;
        }
        a.x[idx] = v;
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
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

// ------------------------------------------------------------------------- skinny GEMM
enum { EPI_PACKED = 0, EPI_GELU = 1, EPI_QKV = 2, EPI_LOGITS = 3 };

struct GemmArgs {
    const float4* Wp;          // [NT][KB][64]
    const float4* Xp;          // [KB][MT][64]
    const float* bias;         // [N] or null
    int KB, NT, MT, S;
    // fused LayerNorm on the B operand
    const double* stats; int n_chunks; int K;
    // epilogues
    float4* out_packed; long long slab_stride;          // EPI_PACKED (slab s) / EPI_GELU
    float* qbuf; float* kcache; float* vcache;          // EPI_QKV
    const int* pos_dev; int D, H, hd, Tmax;
    float* logits; int V;                               // EPI_LOGITS
    int B;
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
    const int kb0 = (int)((long long)sl * a.KB / slices), kb1 = (int)((long long)(sl + 1) * a.KB / slices);
    const int half = lane >> 5;

    f32x16 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float mu[MTW], rstd[MTW];
    if (LN) {
        const int Mpad = a.MT * 32;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int m = (mt0 + i) * 32 + (lane & 31);
            double sm = 0, sq = 0;
            for (int c = 0; c < a.n_chunks; ++c) {
                const double* p = a.stats + ((long long)c * Mpad + m) * 2;
                sm += p[0]; sq += p[1];
            }
            double mean = sm / a.K;
            double var = sq / a.K - mean * mean;
            mu[i] = (float)mean;
            rstd[i] = (float)(1.0 / sqrt(var + 1e-5));
        }
    }

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
#define WMAR_MMA1(WV, XV)                                                                       \
    {                                                                                           \
        float4 xv[MTW];                                                                         \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i) {                                      \
            xv[i] = XV[i];                                                                      \
            if (LN) {                                                                           \
                xv[i].x = (xv[i].x - mu[i]) * rstd[i]; xv[i].y = (xv[i].y - mu[i]) * rstd[i];   \
                xv[i].z = (xv[i].z - mu[i]) * rstd[i]; xv[i].w = (xv[i].w - mu[i]) * rstd[i];   \
            }                                                                                   \
        }                                                                                       \
        if (ABL == 3) {                                                                         \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                    \
                acc[i][0] += WV.x * xv[i].x + WV.y * xv[i].y + WV.z * xv[i].z + WV.w * xv[i].w; \
        } else {                                                                                \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV.x, xv[i].x, acc[i], 0, 0, 0);      \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV.y, xv[i].y, acc[i], 0, 0, 0);      \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV.z, xv[i].z, acc[i], 0, 0, 0);      \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                        \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV.w, xv[i].w, acc[i], 0, 0, 0);      \
        }                                                                                       \
    }
#define WMAR_MMA(WBUF, XBUF) _Pragma("unroll") for (int u = 0; u < U; ++u) WMAR_MMA1(WBUF[u], XBUF[u])
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
    }
    for (; kb < kb1; ++kb) {  // tail (slices that are not a multiple of 2U blocks)
        float4 wv = ld_nt(Wp + (long long)kb * 64);
        float4 xt[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) xt[i] = Xp[(long long)kb * xstep + i * 64];
        WMAR_MMA1(wv, xt)
    }
#undef WMAR_MMA1
#undef WMAR_LOAD
#undef WMAR_MMA

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
}

// --------------------------------------------------------------------- decode attention
// One wave per (sequence, head).  K/V rows are hd floats; LPR = hd/4 lanes cover a row with
// float4s and RPI = 64/LPR rows are read per wave-wide load (1 KiB, coalesced).
struct AttnArgs {
    const float* qbuf;   // [M][D]
    const float* kcache; // [B][H][Tmax][hd] (this layer)
    const float* vcache;
    float4* y;           // packed [KB][MT][64]
    const int* pos_dev;
    int D, H, Tmax, MT;
    float scale;
};

template <int HD>
__global__ __launch_bounds__(64) void k_attn_decode(AttnArgs a) {
    constexpr int LPR = HD / 4, RPI = 64 / LPR;
    extern __shared__ float sc[];  // [Tmax] scores / probabilities
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int lane = threadIdx.x;
    const int T = *a.pos_dev + 1;
    const int sub = lane % LPR, rsel = lane / LPR;
    const float4 q = *(const float4*)(a.qbuf + (long long)b * a.D + h * HD + sub * 4);
    const float* K = a.kcache + ((long long)b * a.H + h) * a.Tmax * HD;
    const float* Vv = a.vcache + ((long long)b * a.H + h) * a.Tmax * HD;

    float mx = -INFINITY;
    for (int t0 = 0; t0 < T; t0 += RPI * 4) {
        float4 kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int t = t0 + u * RPI + rsel;
            kv[u] = (t < T) ? *(const float4*)(K + (long long)t * HD + sub * 4) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float p = kv[u].x * q.x + kv[u].y * q.y + kv[u].z * q.z + kv[u].w * q.w;
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) p += __shfl_xor(p, o);
            p *= a.scale;
            int t = t0 + u * RPI + rsel;
            if (t < T) {
                if (sub == 0) sc[t] = p;
                mx = fmaxf(mx, p);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __syncthreads();
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
        float e = __expf(sc[t] - mx);
        sc[t] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __syncthreads();
    const float inv = 1.0f / sum;

    float4 acc = make_float4(0, 0, 0, 0);
    for (int t0 = 0; t0 < T; t0 += RPI * 4) {
        float4 vv[4];
        float pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int t = t0 + u * RPI + rsel;
            bool ok = t < T;
            vv[u] = ok ? *(const float4*)(Vv + (long long)t * HD + sub * 4) : make_float4(0, 0, 0, 0);
            pp[u] = ok ? sc[t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc.x += pp[u] * vv[u].x; acc.y += pp[u] * vv[u].y; acc.z += pp[u] * vv[u].z; acc.w += pp[u] * vv[u].w;
        }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
        acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (rsel == 0) {
        // y[b][h*HD + sub*4 .. +3] into the packed activation layout
        const int k = h * HD + sub * 4;
        const int kb = k >> 3, hf = (k >> 2) & 1;
        const int mt = b >> 5;
        a.y[((long long)kb * a.MT + mt) * 64 + (b & 31) + 32 * hf] =
            make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

}  // namespace wmar

using namespace wmar;

// ------------------------------------------------------------------------------ engine
struct LayerW {
    float4 *wqkv, *wproj, *wfc1, *wfc2;
    float *bqkv, *bproj, *bfc1, *bfc2;
};

struct wmar_gpt {
    wmar_gpt_config cfg{};
    int D = 0, H = 0, hd = 0, V = 0, L = 0, Tmax = 0, Bmax = 0, MTmax = 0;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    std::vector<LayerW> layers;
    float *tok_emb = nullptr, *pos_emb = nullptr, *bhead = nullptr;
    float4* whead = nullptr;
    // workspaces
    float4 *x = nullptr, *y = nullptr, *hbuf = nullptr, *slabs = nullptr;
    float* qbuf = nullptr;
    double* stats = nullptr;
    float *kcache = nullptr, *vcache = nullptr;
    float *logits = nullptr, *scratch = nullptr;
    long long* past = nullptr;  // [Bmax][Tmax+1]
    int *pos_dev = nullptr, *step_dev = nullptr;
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int timing = 0;
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
int gemm_split(GemmArgs a, int* S_out, hipStream_t st) {
    constexpr int NW = 4;
    if (a.MT % 2 == 0) {
        a.S = pick_split(a.NT * (a.MT / 2), a.KB, NW);
        *S_out = a.S;
        return launch_gemm<2, NW, EPI_PACKED, false>(a, st);
    }
    a.S = pick_split(a.NT * a.MT, a.KB, NW);
    *S_out = a.S;
    return launch_gemm<1, NW, EPI_PACKED, false>(a, st);
}

int stat_chunks(int KB) {
    int c = KB / 24;
    if (c < 1) c = 1;
    if (c > STAT_CHUNKS_MAX) c = STAT_CHUNKS_MAX;
    return c;
}

struct StepIO {
    const long long* tok;  // row m's token: tok[m*stride + (use_pos ? pos : 0)]
    long long tok_stride;
    int tok_use_pos;
    float* logits;
};

int enqueue_step(wmar_gpt* g, int64_t B, const StepIO& io, hipStream_t st) {
    const int MT = mt_for(B);
    const int D = g->D, KBD = D / 8, KBF = 4 * D / 8;
    const int nch = stat_chunks(KBD);
    const long long act = (long long)KBD * MT * 64;  // float4 units of one [M][D] packed activation
    int rc;

    ResidArgs r{};
    r.x = g->x; r.stats = g->stats; r.KB = KBD; r.MT = MT; r.n_chunks = nch;
    r.tok_emb = g->tok_emb; r.pos_emb = g->pos_emb; r.tok = io.tok; r.tok_stride = io.tok_stride;
    r.tok_use_pos = io.tok_use_pos; r.pos_dev = g->pos_dev; r.B = (int)B; r.K = D;
    g->span_begin(WMAR_T_EMBED, st);
    hipLaunchKernelGGL(k_resid_stats<true>, dim3(nch * MT), dim3(256), 0, st, r);
    g->span_end(st);
    if ((rc = launch_status("k_resid_stats<embed>"))) return rc;

    int S_prev = 0;
    const float* bias_prev = nullptr;
    for (int l = 0; l <= g->L; ++l) {
        if (l > 0) {  // fold the previous layer's FC2 partial sums into the residual stream
            r.slabs = g->slabs; r.slab_stride = act; r.S = S_prev; r.bias = bias_prev;
            g->span_begin(WMAR_T_RESID, st);
            hipLaunchKernelGGL(k_resid_stats<false>, dim3(nch * MT), dim3(256), 0, st, r);
            g->span_end(st);
            if ((rc = launch_status("k_resid_stats"))) return rc;
        }
        if (l == g->L) break;
        const LayerW& w = g->layers[l];
        GemmArgs a{};
        a.MT = MT; a.B = (int)B; a.stats = g->stats; a.n_chunks = nch; a.K = D;
        a.pos_dev = g->pos_dev; a.D = D; a.H = g->H; a.hd = g->hd; a.Tmax = g->Tmax;
        // LN1 -> QKV (+bias) -> q buffer and KV cache
        a.Wp = w.wqkv; a.Xp = g->x; a.bias = w.bqkv; a.KB = KBD; a.NT = 3 * D / 32;
        a.qbuf = g->qbuf;
        const long long lstride = (long long)g->Bmax * g->H * g->Tmax * g->hd;
        a.kcache = g->kcache + l * lstride; a.vcache = g->vcache + l * lstride;
        g->span_begin(WMAR_T_QKV, st);
        rc = gemm_dispatch<EPI_QKV, true>(a, false, st);
        g->span_end(st);
        if (rc) return rc;
        // attention
        AttnArgs t{};
        t.qbuf = g->qbuf; t.kcache = a.kcache; t.vcache = a.vcache; t.y = g->y; t.pos_dev = g->pos_dev;
        t.D = D; t.H = g->H; t.Tmax = g->Tmax; t.MT = MT; t.scale = 1.0f / sqrtf((float)g->hd);
        const size_t lds = (size_t)g->Tmax * sizeof(float);
        const dim3 grid((unsigned)(B * g->H));
        g->span_begin(WMAR_T_ATTN, st);
        if (g->hd == 64) hipLaunchKernelGGL(k_attn_decode<64>, grid, dim3(64), lds, st, t);
        else if (g->hd == 32) hipLaunchKernelGGL(k_attn_decode<32>, grid, dim3(64), lds, st, t);
        else hipLaunchKernelGGL(k_attn_decode<128>, grid, dim3(64), lds, st, t);
        g->span_end(st);
        if ((rc = launch_status("k_attn_decode"))) return rc;
        // proj (split-K partial slabs; bias + residual folded by the next k_resid_stats)
        GemmArgs p = a;
        p.Wp = w.wproj; p.Xp = g->y; p.bias = nullptr; p.KB = KBD; p.NT = D / 32;
        p.out_packed = g->slabs; p.slab_stride = act;
        int S_proj = 1;
        g->span_begin(WMAR_T_PROJ, st);
        rc = gemm_split(p, &S_proj, st);
        g->span_end(st);
        if (rc) return rc;
        r.slabs = g->slabs; r.slab_stride = act; r.S = S_proj; r.bias = w.bproj;
        g->span_begin(WMAR_T_RESID, st);
        hipLaunchKernelGGL(k_resid_stats<false>, dim3(nch * MT), dim3(256), 0, st, r);
        g->span_end(st);
        if ((rc = launch_status("k_resid_stats"))) return rc;
        // LN2 -> FC1 (+bias, GELU) -> packed hidden
        GemmArgs f = a;
        f.Wp = w.wfc1; f.Xp = g->x; f.bias = w.bfc1; f.KB = KBD; f.NT = 4 * D / 32;
        f.out_packed = g->hbuf; f.slab_stride = 0;
        g->span_begin(WMAR_T_FC1, st);
        rc = gemm_dispatch<EPI_GELU, true>(f, false, st);
        g->span_end(st);
        if (rc) return rc;
        // FC2 (split-K partial slabs)
        GemmArgs q = a;
        q.Wp = w.wfc2; q.Xp = g->hbuf; q.bias = nullptr; q.KB = KBF; q.NT = D / 32;
        q.out_packed = g->slabs; q.slab_stride = act;
        g->span_begin(WMAR_T_FC2, st);
        rc = gemm_split(q, &S_prev, st);
        g->span_end(st);
        if (rc) return rc;
        bias_prev = w.bfc2;
    }
    // ln_f -> head
    GemmArgs hsd{};
    hsd.MT = MT; hsd.B = (int)B; hsd.stats = g->stats; hsd.n_chunks = nch; hsd.K = D;
    hsd.Wp = g->whead; hsd.Xp = g->x; hsd.KB = KBD; hsd.NT = g->V / 32;
    hsd.bias = g->bhead; hsd.logits = io.logits; hsd.V = g->V;
    g->span_begin(WMAR_T_HEAD, st);
    rc = gemm_dispatch<EPI_LOGITS, true>(hsd, false, st);
    g->span_end(st);
    return rc;
}

}  // namespace

extern "C" {

int wmar_gpt_create(const wmar_gpt_config* cfg, const char* const* names, const void* const* tensors_dev,
                    int32_t n_tensors, void* stream, wmar_gpt** out) {
    WMAR_REQUIRE(cfg && names && tensors_dev && out, "gpt_create: null argument");
    WMAR_REQUIRE(cfg->n_embd % cfg->n_head == 0, "n_embd %% n_head != 0");
    const int D = cfg->n_embd, H = cfg->n_head, hd = D / H, V = cfg->vocab_size, L = cfg->n_layer;
    WMAR_REQUIRE(D % 32 == 0 && V % 32 == 0, "n_embd and vocab_size must be multiples of 32 (got %d, %d)", D, V);
    WMAR_REQUIRE(hd == 32 || hd == 64 || hd == 128, "head_dim %d unsupported (32, 64, 128)", hd);
    WMAR_REQUIRE(cfg->max_batch >= 1 && cfg->max_batch <= 128, "max_batch must be in 1..128");
    WMAR_REQUIRE(cfg->block_size >= 1 && L >= 1, "bad block_size / n_layer");
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) tm.m[names[i]] = tensors_dev[i];
    hipStream_t st = (hipStream_t)stream;
    auto* g = new wmar_gpt();
    g->cfg = *cfg; g->D = D; g->H = H; g->hd = hd; g->V = V; g->L = L; g->Tmax = cfg->block_size; g->Bmax = cfg->max_batch;
    g->MTmax = mt_for(cfg->max_batch);
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
        TRY(g->alloc(&w.wproj, (size_t)D * D / 4));
        TRY(pack(g, pw, w.wproj, D, D, 0, st));
        TRY(copy_vec(g, &w.bproj, pb, D, st));
        TRY(g->alloc(&w.wfc1, (size_t)4 * D * D / 4));
        TRY(pack(g, f1w, w.wfc1, 4 * D, D, 0, st, l2w));
        TRY(g->alloc(&w.bfc1, (size_t)4 * D));
        TRY(fold_bias(f1w, f1b, l2b, w.bfc1, 4 * D, D, st));
        TRY(g->alloc(&w.wfc2, (size_t)4 * D * D / 4));
        TRY(pack(g, f2w, w.wfc2, D, 4 * D, 0, st));
        TRY(copy_vec(g, &w.bfc2, f2b, D, st));
    }
    const size_t Mpad = (size_t)g->MTmax * 32;
    TRY(g->alloc(&g->x, Mpad * D / 4));
    TRY(g->alloc(&g->y, Mpad * D / 4));
    TRY(g->alloc(&g->hbuf, Mpad * 4 * D / 4));
    TRY(g->alloc(&g->slabs, (size_t)MAX_SLABS * Mpad * D / 4));
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
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        WMAR_HIP_CHECK(hipStreamBeginCapture(g->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = one_step(g->cap_stream);
        hipError_t e = hipStreamEndCapture(g->cap_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return WMAR_EHIP; }
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { (void)hipGraphDestroy(graph); set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return WMAR_EHIP; }
        for (int n = 0; n < steps; ++n) {
            e = hipGraphLaunch(exec, st);
            if (e != hipSuccess) break;
        }
        if (g->timing && e == hipSuccess) e = hipEventRecord(g->ev1, st);
        // the exec must outlive its launches
        hipError_t e2 = hipStreamSynchronize(st);
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess || e2 != hipSuccess) {
            set_error("graph replay failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
            return WMAR_EHIP;
        }
    } else {
        g->span_on = g->timing != 0;
        int rc = WMAR_OK;
        for (int n = 0; n < steps && rc == WMAR_OK; ++n) rc = one_step(st);
        g->span_on = false;
        if (rc) return rc;
        if (g->timing) WMAR_HIP_CHECK(hipEventRecord(g->ev1, st));
        WMAR_HIP_CHECK(hipStreamSynchronize(st));
        if (g->timing) g->span_collect();
    }
    if (g->timing) {
        float ms = 0;
        WMAR_HIP_CHECK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
        g->step_ms = ms / steps;
    }
    return WMAR_OK;
}

}  // extern "C"
