// Gumbel-key ("Aaronson") sampler and detector score -- SURVEY.md section 8a row G1.
//
// Reference: wmar_audio/watermark/engine.py:29-75 (gumbel_sample) and :123-134
// (gumbel_score_tok).  The reference's image code never calls them: RAR + Gumbel key is an
// extension whose semantics follow that file with a fixed key (ngram = 0, engine.py:17-18).
//
// One workgroup per row, the row in registers (V <= 16384).  Every sum is the order-independent
// fixed-point sum of include/wmar_math.h and the two order statistics (top-p cut, k-th largest)
// are bisections over a composite key (probability bits, then lower index first), so a row's
// result does not depend on how it is laid over the lanes (the CPU checker restates the same
// arithmetic and must agree bit for bit).  Compile with -ffp-contract=off.
#include "sampler.h"
#include "../../include/wmar_math.h"

namespace wmar {

constexpr int GUM_THREADS = 1024;
constexpr int GUM_WAVES = GUM_THREADS / 64;
constexpr int GUM_EPT = 16;
constexpr int GUM_IDX_BITS = 14;
constexpr float GUM_UNDERFLOW = -103.972077f;   // log(2^-150): below this fp32 pow(rs, 1/p) is exactly 0

__device__ __forceinline__ unsigned long long gum_block_sum(unsigned long long v, unsigned long long* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long s = 0;
    for (int i = 0; i < GUM_WAVES; ++i) s += red[i];
    return s;
}

__device__ __forceinline__ unsigned long long gum_ckey(float p, int v) {
    return ((unsigned long long)wmar_f32_key(p) << GUM_IDX_BITS) | (unsigned long long)((1 << GUM_IDX_BITS) - 1 - v);
}

__global__ __launch_bounds__(GUM_THREADS) void k_gumbel_sample(GumbelArgs a) {
    __shared__ unsigned long long red[GUM_WAVES];
    __shared__ float red_f[GUM_WAVES];
    __shared__ unsigned long long red_k[GUM_WAVES];
    const long long b = blockIdx.x;
    const int V = (int)a.V;
    const int tid = threadIdx.x;
    const long long step = a.step_dev ? (long long)*a.step_dev : 0;
    const long long t = a.t_dev ? (long long)*a.t_dev : 0;
    const float* lg = a.logits + b * a.V;
    const float* ul = a.logits_uncond ? a.logits_uncond + b * a.V : nullptr;
    const float cfg = ul ? a.cfg_scale[step] : 0.f;
    const float* lr = a.log_rs + b * a.key_row_stride;

    float x[GUM_EPT];
    const bool sampling = a.use_sampling && a.temp > 0.0f;
#pragma unroll
    for (int i = 0; i < GUM_EPT; ++i) {
        const int v = tid + i * GUM_THREADS;
        float xv = -INFINITY;
        if (v < V) {
            xv = lg[v];
            if (ul) { const float u = ul[v]; const float dlt = xv - u; const float sc = dlt * cfg; xv = u + sc; }
            if (sampling) xv = xv / a.temp;
        }
        x[i] = xv;
    }
    // row maximum (and, without sampling, its first index: torch.argmax)
    unsigned long long best = 0;
#pragma unroll
    for (int i = 0; i < GUM_EPT; ++i) {
        const int v = tid + i * GUM_THREADS;
        if (v < V) best = max(best, gum_ckey(x[i], v));
    }
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned long long)__shfl_xor((long long)best, o));
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    best = 0;
    for (int i = 0; i < GUM_WAVES; ++i) best = max(best, red[i]);
    __syncthreads();
    long long token;
    if (!sampling) {
        token = (1 << GUM_IDX_BITS) - 1 - (long long)(best & ((1u << GUM_IDX_BITS) - 1));
    } else {
        const float m = wmar_key_f32((uint32_t)(best >> GUM_IDX_BITS));
        float p[GUM_EPT];
        unsigned long long z = 0;
#pragma unroll
        for (int i = 0; i < GUM_EPT; ++i) {
            const int v = tid + i * GUM_THREADS;
            p[i] = v < V ? wmar_expf(x[i] - m) : 0.f;
            z += wmar_fx(p[i]);
        }
        const float Z = wmar_fx_to_f32(gum_block_sum(z, red));
#pragma unroll
        for (int i = 0; i < GUM_EPT; ++i) p[i] = p[i] / Z;

        const bool by_rank = a.top_p > 0.0f;      // ties of the race resolve in sorted order (engine.py:69-74)
        if (a.top_p > 0.0f) {
            // smallest composite key K with mass{ckey > K} <= top_p: everything from K upwards is kept
            unsigned long long lo = 0, hi = (1ull << (32 + GUM_IDX_BITS)) - 1;
            while (lo < hi) {
                const unsigned long long mid = lo + ((hi - lo) >> 1);
                unsigned long long ms = 0;
#pragma unroll
                for (int i = 0; i < GUM_EPT; ++i) {
                    const int v = tid + i * GUM_THREADS;
                    if (v < V && gum_ckey(p[i], v) > mid) ms += wmar_fx(p[i]);
                }
                ms = gum_block_sum(ms, red);
                if (wmar_fx_to_f32(ms) > a.top_p) lo = mid + 1; else hi = mid;
            }
            unsigned long long ks = 0;
#pragma unroll
            for (int i = 0; i < GUM_EPT; ++i) {
                const int v = tid + i * GUM_THREADS;
                if (!(v < V && gum_ckey(p[i], v) >= lo)) p[i] = 0.f;
                ks += wmar_fx(p[i]);
            }
            const float S = wmar_fx_to_f32(gum_block_sum(ks, red));
#pragma unroll
            for (int i = 0; i < GUM_EPT; ++i) p[i] = p[i] / S;
        } else if (a.top_k > 0) {
            const unsigned long long kk = (unsigned long long)min((long long)a.top_k, (long long)V);
            // largest composite key K with count{ckey >= K} >= k
            unsigned long long lo = 0, hi = (1ull << (32 + GUM_IDX_BITS)) - 1;
            while (lo < hi) {
                const unsigned long long mid = lo + ((hi - lo + 1) >> 1);
                unsigned long long c = 0;
#pragma unroll
                for (int i = 0; i < GUM_EPT; ++i) {
                    const int v = tid + i * GUM_THREADS;
                    if (v < V && gum_ckey(p[i], v) >= mid) ++c;
                }
                c = gum_block_sum(c, red);
                if (c >= kk) lo = mid; else hi = mid - 1;
            }
            unsigned long long ks = 0;
#pragma unroll
            for (int i = 0; i < GUM_EPT; ++i) {
                const int v = tid + i * GUM_THREADS;
                if (v < V) {
                    if (!(gum_ckey(p[i], v) >= lo)) p[i] = 1e-6f;
                    ks += wmar_fx(p[i]);
                }
            }
            const float S = wmar_fx_to_f32(gum_block_sum(ks, red));
#pragma unroll
            for (int i = 0; i < GUM_EPT; ++i) p[i] = p[i] / S;
        }
        // the race
        float bs = -INFINITY;
        unsigned long long bk = 0;      // tie order: larger is earlier
        bool have = false;
#pragma unroll
        for (int i = 0; i < GUM_EPT; ++i) {
            const int v = tid + i * GUM_THREADS;
            if (v < V) {
                const float e = 1.0f / p[i];
                float s = lr[v] * e;
                if (!(s >= GUM_UNDERFLOW)) s = -INFINITY;
                const unsigned long long k = by_rank ? gum_ckey(p[i], v)
                                                     : (unsigned long long)((1 << GUM_IDX_BITS) - 1 - v);
                if (!have || s > bs || (s == bs && k > bk)) { bs = s; bk = k; have = true; }
            }
        }
        if (!have) { bs = -INFINITY; bk = 0; }
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o);
            const unsigned long long ok = (unsigned long long)__shfl_xor((long long)bk, o);
            if (os > bs || (os == bs && ok > bk)) { bs = os; bk = ok; }
        }
        if ((tid & 63) == 0) { red_f[tid >> 6] = bs; red_k[tid >> 6] = bk; }
        __syncthreads();
        for (int i = 0; i < GUM_WAVES; ++i)
            if (red_f[i] > bs || (red_f[i] == bs && red_k[i] > bk)) { bs = red_f[i]; bk = red_k[i]; }
        token = (1 << GUM_IDX_BITS) - 1 - (long long)(bk & ((1u << GUM_IDX_BITS) - 1));
    }
    if (tid == 0) {
        a.tok_out[b * a.tok_out_stride + step] = token;
        if (a.past_append) a.past_append[b * a.past_stride + t] = token;
    }
}

__global__ void k_gumbel_score(const long long* tokens, long long n, long long L, long long V, const float* key,
                               long long key_row_stride, long long* out_i, float* out_f, int* bad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long tk = tokens[i];
    float s = 0.f;
    if (tk < 0 || tk >= V) *bad = 1;
    else s = key[(i / L) * key_row_stride + tk];
    if (out_f) out_f[i] = s;
    if (out_i) out_i[i] = (s == INFINITY) ? (long long)0x8000000000000000ull : (long long)s;   // torch's float -> int64 of inf
}

int launch_gumbel_sample(const GumbelArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_gumbel_sample, dim3((unsigned)a.B), dim3(GUM_THREADS), 0, st, a);
    return launch_status("k_gumbel_sample");
}

}  // namespace wmar

using namespace wmar;

extern "C" {

int wmar_gumbel_sample(const float* logits_dev, int64_t B, int64_t V, const float* log_rs_dev, int64_t key_row_stride,
                       int32_t use_sampling, float temp, float top_p, int32_t top_k, int64_t* tok_out_dev, void* stream) {
    WMAR_REQUIRE(logits_dev && log_rs_dev && tok_out_dev, "gumbel_sample: null argument");
    WMAR_REQUIRE(B >= 0 && V >= 1 && V <= GUM_EPT * GUM_THREADS, "gumbel_sample: vocabulary %lld outside 1..%d", (long long)V,
                 GUM_EPT * GUM_THREADS);
    if (B == 0) return WMAR_OK;
    GumbelArgs a{};
    a.logits = logits_dev; a.V = V; a.B = B; a.log_rs = log_rs_dev; a.key_row_stride = key_row_stride;
    a.use_sampling = use_sampling; a.temp = temp; a.top_p = top_p; a.top_k = top_k;
    a.tok_out = (long long*)tok_out_dev; a.tok_out_stride = 1;
    return launch_gumbel_sample(a, (hipStream_t)stream);
}

int wmar_gumbel_score(const int64_t* tokens_dev, int64_t B, int64_t L, int64_t V, const float* score_key_dev,
                      int64_t key_row_stride, int64_t* scores_i64_dev, float* scores_f32_dev, void* stream) {
    WMAR_REQUIRE(tokens_dev && score_key_dev, "gumbel_score: null argument");
    WMAR_REQUIRE(B >= 0 && L >= 1 && V >= 1, "gumbel_score: bad shape");
    if (B == 0) return WMAR_OK;
    hipStream_t st = (hipStream_t)stream;
    int* bad = nullptr;
    WMAR_HIP_CHECK(hipMalloc(&bad, sizeof(int)));
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), st);
    const long long n = B * L;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gumbel_score, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const long long*)tokens_dev, n,
                           (long long)L, (long long)V, score_key_dev, (long long)key_row_stride, (long long*)scores_i64_dev,
                           scores_f32_dev, bad);
        e = hipGetLastError();
    }
    int hbad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(bad);
    if (e != hipSuccess) { set_error("gumbel_score: %s", hipGetErrorString(e)); return WMAR_EHIP; }
    WMAR_REQUIRE(!hbad, "gumbel_score: token id outside [0, %lld)", (long long)V);
    return WMAR_OK;
}

}  // extern "C"
