// Watermark kernels for gfx950: logit bias, fused bias+warp+sample, detector.
//
// All three are HBM/latency-bound integer-and-compare work: one workgroup per row,
// coalesced row reads, order-independent integer reductions (so results do not depend
// on wave scheduling), LDS histograms for the exact top-k / top-p boundaries.
// Compile with -ffp-contract=off: the arithmetic is the one pinned in include/wmar_math.h.
#include "sampler.h"
#include "../../include/wmar_math.h"

namespace wmar {

__device__ __forceinline__ const uint32_t* wm_row(const WmDev& wm, const long long* past, long long t) {
    if (!wm.enabled) return nullptr;
    long long r = ctx_row(past, t, wm.seed_mode, wm.h, wm.S);
    if (r < 0 || r >= wm.n_rows) return nullptr;
    return wm.table + r * wm.row_words;
}

// ------------------------------------------------------------------ logit processor
__global__ __launch_bounds__(256) void k_wm_bias(WmDev wm, float* logits, long long V, const long long* past,
                                                 long long t, long long past_stride) {
    const long long b = blockIdx.y;
    const uint32_t* row = wm_row(wm, past + b * past_stride, t);
    if (!row) return;
    float* lg = logits + b * V;
    const float d = wm.delta;
    for (long long w = blockIdx.x * blockDim.x + threadIdx.x; w * 32 < V; w += (long long)gridDim.x * blockDim.x) {
        uint32_t bits = row[w];
        if (!bits) continue;
        long long v0 = w * 32;
        for (int j = 0; j < 32 && v0 + j < V; ++j)
            if ((bits >> j) & 1u) lg[v0 + j] = lg[v0 + j] + d;
    }
}

// ---------------------------------------------------------------------- fused sampler
constexpr int SAMP_THREADS = 1024;
constexpr int SAMP_WAVES = SAMP_THREADS / 64;

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Block-wide order-independent reductions through LDS.  Every call site owns its slot `red` (SAMP_WAVES words nobody else
// writes), so ONE barrier per reduction is enough: no wave can still be reading the slot from an earlier use.
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* red) {
    v = wave_sum_u64(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) red[w] = v;
    __syncthreads();
    unsigned long long s = 0;
    for (int i = 0; i < SAMP_WAVES; ++i) s += red[i];
    return s;
}

__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, unsigned long long* red) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) red[w] = v;
    __syncthreads();
    uint32_t s = 0;
    for (int i = 0; i < SAMP_WAVES; ++i) s = max(s, (uint32_t)red[i]);
    return s;
}

// One wave scans a 256-bucket u64 histogram (every wave of the sampler does, redundantly: the result is wave-uniform and no
// broadcast through LDS -- two more workgroup barriers per pass -- is needed).
//  DESC_COUNT: from bucket 255 downwards, first bucket where the running count reaches `need`;
//              returns bucket, and *above = count strictly above it.
//  ASC_MASS:   from bucket 0 upwards, first bucket whose inclusive mass (base + ...) converts to
//              an fp32 value > thr; returns bucket (or -1 if none), *above = mass strictly below it.
template <bool ASC_MASS>
__device__ __forceinline__ int scan_hist(const unsigned long long* hist, unsigned long long need_or_base, float thr,
                                         unsigned long long* other) {
    const int l = threadIdx.x & 63;
    unsigned long long h0, h1, h2, h3;
    if (ASC_MASS) {
        h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
    } else {  // reversed order: lane l covers buckets 255-4l .. 252-4l
        h0 = hist[255 - 4 * l], h1 = hist[254 - 4 * l], h2 = hist[253 - 4 * l], h3 = hist[252 - 4 * l];
    }
    unsigned long long tot = h0 + h1 + h2 + h3;
    unsigned long long inc = tot;
    for (int o = 1; o < 64; o <<= 1) {
        unsigned long long n = __shfl_up(inc, o);
        if (l >= o) inc += n;
    }
    unsigned long long exc = inc - tot;  // sum of all earlier lanes' buckets
    unsigned long long c0 = exc + h0, c1 = c0 + h1, c2 = c1 + h2, c3 = c2 + h3;
    int hit = -1;
    unsigned long long before = 0;
    if (ASC_MASS) {
        const unsigned long long base = need_or_base;
        if (!(wmar_fx_to_f32(base + c0) <= thr)) { hit = 0; before = exc; }
        else if (!(wmar_fx_to_f32(base + c1) <= thr)) { hit = 1; before = c0; }
        else if (!(wmar_fx_to_f32(base + c2) <= thr)) { hit = 2; before = c1; }
        else if (!(wmar_fx_to_f32(base + c3) <= thr)) { hit = 3; before = c2; }
    } else {
        const unsigned long long need = need_or_base;
        if (c0 >= need) { hit = 0; before = exc; }
        else if (c1 >= need) { hit = 1; before = c0; }
        else if (c2 >= need) { hit = 2; before = c1; }
        else if (c3 >= need) { hit = 3; before = c2; }
    }
    unsigned long long m = __ballot(hit >= 0);
    if (m == 0) { *other = __shfl(inc, 0); return -1; }  // wave-uniform (lane 0's running sum, what wave 0 used to publish; no caller uses it)
    int first = __ffsll((long long)m) - 1;
    int hit_f = __shfl(hit, first);
    unsigned long long before_f = __shfl(before, first);
    *other = before_f;
    int bucket = 4 * first + hit_f;
    return ASC_MASS ? bucket : 255 - bucket;
}

// EPT > 0: the whole row lives in registers (EPT values per thread, V <= EPT*1024) and every pass
// after the first runs out of registers -- with the row in memory each of the ~13 passes is a chain
// of dependent L2 round trips (measured 75 us/step).  EPT == 0: generic path through `scratch`.
// PLAIN: no gather table, no guidance streams, no allow-list, no logits trace (the Taming / RAR-without-guidance call): every load of
// the first pass is then unconditional (clamped index) and issued before the first use -- under the per-element conditions of the
// general path hipcc waits for each load on its own (`s_waitcnt vmcnt(0)` behind every one: ~48 serialized L2 round trips per row).
template <int EPT, bool PLAIN = false>
__global__ __launch_bounds__(SAMP_THREADS) void k_sample_fused(SampArgs a) {
    // One histogram per radix pass (4 top-k + 4 top-p + 4 tie-break), zeroed once up front, and one reduction slot per call site:
    // a pass is count -> ONE barrier -> every wave scans the histogram itself (round 5; it was zero / barrier / count / barrier /
    // wave 0 scans / barrier / read / barrier: ~46 sixteen-wave barriers per row, now 13).
    constexpr int N_HIST = 12, N_RED = 5;
    __shared__ unsigned long long hist_all[N_HIST][256];
    __shared__ unsigned long long red_all[N_RED][SAMP_WAVES];
    __shared__ float red_f[SAMP_WAVES];
    __shared__ int red_i[SAMP_WAVES];
    // the noise row of rows that fit registers 16 values per thread (V <= 16384): requested in P0, parked here (64 KB: the
    // workgroup owns its CU) and read back for the final race -- held in registers through the passes it cost 30 spilled VGPRs
    // at the 128 the 1024-thread workgroup allows (round-3 review)
    __shared__ float q_lds[(EPT > 0 && EPT <= 16) ? EPT * SAMP_THREADS : 1];
    // survivors of top-k compacted to one per thread (round 5): every pass behind top-k then costs one element per thread instead
    // of a sweep over the EPT registers of which ~1.5 % are alive
    __shared__ float surv_x[EPT > 0 ? SAMP_THREADS : 1];
    __shared__ int surv_v[EPT > 0 ? SAMP_THREADS : 1];
    __shared__ int surv_n;
    if (threadIdx.x == 0) surv_n = 0;
    static_assert(sizeof(hist_all) + sizeof(red_all) + sizeof(q_lds) + 2 * SAMP_WAVES * 4 + 2 * SAMP_THREADS * 4 + 4 <= 160 * 1024,
                  "k_sample_fused: static LDS beyond the 160 KB of a gfx950 CU (the noise row alone is 64 KB: gfx950 only)");
    for (int i = threadIdx.x; i < N_HIST * 256; i += SAMP_THREADS) (&hist_all[0][0])[i] = 0;      // visible after the row-max barrier

    const long long b = blockIdx.x;
    const long long V = a.V;
    const int tid = threadIdx.x;
    const long long t = a.t_dev ? (long long)*a.t_dev : a.t_host;
    const long long step = a.step_dev ? (long long)*a.step_dev : 0;
    const long long* past = a.past ? a.past + b * a.past_stride : nullptr;
    const uint32_t* grow = (past || a.wm.seed_mode == WMAR_SEED_FIXED) ? wm_row(a.wm, past, t) : nullptr;
    const int* gth = a.gather;
    const long long Vs = gth ? a.Vsrc : V;          // width of the source rows
#define WMAR_SRC(v) (gth ? (long long)gth[v] : (long long)(v))
    const float* lg = a.logits + b * Vs;
    const float* ul = a.logits_uncond ? a.logits_uncond + b * Vs : nullptr;
    const float* il = a.logits_img ? a.logits_img + b * Vs : nullptr;
    const float cfg = (ul && !il) ? a.cfg_scale[step] : 0.f;
    const float* q = a.q + step * a.q_step_stride + b * Vs;
    float* x = a.scratch + b * V;
    float* trace = a.trace ? a.trace + (step * a.B + b) * Vs : nullptr;
    const float T = a.temperature;
    const float delta = a.wm.delta;
    float xr[EPT > 0 ? EPT : 1];

#define WMAR_FOR_ROW(BODY)                                                                   \
    if (EPT > 0) {                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < (EPT > 0 ? EPT : 1); ++i_) {                \
            const long long v = tid + (long long)i_ * SAMP_THREADS;                          \
            if (v < V) { const float xv = xr[i_]; BODY }                                     \
        }                                                                                    \
    } else {                                                                                 \
        for (long long v = tid; v < V; v += SAMP_THREADS) { const float xv = x[v]; BODY }    \
    }

    // P0: bias, temperature, row max.  The noise row is requested here as well when it fits beside the row (EPT <= 16): its
    // latency then hides under the passes instead of being paid in front of the final argmax.
    uint32_t kmax = 0;
    constexpr int QN = (EPT > 0 && EPT <= 16) ? EPT : 1;
    float qpre[QN];
    if (EPT > 0 && EPT <= 16) {
#pragma unroll
        for (int i = 0; i < QN; ++i) {
            const long long v = tid + (long long)i * SAMP_THREADS;
            if (PLAIN) qpre[i] = __builtin_nontemporal_load(q + (v < V ? v : V - 1));      // 1 GB of noise per generation, read once
            else qpre[i] = v < V ? q[WMAR_SRC(v)] : 1.f;
        }
    }
#define WMAR_PARK_Q()                                                                        \
    if (EPT > 0 && EPT <= 16) {                                                              \
        _Pragma("unroll") for (int i = 0; i < QN; ++i) q_lds[tid + i * SAMP_THREADS] = qpre[i]; \
    }
    if (EPT > 0 && PLAIN) {
        constexpr int CH = EPT > 8 ? 8 : (EPT > 0 ? EPT : 1);        // 8 logits + 8 key words in flight beside the 16 noise values: no spills at 128 VGPRs
        const uint32_t* gp = grow ? grow : reinterpret_cast<const uint32_t*>(lg);     // no key row: any valid words, never used
#pragma unroll
        for (int c0 = 0; c0 < (EPT > 0 ? EPT : 1); c0 += CH) {
            float lv[CH];
            uint32_t gw[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const long long v = tid + (long long)(c0 + j) * SAMP_THREADS;
                const long long vv = v < V ? v : V - 1;
                lv[j] = __builtin_nontemporal_load(lg + vv);
                gw[j] = gp[vv >> 5];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = c0 + j;
                const long long v = tid + (long long)i * SAMP_THREADS;
                if (v < V) {
                    float xv = lv[j];
                    if (grow && ((gw[j] >> (v & 31)) & 1u)) xv = xv + delta;
                    if (T != 1.0f) xv = xv / T;          // (x / 1 is x: ten instructions per value less)
                    xr[i] = xv;
                    if (a.scratch) x[v] = xv;
                    kmax = max(kmax, wmar_f32_key(xv));
                }
            }
        }
    } else if (EPT > 0) {
        // chunks of 16 values per thread: all loads of a chunk are issued before its arithmetic (one round trip per chunk),
        // without holding a second copy of a 64-value row in registers
        constexpr int CH = EPT > 16 ? 16 : (EPT > 0 ? EPT : 1);
#pragma unroll
        for (int c0 = 0; c0 < (EPT > 0 ? EPT : 1); c0 += CH) {
            float lv[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = c0 + j;
                const long long v = tid + (long long)i * SAMP_THREADS;
                const long long sv = v < V ? WMAR_SRC(v) : 0;
                lv[j] = v < V ? lg[sv] : 0.f;
                if (il && v < V) {
                    const float u = ul[sv], im = il[sv];
                    const float d1 = im - u; const float t1 = a.g_image * d1; const float s1 = u + t1;
                    const float d2 = lv[j] - im; const float t2 = a.g_text * d2; lv[j] = s1 + t2;
                } else if (ul && v < V) { const float u = ul[sv]; const float dlt = lv[j] - u; const float sc = dlt * cfg; lv[j] = u + sc; }
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = c0 + j;
                const long long v = tid + (long long)i * SAMP_THREADS;
                if (v < V) {
                    float xv = lv[j];
                    const long long sv = WMAR_SRC(v);
                    if (trace) trace[sv] = xv;
                    if (grow && ((grow[sv >> 5] >> (sv & 31)) & 1u)) xv = xv + delta;
                    if (a.allow && !((a.allow[sv >> 5] >> (sv & 31)) & 1u)) xv = -INFINITY;
                    xv = xv / T;
                    xr[i] = xv;                  // the row lives in registers; `scratch` gets a copy only when the caller asks for one
                    if (a.scratch) x[v] = xv;
                    kmax = max(kmax, wmar_f32_key(xv));
                }
            }
        }
    } else {
        for (long long v = tid; v < V; v += SAMP_THREADS) {
            const long long sv = WMAR_SRC(v);
            float xv = lg[sv];
            if (il) {
                const float u = ul[sv], im = il[sv];
                const float d1 = im - u; const float t1 = a.g_image * d1; const float s1 = u + t1;
                const float d2 = xv - im; const float t2 = a.g_text * d2; xv = s1 + t2;
            } else if (ul) { const float u = ul[sv]; const float dlt = xv - u; const float sc = dlt * cfg; xv = u + sc; }
            if (trace) trace[sv] = xv;
            if (grow && ((grow[sv >> 5] >> (sv & 31)) & 1u)) xv = xv + delta;
            if (a.allow && !((a.allow[sv >> 5] >> (sv & 31)) & 1u)) xv = -INFINITY;
            xv = xv / T;
            x[v] = xv;
            kmax = max(kmax, wmar_f32_key(xv));
        }
    }
    WMAR_PARK_Q()            // each thread reads back its own words: no barrier needed
#undef WMAR_PARK_Q
    kmax = block_max_u32(kmax, red_all[0]);
    const float m = wmar_key_f32(kmax);
    const uint32_t NEG_INF_KEY = 0x007fffffu;  // wmar_f32_key(-inf)

    // top-k: exact k-th largest key by 4 radix passes (count histograms)
    uint32_t thr_key = 0;
    unsigned long long n_alive = ~0ull;      // keys >= thr_key when top-k is on (an upper bound: -inf keys are dropped below)
    if (a.top_k > 0 && (long long)a.top_k < V) {
        uint32_t prefix = 0, mask = 0;
        unsigned long long need = (unsigned long long)a.top_k;
        for (int pass = 3; pass >= 0; --pass) {
            unsigned long long* hist = hist_all[3 - pass];
            if (pass == 3) {
                // The top byte (sign + 7 exponent bits) takes a handful of values.  Round 5: ONE de-duplication round per value --
                // the lanes that share the first active lane's digit are counted by ballot and added once, the others issue a
                // plain 32-bit LDS add (the LDS serialises equal addresses at a lane per cycle).  Rounds 1-4 looped the ballot
                // once per distinct digit: ~12 iterations x 16 values, half of the kernel's VALU instructions.
#define WMAR_TOPBYTE(V_, XV_)                                                                         \
                {                                                                                         \
                    const bool active = (V_) < V;                                                         \
                    const uint32_t d = active ? (wmar_f32_key(XV_) >> 24) : 0xffffffffu;                  \
                    const unsigned long long act = __ballot(active);                                      \
                    if (act) {                                                                            \
                        const int leader = __ffsll((long long)act) - 1;                                   \
                        const uint32_t dl = (uint32_t)__shfl((int)d, leader);                             \
                        const unsigned long long same = __ballot(d == dl);                                \
                        uint32_t* h32 = reinterpret_cast<uint32_t*>(hist);      /* counts < 2^32: low words */ \
                        if ((tid & 63) == leader) atomicAdd(&h32[2 * dl], (uint32_t)__popcll(same));      \
                        else if (active && d != dl) atomicAdd(&h32[2 * d], 1u);                           \
                    }                                                                                     \
                }
                if (EPT > 0) {       // (direct register access: selecting xr[i_] by a run-time i_ was an EPT^2 chain of v_cndmask)
#pragma unroll
                    for (int i_ = 0; i_ < (EPT > 0 ? EPT : 1); ++i_) {
                        if ((long long)i_ * SAMP_THREADS >= V) break;            // block-uniform
                        const long long v = tid + (long long)i_ * SAMP_THREADS;
                        WMAR_TOPBYTE(v, xr[i_])
                    }
                } else {
                    for (long long v0 = 0; v0 < V; v0 += SAMP_THREADS) {
                        const long long v = v0 + tid;
                        const float xv = v < V ? x[v] : 0.f;
                        WMAR_TOPBYTE(v, xv)
                    }
                }
#undef WMAR_TOPBYTE
            } else {
                WMAR_FOR_ROW({
                    const uint32_t k = wmar_f32_key(xv);
                    if ((k & mask) == prefix) atomicAdd(&hist[(k >> (8 * pass)) & 255u], 1ull);
                })
            }
            __syncthreads();
            unsigned long long above;
            const int bkt = scan_hist<false>(hist, need, 0.f, &above);
            prefix |= (uint32_t)bkt << (8 * pass);
            mask |= 0xffu << (8 * pass);
            need -= above;
        }
        thr_key = prefix;
        // `need` is now the rank of the k-th entry inside the last bucket: top_k - need keys lie strictly above thr_key and the
        // last histogram counts the keys equal to it
        n_alive = (unsigned long long)a.top_k - need + hist_all[3][prefix & 255u];
    }
    if (thr_key <= NEG_INF_KEY) thr_key = NEG_INF_KEY + 1;  // -inf entries are never alive

    float best = -INFINITY;
    int besti = 0;
    if (EPT > 0 && n_alive <= (unsigned long long)SAMP_THREADS) {
        // ------------------------------------------------------------------------------------------------ compacted tail
        // At most one survivor per thread (any order: every reduction below is an integer sum, a histogram or a max with an index
        // tie-break); the arithmetic per element is the one of the sweeps in the general tail below.
        WMAR_FOR_ROW({
            if (wmar_f32_key(xv) >= thr_key) { const int s_ = atomicAdd(&surv_n, 1); surv_x[s_] = xv; surv_v[s_] = (int)v; }
        })
        __syncthreads();
        const bool my = tid < surv_n;
        const float cx = my ? surv_x[tid] : 0.f;
        const long long cv = my ? (long long)surv_v[tid] : 0;
        const uint32_t ck = wmar_f32_key(cx);
        const float ce = wmar_expf(cx - m);
        uint32_t bkey = 0;
        long long bidx = 0;
        if (a.use_top_p) {
            const unsigned long long S = block_sum_u64(my ? wmar_fx(ce) : 0ull, red_all[1]);
            const float Sf = wmar_fx_to_f32(S);
            const unsigned long long mass = my ? wmar_fx(ce / Sf) : 0ull;
            const float thr = a.top_p_thr;
            uint32_t prefix = 0, mask = 0;
            unsigned long long base = 0;
            bool all_pass = false;
            for (int pass = 3; pass >= 0; --pass) {
                unsigned long long* hist = hist_all[4 + 3 - pass];
                if (my && (ck & mask) == prefix) atomicAdd(&hist[(ck >> (8 * pass)) & 255u], mass);
                __syncthreads();
                unsigned long long below;
                const int bkt = scan_hist<true>(hist, base, thr, &below);
                if (bkt < 0) { all_pass = true; break; }
                prefix |= (uint32_t)bkt << (8 * pass);
                mask |= 0xffu << (8 * pass);
                base += below;
            }
            if (!all_pass) {
                bkey = prefix;
                const unsigned long long cnt = block_sum_u64((my && ck == bkey) ? 1ull : 0ull, red_all[2]);
                if (cnt > 1) {
                    uint32_t ipre = 0, imask = 0;
                    for (int pass = 3; pass >= 0; --pass) {
                        unsigned long long* hist = hist_all[8 + 3 - pass];
                        if (my && ck == bkey && (((uint32_t)cv) & imask) == ipre)
                            atomicAdd(&hist[(((uint32_t)cv) >> (8 * pass)) & 255u], mass);
                        __syncthreads();
                        unsigned long long below;
                        const int bkt = scan_hist<true>(hist, base, thr, &below);
                        ipre |= (uint32_t)(bkt < 0 ? 255 : bkt) << (8 * pass);
                        imask |= 0xffu << (8 * pass);
                        base += below;
                    }
                    bidx = (long long)ipre;
                }
            } else {
                bkey = kmax;
                bidx = (long long)block_max_u32((my && ck == kmax) ? (uint32_t)cv : 0u, red_all[3]);
            }
        }
        const bool kept = my && (ck > bkey || (ck == bkey && cv >= bidx));
        const unsigned long long S2 = block_sum_u64(kept ? wmar_fx(ce) : 0ull, red_all[4]);
        const float Sf2 = wmar_fx_to_f32(S2);
        if (kept) {
            // (entries outside the kept set race with p = 0 in the general tail and never win: the row maximum is kept and its
            // ratio is positive)
            const float qv = (EPT <= 16) ? q_lds[cv] : q[WMAR_SRC(cv)];
            best = (ce / Sf2) / qv;
            besti = (int)cv;
        }
    } else {
    // ---------------------------------------------------------------------------------------------------- general tail

    // top-p: boundary (K*, i*) of the ascending (value, index) order by mass-radix descent
    uint32_t bkey = 0;       // kept  <=>  key >= thr_key && (key > bkey || (key == bkey && idx >= bidx))
    long long bidx = 0;
    if (a.use_top_p) {
        unsigned long long S = 0;
        WMAR_FOR_ROW({
            if (wmar_f32_key(xv) >= thr_key) S += wmar_fx(wmar_expf(xv - m));
        })
        S = block_sum_u64(S, red_all[1]);
        const float Sf = wmar_fx_to_f32(S);
        const float thr = a.top_p_thr;
        uint32_t prefix = 0, mask = 0;
        unsigned long long base = 0;
        bool all_pass = false;
        for (int pass = 3; pass >= 0 && !all_pass; --pass) {
            unsigned long long* hist = hist_all[4 + 3 - pass];
            WMAR_FOR_ROW({
                const uint32_t k = wmar_f32_key(xv);
                if (k >= thr_key && (k & mask) == prefix)
                    atomicAdd(&hist[(k >> (8 * pass)) & 255u], wmar_fx(wmar_expf(xv - m) / Sf));
            })
            __syncthreads();
            unsigned long long below;
            const int bkt = scan_hist<true>(hist, base, thr, &below);
            if (bkt < 0) { all_pass = true; break; }
            prefix |= (uint32_t)bkt << (8 * pass);
            mask |= 0xffu << (8 * pass);
            base += below;
        }
        if (!all_pass) {
            // ties at the boundary value: refine on the index (ascending), 8 bits at a time
            bkey = prefix;
            unsigned long long cnt = 0;
            WMAR_FOR_ROW({ cnt += (wmar_f32_key(xv) == bkey); })
            cnt = block_sum_u64(cnt, red_all[2]);
            if (cnt > 1) {
                uint32_t ipre = 0, imask = 0;
                for (int pass = 3; pass >= 0; --pass) {
                    unsigned long long* hist = hist_all[8 + 3 - pass];
                    WMAR_FOR_ROW({
                        if (wmar_f32_key(xv) == bkey && (((uint32_t)v) & imask) == ipre)
                            atomicAdd(&hist[(((uint32_t)v) >> (8 * pass)) & 255u], wmar_fx(wmar_expf(xv - m) / Sf));
                    })
                    __syncthreads();
                    unsigned long long below;
                    const int bkt = scan_hist<true>(hist, base, thr, &below);
                    // the boundary value's bucket failed as a whole, so some index bucket fails too
                    ipre |= (uint32_t)(bkt < 0 ? 255 : bkt) << (8 * pass);
                    imask |= 0xffu << (8 * pass);
                    base += below;
                }
                bidx = (long long)ipre;
            }
        } else {
            // everything satisfies cum <= thr: only the last element of the order survives
            bkey = kmax;
            uint32_t imax = 0;
            WMAR_FOR_ROW({ if (wmar_f32_key(xv) == kmax) imax = max(imax, (uint32_t)v); })
            bidx = (long long)block_max_u32(imax, red_all[3]);
        }
    }

    // final softmax over the kept set + exponential race argmax(p / q), first index on ties
    unsigned long long S2 = 0;
    WMAR_FOR_ROW({
        const uint32_t k = wmar_f32_key(xv);
        const bool kept = k >= thr_key && (k > bkey || (k == bkey && v >= bidx));
        if (kept) S2 += wmar_fx(wmar_expf(xv - m));
    })
    S2 = block_sum_u64(S2, red_all[4]);
    const float Sf2 = wmar_fx_to_f32(S2);
    if (EPT > 0) {
        constexpr int CH = EPT > 16 ? 16 : (EPT > 0 ? EPT : 1);
#pragma unroll
        for (int c0 = 0; c0 < (EPT > 0 ? EPT : 1); c0 += CH) {
            float qv[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const long long v = tid + (long long)(c0 + j) * SAMP_THREADS;
                if (EPT <= 16) qv[j] = q_lds[tid + ((c0 + j) % QN) * SAMP_THREADS];
                else qv[j] = v < V ? q[WMAR_SRC(v)] : 1.f;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = c0 + j;
                const long long v = tid + (long long)i * SAMP_THREADS;
                if (v < V) {
                    const float xv = xr[i];
                    const uint32_t k = wmar_f32_key(xv);
                    const bool kept = k >= thr_key && (k > bkey || (k == bkey && v >= bidx));
                    const float e = kept ? wmar_expf(xv - m) : 0.0f;
                    const float r = (e / Sf2) / qv[j];
                    if (r > best) { best = r; besti = (int)v; }
                }
            }
        }
    } else {
        for (long long v = tid; v < V; v += SAMP_THREADS) {
            const float xv = x[v];
            const uint32_t k = wmar_f32_key(xv);
            const bool kept = k >= thr_key && (k > bkey || (k == bkey && v >= bidx));
            const float e = kept ? wmar_expf(xv - m) : 0.0f;
            const float r = (e / Sf2) / q[WMAR_SRC(v)];
            if (r > best) { best = r; besti = (int)v; }
        }
    }
    }       // general tail
#undef WMAR_FOR_ROW
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o);
        int oi = __shfl_xor(besti, o);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if ((tid & 63) == 0) { red_f[tid >> 6] = best; red_i[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < SAMP_WAVES; ++i)
            if (red_f[i] > best || (red_f[i] == best && red_i[i] < besti)) { best = red_f[i]; besti = red_i[i]; }
        const long long tokv = WMAR_SRC(besti);
        a.tok_out[b * a.tok_out_stride + step] = tokv;
        if (a.past_append) a.past_append[b * a.past_stride + t] = tokv;
    }
#undef WMAR_SRC
}

// ---------------------------------------------------------------------------- detector
struct DetArgs {
    WmDev wm;
    double gamma;
    const long long* codes;
    long long L;
    int* n_scored;
    int* n_green;
    double* pval;
    signed char* mask;
    long long mask_stride;
};

__device__ __forceinline__ long long det_num_ngrams(int seed_mode, int h, long long L, long long S) {
    if (seed_mode != WMAR_SEED_SPATIAL) return L - h;
    if (h == 1) return L - 1;            // every cell except (0,0)
    return (S - 1) * (S - 1);            // 2x2 blocks
}

// i-th n-gram of the enumeration in gentime_watermark.py:33-88, as positions into codes
__device__ __forceinline__ void det_ngram_pos(int seed_mode, int h, long long S, long long i, long long* pos) {
    if (seed_mode != WMAR_SEED_SPATIAL) {
        for (int j = 0; j <= h; ++j) pos[j] = i + j;
    } else if (h == 1) {
        long long cell = i + 1;          // row-major cell index, (0,0) skipped
        long long r = cell / S, c = cell % S;
        pos[0] = (c == 0) ? (r - 1) * S : cell - 1;
        pos[1] = cell;
    } else {
        long long r = i / (S - 1), c = i % (S - 1);
        pos[0] = r * S + c; pos[1] = r * S + c + 1; pos[2] = (r + 1) * S + c; pos[3] = (r + 1) * S + c + 1;
    }
}

__device__ double betainc_int(long long a, long long bb, double x) {
    if (a <= 0) return __longlong_as_double(0x7ff8000000000000ll);
    long long n = a + bb - 1;
    double lt = lgamma((double)n + 1.0) - lgamma((double)a + 1.0) - lgamma((double)(n - a) + 1.0) +
                (double)a * log(x) + (double)(n - a) * log1p(-x);
    double term = exp(lt), sum = term;
    double odds = x / (1.0 - x);
    for (long long k = a; k < n; ++k) {
        term *= ((double)(n - k) / (double)(k + 1)) * odds;
        sum += term;
    }
    return sum > 1.0 ? 1.0 : sum;
}

__global__ __launch_bounds__(256) void k_detect(DetArgs a) {
    __shared__ int red_s[4], red_g[4];
    const long long b = blockIdx.x;
    const long long* codes = a.codes + b * a.L;
    const int h = a.wm.h, n = h + 1;
    long long S = 0;
    if (a.wm.seed_mode == WMAR_SEED_SPATIAL) {
        S = (long long)(sqrt((double)a.L) + 0.5);
    }
    const long long cnt = det_num_ngrams(a.wm.seed_mode, h, a.L, S);
    signed char* mask = a.mask ? a.mask + b * a.mask_stride : nullptr;
    if (mask)
        for (int i = threadIdx.x; i < h; i += blockDim.x) mask[i] = -1;
    int ns = 0, ng = 0;
    const bool spatial = a.wm.seed_mode == WMAR_SEED_SPATIAL;       // (h + 1)-grams of at most 4 cells; LINEAR / FIXED: any h, consecutive
    for (long long i = threadIdx.x; i < cnt; i += blockDim.x) {
        long long pi[4], pj[4], ti[4];
        bool dup = false;
        long long sum = 0, tgt;
        if (spatial) {
            det_ngram_pos(a.wm.seed_mode, h, S, i, pi);
            for (int k = 0; k < n; ++k) ti[k] = codes[pi[k]];
            for (long long j = 0; j < i && !dup; ++j) {
                det_ngram_pos(a.wm.seed_mode, h, S, j, pj);
                bool same = true;
                for (int k = 0; k < n; ++k) same = same && (codes[pj[k]] == ti[k]);
                dup = same;
            }
            for (int k = 0; k < h; ++k) sum += ti[k];
            tgt = ti[h];
        } else {
            // the i-th n-gram is codes[i .. i + h] (gentime_watermark.py:33-44): no position arrays, any context size
            for (long long j = 0; j < i && !dup; ++j) {
                bool same = true;
                for (int k = 0; k < n && same; ++k) same = codes[j + k] == codes[i + k];
                dup = same;
            }
            for (int k = 0; k < h; ++k) sum += codes[i + k];
            tgt = codes[i + h];
        }
        int g = 0;
        if (!dup) {
            long long row = a.wm.seed_mode == WMAR_SEED_FIXED ? 0 : sum;
            if (row >= 0 && row < a.wm.n_rows && tgt >= 0 && tgt < a.wm.row_words * 32)
                g = (a.wm.table[row * a.wm.row_words + (tgt >> 5)] >> (tgt & 31)) & 1u;
            ns += 1;
            ng += g;
        }
        if (mask) mask[h + i] = dup ? -1 : (signed char)g;
    }
    for (int o = 32; o > 0; o >>= 1) { ns += __shfl_xor(ns, o); ng += __shfl_xor(ng, o); }
    if ((threadIdx.x & 63) == 0) { red_s[threadIdx.x >> 6] = ns; red_g[threadIdx.x >> 6] = ng; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ns = red_s[0] + red_s[1] + red_s[2] + red_s[3];
        ng = red_g[0] + red_g[1] + red_g[2] + red_g[3];
        a.n_scored[b] = ns;
        a.n_green[b] = ng;
        a.pval[b] = betainc_int(ng, 1 + (long long)ns - ng, a.gamma);
    }
}

// Launch helper shared with the generation graph (gpt.hip).
int launch_sample_fused(const SampArgs& a, hipStream_t st) {
    const bool plain = !a.gather && !a.logits_uncond && !a.logits_img && !a.allow && !a.trace;
    if (a.V <= 16ll * SAMP_THREADS && plain) hipLaunchKernelGGL((k_sample_fused<16, true>), dim3((unsigned)a.B), dim3(SAMP_THREADS), 0, st, a);
    else if (a.V <= 16ll * SAMP_THREADS) hipLaunchKernelGGL(k_sample_fused<16>, dim3((unsigned)a.B), dim3(SAMP_THREADS), 0, st, a);
    else if (a.V <= 64ll * SAMP_THREADS) hipLaunchKernelGGL(k_sample_fused<64>, dim3((unsigned)a.B), dim3(SAMP_THREADS), 0, st, a);
    else hipLaunchKernelGGL(k_sample_fused<0>, dim3((unsigned)a.B), dim3(SAMP_THREADS), 0, st, a);
    return launch_status("k_sample_fused");
}

}  // namespace wmar

using namespace wmar;

static int check_wm(const wmar_wm_ctx* wm, int64_t V) {
    WMAR_REQUIRE(wm->table_dev && wm->n_rows > 0, "watermark table missing");
    WMAR_REQUIRE(wm->vocab_size == V, "watermark vocab %lld != logits vocab %lld", (long long)wm->vocab_size, (long long)V);
    WMAR_REQUIRE(wm->seed_strategy >= 0 && wm->seed_strategy <= 2, "bad seed strategy");
    if (wm->seed_strategy == WMAR_SEED_SPATIAL)
        WMAR_REQUIRE(wm->context_size == 1 || wm->context_size == 3,
                     "Spatial seeding only implemented for context size in [1,3]");
    // LINEAR / FIXED: any context size (gentime_watermark.py:236-241); the key table has context_size * (V - 1) + 1 rows
    WMAR_REQUIRE(wm->context_size >= 0 && wm->context_size <= WMAR_MAX_CONTEXT, "context size %d unsupported (0..%d)", wm->context_size,
                 WMAR_MAX_CONTEXT);
    return WMAR_OK;
}

extern "C" {

int wmar_wm_process_logits(const wmar_wm_ctx* wm, float* logits_dev, int64_t B, const int64_t* past_ids_dev,
                           int64_t t, int64_t past_stride, void* stream) {
    // an EMPTY past (t == 0, e.g. RAR's first step: ids of shape [B, 0], rar.py:420,451) has no context: the reference catches
    // the ValueError per row and leaves the logits alone (gentime_watermark.py:266-269), so a null pointer with t == 0 is legal
    WMAR_REQUIRE(wm && logits_dev && (past_ids_dev || t == 0 || wm->seed_strategy == WMAR_SEED_FIXED), "process_logits: null argument");
    WMAR_REQUIRE(t >= 0, "process_logits: negative context length");
    if (int rc = check_wm(wm, wm->vocab_size)) return rc;
    if (B == 0) return WMAR_OK;
    if (!past_ids_dev && wm->seed_strategy != WMAR_SEED_FIXED) return WMAR_OK;   // every row skipped
    WmDev d = make_wm(wm);
    long long words = d.row_words;
    dim3 grid((unsigned)((words + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(k_wm_bias, grid, dim3(256), 0, (hipStream_t)stream, d, logits_dev, (long long)wm->vocab_size,
                       (const long long*)past_ids_dev, (long long)t, (long long)past_stride);
    return launch_status("k_wm_bias");
}

int wmar_sample_fused(const wmar_wm_ctx* wm, const float* logits_dev, int64_t B, int64_t V,
                      const int64_t* past_ids_dev, int64_t t, int64_t past_stride, float temperature,
                      int32_t top_k, double top_p, const float* q_dev, float* scratch_dev,
                      int64_t* tok_out_dev, void* stream) {
    WMAR_REQUIRE(logits_dev && q_dev && scratch_dev && tok_out_dev, "sample_fused: null argument");
    WMAR_REQUIRE(V > 0 && V < (1ll << 31), "sample_fused: bad vocab");
    WMAR_REQUIRE(!(top_p >= 0) || top_p <= 1.0, "`top_p` has to be a float > 0 and < 1, but is %f", top_p);
    if (wm) {
        if (int rc = check_wm(wm, V)) return rc;
        WMAR_REQUIRE(past_ids_dev || t == 0 || wm->seed_strategy == WMAR_SEED_FIXED, "sample_fused: past_ids required");
        if (!past_ids_dev && wm->seed_strategy != WMAR_SEED_FIXED) wm = nullptr;   // empty past: no row has a context (rows skipped)
    }
    if (B == 0) return WMAR_OK;
    SampArgs a{};
    a.wm = make_wm(wm);
    a.logits = logits_dev;
    a.V = V;
    a.past = (const long long*)past_ids_dev;
    a.past_stride = past_stride;
    a.t_host = t;
    a.temperature = temperature;
    a.top_k = top_k;
    a.use_top_p = top_p >= 0;
    a.top_p_thr = (float)(1.0 - top_p);
    a.q = q_dev;
    a.scratch = scratch_dev;
    a.tok_out = (long long*)tok_out_dev;
    a.tok_out_stride = 1;
    a.B = B;
    return launch_sample_fused(a, (hipStream_t)stream);
}

int wmar_cham_sample(const wmar_wm_ctx* wm, const float* logits3_dev, int64_t B, int64_t V, const int64_t* past_ids_dev,
                     int64_t t, int64_t past_stride, float temperature, double top_p, float guidance_scale_text,
                     float guidance_scale_image, const uint32_t* allow_dev, const int32_t* allow_ids_dev, int32_t n_allow,
                     const float* q_dev, float* scratch_dev, int64_t* tok_out_dev, void* stream) {
    WMAR_REQUIRE(logits3_dev && q_dev && scratch_dev && tok_out_dev, "cham_sample: null argument");
    WMAR_REQUIRE(V > 0 && V < (1ll << 31), "cham_sample: bad vocab");
    WMAR_REQUIRE(!(top_p >= 0) || top_p <= 1.0, "`top_p` has to be a float > 0 and < 1, but is %f", top_p);
    if (wm) {
        if (int rc = check_wm(wm, V)) return rc;
        WMAR_REQUIRE(past_ids_dev || wm->seed_strategy == WMAR_SEED_FIXED, "cham_sample: past_ids required");
    }
    if (B == 0) return WMAR_OK;
    SampArgs a{};
    a.wm = make_wm(wm);
    a.logits = logits3_dev;
    a.logits_img = logits3_dev + B * V;
    a.logits_uncond = logits3_dev + 2 * B * V;
    a.g_text = guidance_scale_text;
    a.g_image = guidance_scale_image;
    a.allow = allow_dev;
    a.V = V;
    if (allow_ids_dev && n_allow > 0) { a.gather = allow_ids_dev; a.Vsrc = V; a.V = n_allow; }
    a.past = (const long long*)past_ids_dev;
    a.past_stride = past_stride;
    a.t_host = t;
    a.temperature = temperature;
    a.top_k = 0;
    a.use_top_p = top_p >= 0;
    a.top_p_thr = (float)(1.0 - top_p);
    a.q = q_dev;
    a.scratch = scratch_dev;
    a.tok_out = (long long*)tok_out_dev;
    a.tok_out_stride = 1;
    a.B = B;
    return launch_sample_fused(a, (hipStream_t)stream);
}

int64_t wmar_detect_num_ngrams(int32_t seed_strategy, int32_t context_size, int64_t L) {
    if (seed_strategy != WMAR_SEED_SPATIAL) return L - context_size;
    long long S = (long long)(sqrt((double)L) + 0.5);
    if (S * S != L) return WMAR_EINVAL;
    return context_size == 1 ? L - 1 : (S - 1) * (S - 1);
}

int wmar_detect(const wmar_wm_ctx* wm, double gamma, const int64_t* codes_dev, int64_t B, int64_t L,
                int32_t* n_scored_dev, int32_t* n_green_dev, double* pval_dev, int8_t* mask_dev,
                int64_t mask_stride, void* stream) {
    WMAR_REQUIRE(wm && codes_dev && n_scored_dev && n_green_dev && pval_dev, "detect: null argument");
    if (int rc = check_wm(wm, wm->vocab_size)) return rc;
    if (L - wm->context_size < 1) {
        set_error("Must have at least 1 token to score after the first min_context_len=%d tokens required by the seeding scheme.",
                  wm->context_size);
        return WMAR_ESHORT;
    }
    if (wm->seed_strategy == WMAR_SEED_SPATIAL) {
        long long S = (long long)(sqrt((double)L) + 0.5);
        WMAR_REQUIRE(S * S == L, "Sequence must be a square");
    }
    if (mask_dev) WMAR_REQUIRE(mask_stride >= wm->context_size + wmar_detect_num_ngrams(wm->seed_strategy, wm->context_size, L),
                               "detect: mask stride too small");
    if (B == 0) return WMAR_OK;
    DetArgs a{};
    a.wm = make_wm(wm);
    a.gamma = gamma;
    a.codes = (const long long*)codes_dev;
    a.L = L;
    a.n_scored = n_scored_dev;
    a.n_green = n_green_dev;
    a.pval = pval_dev;
    a.mask = (signed char*)mask_dev;
    a.mask_stride = mask_stride;
    hipLaunchKernelGGL(k_detect, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    return launch_status("k_detect");
}

}  // extern "C"
