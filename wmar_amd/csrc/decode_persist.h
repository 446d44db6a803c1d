// ONE launch per decode step for 1..5 sequences (the reference's own batch sizes: 1 and 5): the phases of decode_small.h -- per layer
// QKV, attention, projection, FC1, FC2; then the vocabulary head -- inside a persistent kernel, 256 workgroups (one per CU), with a
// device-wide barrier between phases and the WEIGHT STREAM RUNNING ACROSS THE BARRIERS.
//
// Reference: deps/taming/modules/transformer/mingpt.py:69-95, :112-122, :183-214 (the same arithmetic as decode_small.h).
//
// Why.  As five launches per layer the small-batch step sits at its design floor: every launch pays ~4 us during which the HBM idles
// (boundary, first-load latency, reduction + store tail, drain) against 1.6-6 us of streaming -- 33 us per layer at batch 1 for a
// 19 us stream (round 6: 1.63 ms per step = 0.43 of the byte floor).  A barrier costs what a boundary costs (2.8 us measured,
// profiles/r05_grid_barrier.log), so a persistent kernel gains nothing by itself; what it buys is that the NEXT phase's weights do not
// depend on the barrier: every wave keeps a ring of two weight groups (9 KiB each) and refills a slot the moment it has multiplied
// out of it -- with the following phase's (and, behind FC2, the following layer's) first groups.  19 MB are in flight chip-wide while
// the workgroups reduce, publish, wait at the barrier and stage their input rows.
//
// Structure of a workgroup: 4 COMPUTE waves + 1 SERVICE wave (320 threads).
//   * compute wave w owns (K segment, column range) of the phase as in k_sgemv; it only ever waits for its own weight loads.  The
//     input rows live in LDS (staged once per phase by all five waves, LayerNorm applied on the way in, row PAIRS interleaved so that a
//     16-byte LDS read is two v_pk_fma_f32 operands);
//   * the service wave does everything that needs `s_waitcnt vmcnt(0)`: it sums the segments' partials, applies the epilogue,
//     publishes the results (agent-scope 4-byte atomic stores: write-through), waits for their acknowledgement, arrives at the barrier
//     and polls it.  On gfx9 a store is only known to be complete at vmcnt(0), which would also wait for every prefetched weight of
//     the same wave -- so the waves that stream never publish, and the wave that publishes never streams.
//   * workgroup-level synchronisation is `s_waitcnt lgkmcnt(0); s_barrier` by hand: hipcc's __syncthreads() drains vmcnt as well.
// Exchange between workgroups (residual rows, q, attention output, hidden rows; the new K / V rows): agent-scope relaxed atomic stores
// and 8-byte agent-scope atomic loads on both sides (MI355X_MICROARCH.md, "valid forms"); K / V rows of EARLIER steps and all weights
// are read with plain non-temporal loads (a kernel boundary lies in between).
// Barrier: XCD-hierarchical (scripts/grid_barrier2_bench.hip): arrival and release inside an XCD through its own L2, one device-scope
// atomic per XCD between them; every wait is BOUNDED -- a workgroup that gives up raises a flag and leaves, the others follow at their
// next wait, later launches of the captured loop return at once, and the host re-runs the call on the five-launch plan
// (gpt_sync_failed).  Needs: all 256 workgroups resident (checked at engine creation) and the blockIdx % 8 -> XCD grouping the
// engine already probes for k_bx_xr; if either fails the barrier times out and the engine falls back -- slower, never wrong.
#pragma once
#include "decode_small.h"

namespace wmar {

#pragma clang fp contract(off)      // pinned arithmetic, as decode_small.h

// The small-batch weights of all layers live in ONE allocation with a fixed per-layer stride, so that the kernel forms every address
// from a kernel-argument base (global address space: `global_load`, counted by vmcnt only -- a pointer fetched from a table in memory is
// a flat pointer, and flat loads also count on lgkmcnt, which the workgroup barrier below waits for).  Offsets in floats, n_embd 1536.
constexpr long long SSD = 2 * SG_SEG, SSDD = SSD * SSD;
constexpr long long SS_OFF_WQKV = 0, SS_OFF_WPROJ = 3 * SSDD, SS_OFF_WFC1 = 4 * SSDD, SS_OFF_WFC2 = 8 * SSDD, SS_OFF_BQKV = 12 * SSDD,
                    SS_OFF_BPROJ = SS_OFF_BQKV + 3 * SSD, SS_OFF_BFC1 = SS_OFF_BPROJ + SSD, SS_OFF_BFC2 = SS_OFF_BFC1 + 4 * SSD,
                    SS_OFF_LN1W = SS_OFF_BFC2 + SSD, SS_OFF_LN1B = SS_OFF_LN1W + SSD, SS_OFF_LN2W = SS_OFF_LN1B + SSD,
                    SS_OFF_LN2B = SS_OFF_LN2W + SSD, SS_LAYER_FLOATS = SS_OFF_LN2B + SSD;

struct SsArgs {
    const float* arena;        // [L][SS_LAYER_FLOATS]
    int L;
    const float *tok_emb, *pos_emb, *whead, *lnfw, *lnfb;
    const long long* tok; long long tok_stride; int tok_use_pos;
    float *xs, *ys, *hs, *qs;  // [rows][D], [rows][D], [rows][4D], [rows][D]: exchanged between workgroups inside the launch
    float *kcache, *vcache; long long lstride;      // layer stride in floats
    float* logits; int V;
    const int* pos_dev;
    int D, H, Tmax;
    unsigned* bar;             // [8][64] per-XCD words (0: arrivals, 32: generation), [64] device words (0: arrivals, 32: generation), then the fail flag
};

constexpr int SS_THREADS = 320;            // 4 compute waves + the service wave
constexpr int SS_WGS = 256;
constexpr int SS_BAR_WORDS = 8 * 64 + 64 + 64;
constexpr int SS_MAX_POLLS = 1 << 16;      // ~20-30 ms against a 3 us barrier
constexpr int SS_MAX_ROWS = 5;
// LDS carve-up (bytes)
constexpr int SS_LDS_LNRED = 0;            // double [5][16]
constexpr int SS_LDS_PART = 1024;          // float [<= 1024]
constexpr int SS_LDS_FLAG = 5120;          // unsigned [4]
constexpr int SS_LDS_X = 8192;             // input rows (or the attention's merge buffers)
__host__ __device__ constexpr int ss_nbp(int nb) { return (nb + 1) / 2; }
__host__ __device__ constexpr int ss_xstride(int nb, bool pad) { return ss_nbp(nb) * 8 + (pad ? 4 : 0); }     // floats per 4 k
__host__ __device__ constexpr int ss_lds_bytes(int nb) { return SS_LDS_X + (6144 / 4) * ss_xstride(nb, false) * 4; }

// ---- exchange: agent-scope atomics, tracked by the compiler's wait counts
__device__ __forceinline__ float2 ld_x2(const float* p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ float ld_x1(const float* p) {
    return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_x1(float* p, float v) {
    __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// workgroup barrier that does NOT drain the vector-memory counter (the prefetched weights stay in flight)
__device__ __forceinline__ void wg_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// device-scope words of the barrier (sc1: performed at the memory side); the XCD-local ones are l2_* of decoder_kernels.h
__device__ __forceinline__ unsigned dev_add_u32(unsigned* p, unsigned v) {
    unsigned o;
    asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(o) : "v"(p), "v"(v) : "memory");
    return o;
}
__device__ __forceinline__ void dev_add_u32_noret(unsigned* p, unsigned v) { asm volatile("global_atomic_add %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned dev_read_u32(unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// One lane of the service wave.  Returns false when a wait gave up (the fail flag is then up).
__device__ __forceinline__ bool ss_grid_barrier(unsigned* bar, unsigned& phase) {
    unsigned* cnt = bar + (blockIdx.x & 7) * 64;
    unsigned* gen = cnt + 32;
    unsigned* gs = bar + 8 * 64;
    unsigned* fail = bar + 8 * 64 + 64;
    bool ok = true;
    const unsigned old = l2_add_u32(cnt, 1u);
    if (old == SS_WGS / 8 - 1u) {
        l2_swap_u32_noret(cnt, 0u);
        const unsigned g = dev_add_u32(gs, 1u);
        if (g == 8u * phase + 7u) dev_add_u32_noret(gs + 32, 1u);
        else {
            int n = 0;
            while (dev_read_u32(gs + 32) == phase && ++n < SS_MAX_POLLS) __builtin_amdgcn_s_sleep(1);
            ok = n < SS_MAX_POLLS;
        }
        l2_add_u32_noret(gen, 1u);          // released even after a timeout: the group's members must not wait their full bound as well
    } else {
        int n = 0;
        while (l2_read_u32(gen) == phase && ++n < SS_MAX_POLLS) __builtin_amdgcn_s_sleep(1);
        ok = n < SS_MAX_POLLS;
    }
    if (!ok) dev_add_u32_noret(fail, 1u);
    phase += 1u;
    return ok;
}

enum { SS_QKV = 0, SS_PROJ = 1, SS_FC1 = 2, SS_FC2 = 3 };
template <int ROLE> __host__ __device__ constexpr int ss_groups() { return ROLE == SS_QKV ? 3 : ROLE == SS_PROJ ? 1 : 4; }        // per wave
template <int ROLE> __host__ __device__ constexpr int ss_cols_wg() { return ROLE == SS_QKV ? 18 : ROLE == SS_FC1 ? 24 : 6; }
template <int ROLE> __host__ __device__ constexpr int ss_K() { return ROLE == SS_FC2 ? 6144 : 1536; }

// weight group G of ROLE for compute wave w: 3 columns x 3 loads of 1 KiB.  K = D roles: segment w >> 1, column half w & 1, three
// columns per group; FC2 (eight segments over four waves): unit u = (segment 2w + (u >> 1), column triple u & 1).
// (Macros on a kernel-local array, as in k_sgemv: passed by reference through inlined functions the ring ended up in scratch memory.)
#define WMAR_SS_LOADW(ROLE, BUF, WPTR, G)                                                                          \
    {                                                                                                              \
        constexpr int K_ = ss_K<ROLE>();                                                                           \
        const int seg_ = ROLE == SS_FC2 ? 2 * w + ((G) >> 1) : (w >> 1);                                           \
        const int col0_ = blockIdx.x * ss_cols_wg<ROLE>() + (ROLE == SS_FC2 ? ((G) & 1) * 3 : (w & 1) * (3 * ss_groups<ROLE>()) + (G) * 3); \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                            \
            const float4* p_ = (const float4*)((WPTR) + (long long)(col0_ + c) * K_ + seg_ * SG_SEG) + lane;       \
            _Pragma("unroll") for (int ch = 0; ch < 3; ++ch) BUF[c * 3 + ch] = ld_nt(p_ + ch * 64);                \
        }                                                                                                          \
    }
// multiply group G out of BUF against the staged rows, reduce over the wave, leave the partial sums in LDS
#define WMAR_SS_COMPUTE(ROLE, PAD, BUF, G)                                                                         \
    {                                                                                                              \
        constexpr int S_ = ss_xstride(NB, PAD), R_ = 3 * NB;                                                       \
        const int seg_ = ROLE == SS_FC2 ? 2 * w + ((G) >> 1) : (w >> 1);                                           \
        f32x2 acc2[3][NBP];                                                                                        \
        _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                              \
            _Pragma("unroll") for (int bp = 0; bp < NBP; ++bp) acc2[c][bp] = f32x2{0.f, 0.f};                      \
        _Pragma("unroll") for (int ch = 0; ch < 3; ++ch) {                                                         \
            const float4* xq = (const float4*)(xl + (long long)(seg_ * 192 + ch * 64 + lane) * S_);                \
            _Pragma("unroll") for (int bp = 0; bp < NBP; ++bp) {                                                   \
                const float4 q0 = xq[2 * bp], q1 = xq[2 * bp + 1];   /* (x_b.x, x_b'.x, x_b.y, x_b'.y), (.z .z .w .w) */ \
                const f32x2 xx = {q0.x, q0.y}, xy = {q0.z, q0.w}, xz = {q1.x, q1.y}, xw = {q1.z, q1.w};           \
                _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                    \
                    const float4 wv = BUF[c * 3 + ch];                                                             \
                    f32x2 s_ = acc2[c][bp];                                                                        \
                    s_ = __builtin_elementwise_fma(f32x2{wv.x, wv.x}, xx, s_); s_ = __builtin_elementwise_fma(f32x2{wv.y, wv.y}, xy, s_); \
                    s_ = __builtin_elementwise_fma(f32x2{wv.z, wv.z}, xz, s_); s_ = __builtin_elementwise_fma(f32x2{wv.w, wv.w}, xw, s_); \
                    acc2[c][bp] = s_;                                                                              \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
        float acc[R_];                                                                                             \
        _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                              \
            _Pragma("unroll") for (int b = 0; b < NB; ++b) acc[c * NB + b] = (b & 1) ? acc2[c][b >> 1].y : acc2[c][b >> 1].x; \
        int ridx;                                                                                                  \
        const float tot_ = wave_reduce_many<R_, float>(acc, lane, &ridx);                                          \
        constexpr int LBR_ = sg_log2p<R_>();                                                                       \
        if ((lane & ((64 >> LBR_) - 1)) == 0 && ridx < R_) {                                                       \
            const int c_ = ridx / NB, b_ = ridx - c_ * NB;                                                         \
            const int slot_ = ROLE == SS_FC2 ? ((G) & 1) * 3 + c_ : (w & 1) * (3 * ss_groups<ROLE>()) + (G) * 3 + c_; \
            part[(seg_ * ss_cols_wg<ROLE>() + slot_) * NB + b_] = tot_;                                            \
        }                                                                                                          \
    }

// Stage the input rows of a phase into LDS, all 320 threads: rows [NB][K] from `src` (agent-scope loads), optional LayerNorm
// (statistics: fp32 per thread, fp64 across the workgroup, fixed order), row pairs interleaved.  Ends with a workgroup barrier.
template <int NB, int K, bool LN, bool PAD>
__device__ __forceinline__ void ss_stage(const float* src, const float* gamma, const float* beta, float* xl, double* lnred) {
    constexpr int NBP = ss_nbp(NB), S = ss_xstride(NB, PAD), NIT = (K / 4 + SS_THREADS - 1) / SS_THREADS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (!LN) {
        // no statistics: stream through (few registers: the compute waves hold their weight ring)
#pragma unroll 2
        for (int it = 0; it < NIT; ++it) {
            const int k4 = tid + it * SS_THREADS;
            if (k4 < K / 4) {
                float4 v[2 * NBP];
#pragma unroll
                for (int b = 0; b < 2 * NBP; ++b) {
                    if (b < NB) {
                        const float2 lo = ld_x2(src + (long long)b * K + k4 * 4), hi = ld_x2(src + (long long)b * K + k4 * 4 + 2);
                        v[b] = make_float4(lo.x, lo.y, hi.x, hi.y);
                    } else v[b] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float4* o = (float4*)(xl + (long long)k4 * S);
#pragma unroll
                for (int bp = 0; bp < NBP; ++bp) {
                    o[2 * bp] = make_float4(v[2 * bp].x, v[2 * bp + 1].x, v[2 * bp].y, v[2 * bp + 1].y);
                    o[2 * bp + 1] = make_float4(v[2 * bp].z, v[2 * bp + 1].z, v[2 * bp].w, v[2 * bp + 1].w);
                }
            }
        }
        wg_sync();
        return;
    }
    // LayerNorm: pass A leaves the RAW rows in LDS and keeps only the per-thread sums (the compute waves hold their weight ring: no
    // room for the rows in registers); pass B normalises the thread's own elements in place.
    double st[2 * NB];
    {
        float s[NB], ss[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { s[b] = 0.f; ss[b] = 0.f; }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k4 = tid + it * SS_THREADS;
            if (k4 < K / 4) {
                float4 v[2 * NBP];
#pragma unroll
                for (int b = 0; b < 2 * NBP; ++b) {
                    if (b < NB) {
                        const float2 lo = ld_x2(src + (long long)b * K + k4 * 4), hi = ld_x2(src + (long long)b * K + k4 * 4 + 2);
                        v[b] = make_float4(lo.x, lo.y, hi.x, hi.y);
                        s[b] += (v[b].x + v[b].y) + (v[b].z + v[b].w);
                        ss[b] += (v[b].x * v[b].x + v[b].y * v[b].y) + (v[b].z * v[b].z + v[b].w * v[b].w);
                    } else v[b] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float4* o = (float4*)(xl + (long long)k4 * S);
#pragma unroll
                for (int bp = 0; bp < NBP; ++bp) {
                    o[2 * bp] = make_float4(v[2 * bp].x, v[2 * bp + 1].x, v[2 * bp].y, v[2 * bp + 1].y);
                    o[2 * bp + 1] = make_float4(v[2 * bp].z, v[2 * bp + 1].z, v[2 * bp].w, v[2 * bp + 1].w);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) { st[2 * b] = (double)s[b]; st[2 * b + 1] = (double)ss[b]; }
    }
    int sidx;
    const double tot = wave_reduce_many<2 * NB, double>(st, lane, &sidx);
    constexpr int LB = sg_log2p<2 * NB>();
    if ((lane & ((64 >> LB) - 1)) == 0 && sidx < 2 * NB) lnred[w * 16 + sidx] = tot;
    wg_sync();
    const double invK = inv_count_f64((double)K);
    float mu[2 * NBP], rstd[2 * NBP];
#pragma unroll
    for (int b = 0; b < 2 * NBP; ++b) {
        if (b < NB) {
            double sm = 0.0, sq = 0.0;
#pragma unroll
            for (int ww = 0; ww < SS_THREADS / 64; ++ww) { sm += lnred[ww * 16 + 2 * b]; sq += lnred[ww * 16 + 2 * b + 1]; }
            const double mean = sm * invK;
            mu[b] = (float)mean;
            rstd[b] = rsqrtf((float)var_f64(sq * invK, mean) + 1e-5f);
        } else { mu[b] = 0.f; rstd[b] = 0.f; }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k4 = tid + it * SS_THREADS;
        if (k4 < K / 4) {
            const float4 gm = *((const float4*)gamma + k4), bt = *((const float4*)beta + k4);
            float4* o = (float4*)(xl + (long long)k4 * S);
#pragma unroll
            for (int bp = 0; bp < NBP; ++bp) {
                float4 q0 = o[2 * bp], q1 = o[2 * bp + 1];             // (b.x, b'.x, b.y, b'.y), (b.z, b'.z, b.w, b'.w)
                const float m0 = mu[2 * bp], r0 = rstd[2 * bp], m1 = mu[2 * bp + 1], r1 = rstd[2 * bp + 1];
                const bool two = 2 * bp + 1 < NB;
                q0.x = (q0.x - m0) * r0 * gm.x + bt.x; q0.z = (q0.z - m0) * r0 * gm.y + bt.y;
                q1.x = (q1.x - m0) * r0 * gm.z + bt.z; q1.z = (q1.z - m0) * r0 * gm.w + bt.w;
                q0.y = two ? (q0.y - m1) * r1 * gm.x + bt.x : 0.f; q0.w = two ? (q0.w - m1) * r1 * gm.y + bt.y : 0.f;
                q1.y = two ? (q1.y - m1) * r1 * gm.z + bt.z : 0.f; q1.w = two ? (q1.w - m1) * r1 * gm.w + bt.w : 0.f;
                o[2 * bp] = q0; o[2 * bp + 1] = q1;
            }
        }
    }
    wg_sync();
}

// the service wave's end of a phase: publish (the caller has stored), wait for the acknowledgement, device-wide barrier.  Every
// thread of the workgroup then passes the SAME workgroup barrier and learns whether to go on.
__device__ __forceinline__ bool ss_phase_end(const SsArgs& a, unsigned* flag_lds, unsigned& phase, bool service, int lane) {
    if (service) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) { if (!ss_grid_barrier(a.bar, phase)) flag_lds[0] = 1u; }
    }
    wg_sync();
    return flag_lds[0] == 0u;
}

template <int NB>
__global__ __launch_bounds__(SS_THREADS) void k_sstep(SsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];
    double* lnred = (double*)(ss_smem + SS_LDS_LNRED);
    float* part = (float*)(ss_smem + SS_LDS_PART);
    unsigned* flag = (unsigned*)(ss_smem + SS_LDS_FLAG);
    float* xl = (float*)(ss_smem + SS_LDS_X);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = w == 4;
    const int pos = *a.pos_dev;
    unsigned phase = 0;

    // a launch behind a failed one returns at once; the barrier generation continues where the last launch left it
    if (tid == 0) flag[0] = dev_read_u32(a.bar + 8 * 64 + 64) != 0u ? 1u : 0u;
    if (service && lane == 0) phase = l2_read_u32(a.bar + (blockIdx.x & 7) * 64 + 32);
    wg_sync();
    if (flag[0] != 0u) return;

    constexpr int NBP = ss_nbp(NB);
    // the weight ring: TWO groups of 9 KiB per compute wave (19 MB chip-wide: ~3 us of HBM time, a barrier + a staging); group i of the
    // layer's twelve (QKV 0-2, projection 3, FC1 4-7, FC2 8-11) lives in slot i & 1 and is refilled with group i + 2 as soon as it has
    // been multiplied out.  (Three slots spilled: a fifth wave per workgroup caps every wave at 256 registers.)
    float4 wb[2][9];
    if (!service) {
        const float* W0 = a.arena + SS_OFF_WQKV;
        WMAR_SS_LOADW(SS_QKV, wb[0], W0, 0) WMAR_SS_LOADW(SS_QKV, wb[1], W0, 1)
    }
    // ---- embedding (mingpt.py:186-200): the service waves of all workgroups share the rows
    if (service) {
        const int n4 = NB * (a.D / 4);
        for (int i = blockIdx.x * 64 + lane; i < n4; i += SS_WGS * 64) {
            const int b = i / (a.D / 4), k4 = i - b * (a.D / 4);
            const long long tk = a.tok[(long long)b * a.tok_stride + (a.tok_use_pos ? pos : 0)];
            const float4 e = *((const float4*)(a.tok_emb + tk * a.D) + k4), pe = *((const float4*)(a.pos_emb + (long long)pos * a.D) + k4);
            float* o = a.xs + (long long)b * a.D + k4 * 4;
            st_x1(o, e.x + pe.x); st_x1(o + 1, e.y + pe.y); st_x1(o + 2, e.z + pe.z); st_x1(o + 3, e.w + pe.w);
        }
    }
    if (!ss_phase_end(a, flag, phase, service, lane)) return;

    for (int l = 0; l < a.L; ++l) {
        const float* Lw = a.arena + (long long)l * SS_LAYER_FLOATS;
        const float* Ln = a.arena + (long long)(l + 1 < a.L ? l + 1 : l) * SS_LAYER_FLOATS;           // behind the last layer: a clamped (redundant) prefetch
        float* kc = a.kcache + (long long)l * a.lstride;
        float* vc = a.vcache + (long long)l * a.lstride;

        // ================= QKV: LN1 + projection + bias; k / v rows into the cache, q into its buffer
        float eb0 = 0.f, eb1 = 0.f;
        if (service) {        // the epilogue's operands, requested at the start of the phase
            if (lane < 18 * NB) eb0 = Lw[SS_OFF_BQKV + blockIdx.x * 18 + lane % 18];
            if (lane + 64 < 18 * NB) eb1 = Lw[SS_OFF_BQKV + blockIdx.x * 18 + (lane + 64) % 18];
        }
        ss_stage<NB, 1536, true, true>(a.xs, Lw + SS_OFF_LN1W, Lw + SS_OFF_LN1B, xl, lnred);
        if (!service) {
            WMAR_SS_COMPUTE(SS_QKV, true, wb[0], 0) WMAR_SS_LOADW(SS_QKV, wb[0], Lw + SS_OFF_WQKV, 2)
            WMAR_SS_COMPUTE(SS_QKV, true, wb[1], 1) WMAR_SS_LOADW(SS_PROJ, wb[1], Lw + SS_OFF_WPROJ, 0)
            WMAR_SS_COMPUTE(SS_QKV, true, wb[0], 2) WMAR_SS_LOADW(SS_FC1, wb[0], Lw + SS_OFF_WFC1, 0)
        }
        wg_sync();
        if (service) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int t = lane + 64 * r;
                if (t < 18 * NB) {
                    const int b = t / 18, col = t - b * 18, n = blockIdx.x * 18 + col;
                    const float s = part[col * NB + b] + part[(18 + col) * NB + b] + (r ? eb1 : eb0);
                    const int which = n / a.D, j = n - which * a.D;
                    if (which == 0) st_x1(a.qs + (long long)b * a.D + j, s);
                    else st_x1((which == 1 ? kc : vc) + (((long long)b * a.H + (j >> 6)) * a.Tmax + pos) * 64 + (j & 63), s);
                }
            }
        }
        if (!ss_phase_end(a, flag, phase, service, lane)) return;

        // ================= attention: workgroup (b, h) for blockIdx < NB * H; the others go straight to the barrier
        {
            float* sm = xl; float* sl = xl + 16; float* so = xl + 64;       // merge buffers alias the (free) row region
            const bool mine = (int)blockIdx.x < NB * a.H;
            const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
            if (mine && !service) {
                const int p = lane & 15, rr = lane >> 4, T = pos + 1;
                const float* qp = a.qs + (long long)b * a.D + h * 64 + p * 4;
                const float2 qlo = ld_x2(qp), qhi = ld_x2(qp + 2);
                const float4 q4 = make_float4(qlo.x, qlo.y, qhi.x, qhi.y);
                const long long base = ((long long)b * a.H + h) * a.Tmax * 64;
                const float4* Kp = (const float4*)(kc + base) + p;
                const float4* Vp = (const float4*)(vc + base) + p;
                float m = -INFINITY, lsum = 0.f;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                const int nu = (T + 3) >> 2;
                for (int u0 = w; u0 < nu; u0 += 32) sattn_rows<8>(Kp, Vp, T, u0, rr, q4, 0.125f, m, lsum, o);
                const int gi = w * 4 + rr;
                if (p == 0) { sm[gi] = m; sl[gi] = lsum; }
                *((float4*)&so[gi * 64 + p * 4]) = o;
            }
            wg_sync();
            if (mine && service) {
                float M = sm[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) M = fmaxf(M, sm[i]);
                float Ls = 0.f, O = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = expf(sm[i] - M);
                    Ls = fmaf(sl[i], e, Ls);
                    O = fmaf(so[i * 64 + lane], e, O);
                }
                st_x1(a.ys + (long long)b * a.D + h * 64 + lane, O / Ls);
            }
            if (!ss_phase_end(a, flag, phase, service, lane)) return;
        }

        // ================= output projection + residual
        if (service && lane < 6 * NB) { eb0 = Lw[SS_OFF_BPROJ + blockIdx.x * 6 + lane % 6]; eb1 = ld_x1(a.xs + (long long)(lane / 6) * a.D + blockIdx.x * 6 + lane % 6); }
        ss_stage<NB, 1536, false, true>(a.ys, nullptr, nullptr, xl, lnred);
        if (!service) { WMAR_SS_COMPUTE(SS_PROJ, true, wb[1], 0) WMAR_SS_LOADW(SS_FC1, wb[1], Lw + SS_OFF_WFC1, 1) }
        wg_sync();
        if (service && lane < 6 * NB) {
            const int b = lane / 6, col = lane - b * 6;
            st_x1(a.xs + (long long)b * a.D + blockIdx.x * 6 + col, eb1 + (part[col * NB + b] + part[(6 + col) * NB + b] + eb0));
        }
        if (!ss_phase_end(a, flag, phase, service, lane)) return;

        // ================= LN2 + FC1 + bias + GELU
        if (service) {
            if (lane < 24 * NB) eb0 = Lw[SS_OFF_BFC1 + blockIdx.x * 24 + lane % 24];
            if (lane + 64 < 24 * NB) eb1 = Lw[SS_OFF_BFC1 + blockIdx.x * 24 + (lane + 64) % 24];
        }
        ss_stage<NB, 1536, true, true>(a.xs, Lw + SS_OFF_LN2W, Lw + SS_OFF_LN2B, xl, lnred);
        if (!service) {
            WMAR_SS_COMPUTE(SS_FC1, true, wb[0], 0) WMAR_SS_LOADW(SS_FC1, wb[0], Lw + SS_OFF_WFC1, 2)
            WMAR_SS_COMPUTE(SS_FC1, true, wb[1], 1) WMAR_SS_LOADW(SS_FC1, wb[1], Lw + SS_OFF_WFC1, 3)
            WMAR_SS_COMPUTE(SS_FC1, true, wb[0], 2) WMAR_SS_LOADW(SS_FC2, wb[0], Lw + SS_OFF_WFC2, 0)
            WMAR_SS_COMPUTE(SS_FC1, true, wb[1], 3) WMAR_SS_LOADW(SS_FC2, wb[1], Lw + SS_OFF_WFC2, 1)
        }
        wg_sync();
        if (service) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int t = lane + 64 * r;
                if (t < 24 * NB) {
                    const int b = t / 24, col = t - b * 24;
                    st_x1(a.hs + (long long)b * (4 * a.D) + blockIdx.x * 24 + col, gelu_erf(part[col * NB + b] + part[(24 + col) * NB + b] + (r ? eb1 : eb0)));
                }
            }
        }
        if (!ss_phase_end(a, flag, phase, service, lane)) return;

        // ================= FC2 + residual (K = 6144: eight segments, two per compute wave)
        if (service && lane < 6 * NB) { eb0 = Lw[SS_OFF_BFC2 + blockIdx.x * 6 + lane % 6]; eb1 = ld_x1(a.xs + (long long)(lane / 6) * a.D + blockIdx.x * 6 + lane % 6); }
        ss_stage<NB, 6144, false, false>(a.hs, nullptr, nullptr, xl, lnred);
        if (!service) {
            WMAR_SS_COMPUTE(SS_FC2, false, wb[0], 0) WMAR_SS_LOADW(SS_FC2, wb[0], Lw + SS_OFF_WFC2, 2)
            WMAR_SS_COMPUTE(SS_FC2, false, wb[1], 1) WMAR_SS_LOADW(SS_FC2, wb[1], Lw + SS_OFF_WFC2, 3)
            WMAR_SS_COMPUTE(SS_FC2, false, wb[0], 2) WMAR_SS_LOADW(SS_QKV, wb[0], Ln + SS_OFF_WQKV, 0)
            WMAR_SS_COMPUTE(SS_FC2, false, wb[1], 3) WMAR_SS_LOADW(SS_QKV, wb[1], Ln + SS_OFF_WQKV, 1)
        }
        wg_sync();
        if (service && lane < 6 * NB) {
            const int b = lane / 6, col = lane - b * 6;
            float s = part[col * NB + b];
#pragma unroll
            for (int sg = 1; sg < 8; ++sg) s += part[(sg * 6 + col) * NB + b];
            st_x1(a.xs + (long long)b * a.D + blockIdx.x * 6 + col, eb1 + (s + eb0));
        }
        if (!ss_phase_end(a, flag, phase, service, lane)) return;
    }

    // ================= ln_f + vocabulary head: 64 columns per workgroup, 2 segments x 2 halves x 8 groups of 4 (its own buffers; the
    // layer ring is dead here).  The logits leave with plain stores: the sampler is the next launch.
    ss_stage<NB, 1536, true, true>(a.xs, a.lnfw, a.lnfb, xl, lnred);
    if (!service) {
        constexpr int S = ss_xstride(NB, true), R = 4 * NB;
        const int seg = w >> 1, half = w & 1;
        float4 hb[2][12];
#define WMAR_SS_HLOAD(BUF, GI)                                                                          \
        {                                                                                               \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                             \
                int n_ = blockIdx.x * 64 + half * 32 + (GI) * 4 + c;                                    \
                n_ = n_ < a.V ? n_ : a.V - 1;                                                           \
                const float4* p_ = (const float4*)(a.whead + (long long)n_ * 1536 + seg * SG_SEG) + lane; \
                _Pragma("unroll") for (int ch = 0; ch < 3; ++ch) BUF[c * 3 + ch] = ld_nt(p_ + ch * 64); \
            }                                                                                           \
        }
        WMAR_SS_HLOAD(hb[0], 0)
        WMAR_SS_HLOAD(hb[1], 1)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x2 acc2[4][NBP];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int bp = 0; bp < NBP; ++bp) acc2[c][bp] = f32x2{0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float4* xq = (const float4*)(xl + (long long)(seg * 192 + ch * 64 + lane) * S);
#pragma unroll
                for (int bp = 0; bp < NBP; ++bp) {
                    const float4 q0 = xq[2 * bp], q1 = xq[2 * bp + 1];
                    const f32x2 xx = {q0.x, q0.y}, xy = {q0.z, q0.w}, xz = {q1.x, q1.y}, xw = {q1.z, q1.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float4 wv = hb[g & 1][c * 3 + ch];
                        f32x2 s = acc2[c][bp];
                        s = __builtin_elementwise_fma(f32x2{wv.x, wv.x}, xx, s); s = __builtin_elementwise_fma(f32x2{wv.y, wv.y}, xy, s);
                        s = __builtin_elementwise_fma(f32x2{wv.z, wv.z}, xz, s); s = __builtin_elementwise_fma(f32x2{wv.w, wv.w}, xw, s);
                        acc2[c][bp] = s;
                    }
                }
            }
            if (g + 2 < 8) WMAR_SS_HLOAD(hb[g & 1], g + 2)
            float acc[R];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[c * NB + b] = (b & 1) ? acc2[c][b >> 1].y : acc2[c][b >> 1].x;
            int ridx;
            const float tot = wave_reduce_many<R, float>(acc, lane, &ridx);
            constexpr int LBR = sg_log2p<R>();
            if ((lane & ((64 >> LBR) - 1)) == 0 && ridx < R) {
                const int c = ridx / NB, b = ridx - c * NB;
                part[(seg * 64 + half * 32 + g * 4 + c) * NB + b] = tot;
            }
        }
#undef WMAR_SS_HLOAD
    }
    wg_sync();
    for (int t = tid; t < 64 * NB; t += SS_THREADS) {
        const int b = t / 64, col = t - b * 64, n = blockIdx.x * 64 + col;
        if (n < a.V) a.logits[(long long)b * a.V + n] = part[col * NB + b] + part[(64 + col) * NB + b];
    }
}

template <int NB>
static int launch_sstep_nb(const SsArgs& a, hipStream_t st) {
    constexpr int lds = ss_lds_bytes(NB);
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)k_sstep<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    hipLaunchKernelGGL((k_sstep<NB>), dim3(SS_WGS), dim3(SS_THREADS), (size_t)lds, st, a);
    return launch_status("k_sstep");
}
static int launch_sstep(const SsArgs& a, int rows, hipStream_t st) {
    switch (rows) {
        case 1: return launch_sstep_nb<1>(a, st);
        case 2: return launch_sstep_nb<2>(a, st);
        case 3: return launch_sstep_nb<3>(a, st);
        case 4: return launch_sstep_nb<4>(a, st);
        case 5: return launch_sstep_nb<5>(a, st);
        default: set_error("k_sstep: %d rows (1..5)", rows); return WMAR_EINVAL;
    }
}
// workgroups of k_sstep<NB> that fit one compute unit (0 on error): all 256 must be resident at once
template <int NB>
static int sstep_blocks_per_cu_nb() {
    int nb = 0;
    (void)hipFuncSetAttribute((const void*)k_sstep<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, ss_lds_bytes(NB));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sstep<NB>, SS_THREADS, (size_t)ss_lds_bytes(NB)) != hipSuccess) return 0;
    return nb;
}
static int sstep_blocks_per_cu() {
    const int v[5] = {sstep_blocks_per_cu_nb<1>(), sstep_blocks_per_cu_nb<2>(), sstep_blocks_per_cu_nb<3>(), sstep_blocks_per_cu_nb<4>(), sstep_blocks_per_cu_nb<5>()};
    int m = v[0];
    for (int i = 1; i < 5; ++i) m = v[i] < m ? v[i] : m;
    return m;
}

#pragma clang fp contract(fast)

}  // namespace wmar
