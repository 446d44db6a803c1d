// Dev tool: skinny fp32 GEMM (64 rows) on the bf16 matrix pipe with LOADER waves.  out[64 x N] = X[64 x K] W[N x K]^T.
// Workgroup = 4 consumer waves + 1 weight-loader wave + 1 activation-loader wave.  The loaders do nothing but issue LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave instruction) into two LDS rings and keep them full; the consumers only read LDS, split the
// fp32 weights into bf16 pieces (VALU) and issue MFMAs -- they never touch the vector-memory pipeline, so the weight stream does not
// depend on where the MFMA stream is (and vice versa).  One raw s_barrier per ring slot hands a landed slot to the consumers and a
// drained slot back to the loaders; DMAs stay in flight across the barrier (counted vmcnt in the loader waves only).
//   T = column tiles of 32 per workgroup (1, 2, 4); the 4 consumers are T tiles x KQ = 4 / T k-steps of a slot.
//   slot = KQ k-steps of 16: weights 8 KiB (ring of NSW slots), activation pieces KQ x 6 KiB (ring of NSX slots).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifndef LX_NT
#define LX_NT 2          // aux of the weight DMA: 2 = nt, 0 = default policy
#endif
// ABL (template): 1: no MFMAs, 2: no weight DMA, 4: no activation DMA
#ifndef LX_ROT
#define LX_ROT 1         // every workgroup starts its K walk at a different slot (all of them read the SAME activation slice)
#endif

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float fsub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(a, b);
    const float ra = fsub(a, __uint_as_float(h << 16)), rb = fsub(b, __uint_as_float(h & 0xffff0000u));
    m = pk_bf16(ra, rb);
    const float sa = fsub(ra, __uint_as_float(m << 16)), sb = fsub(rb, __uint_as_float(m & 0xffff0000u));
    l = pk_bf16(sa, sb);
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
    unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    split2(a.x, a.y, h0, m0, l0);
    split2(a.z, a.w, h1, m1, l1);
    split2(b.x, b.y, h2, m2, l2);
    split2(b.z, b.w, h3, m3, l3);
    const u32x4 uh = {h0, h1, h2, h3}, um = {m0, m1, m2, m3}, ul = {l0, l1, l2, l3};
    h = __builtin_bit_cast(bf16x8, uh); m = __builtin_bit_cast(bf16x8, um); l = __builtin_bit_cast(bf16x8, ul);
}

// Wq[tile][ku][half][lane] float4: lane l holds W[n = 32 tile + l % 32][k = 16 ku + 8 (l / 32) + 4 half + 0..3]
__global__ void k_pack_w(const float* __restrict__ W, float4* __restrict__ Wq, int N, int K) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KU = K / 16;
    if (idx >= (long long)(N / 32) * KU * 128) return;
    const int lane = idx & 63, half = (idx >> 6) & 1;
    const long long r = idx >> 7;
    const int ku = r % KU, tile = r / KU;
    const float* p = W + (long long)(tile * 32 + (lane & 31)) * K + ku * 16 + 8 * (lane >> 5) + 4 * half;
    Wq[idx] = make_float4(p[0], p[1], p[2], p[3]);
}
// Xq[ku][mt][piece][lane] uint4 (8 bf16): lane l holds X[m = 32 mt + l % 32][k = 16 ku + 8 (l / 32) + 0..7]
__global__ void k_pack_x(const float* __restrict__ X, u32x4* __restrict__ Xq, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K / 16 * 128) return;
    const int lane = idx & 63, mt = (idx >> 6) & 1, ku = idx >> 7;
    const float* p = X + (long long)(mt * 32 + (lane & 31)) * K + ku * 16 + 8 * (lane >> 5);
    bf16x8 h, m, l;
    const f32x4 a = {p[0], p[1], p[2], p[3]}, b = {p[4], p[5], p[6], p[7]};
    split8(a, b, h, m, l);
    u32x4* o = Xq + ((long long)(ku * 2 + mt) * 3) * 64 + lane;
    o[0] = __builtin_bit_cast(u32x4, h); o[64] = __builtin_bit_cast(u32x4, m); o[128] = __builtin_bit_cast(u32x4, l);
}

struct LxArgs {
    const float4* Wq; const u32x4* Xq; float4* out; int KU; int S; long long slab_stride; unsigned long long* trace;
};

#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#define VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")

template <int T, int NSW, int NSX, int LX_ABL, int NLW, int NLX>
__global__ __launch_bounds__((8 + NLW + NLX) * 64) void k_lx(LxArgs a) {
    constexpr int KQ = 4 / T;
    constexpr int WSLOT = 4 * 2048, XSLOT = KQ * 6144;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wring = smem;
    char* xring = smem + NSW * WSLOT;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = (int)blockIdx.x / a.S, ks = (int)blockIdx.x % a.S;
    const int nsl = a.KU / KQ;                                   // slots over the whole K
    const int s0 = (int)((long long)ks * nsl / a.S), s1 = (int)((long long)(ks + 1) * nsl / a.S);
    const int n = s1 - s0;                                       // >= 1
    const int u0 = s0 * KQ;
    const int rot = LX_ROT ? (int)((grp * 5u + ks * 3u) % (unsigned)n) : 0;
#define LX_SLOT(J) (((J) + rot) >= n ? (J) + rot - n : (J) + rot)        // K-walk position of the J-th fill

    if (w >= 8 && w < 8 + NLW) {
        // ---------------------------------------------------------------- weight loaders: DMA d of a fill by loader d % NLW
        const int lw = w - 8;
        constexpr int PF = 8 / NLW;      // DMAs per fill and wave
        auto issue = [&](int j) {
            if (LX_ABL & 2) return;
            char* dst = wring + (j % NSW) * WSLOT;
#pragma unroll
            for (int e = 0; e < PF; ++e) {
                const int d = e * NLW + lw, c = d >> 1, hf = d & 1;
                const int t = c % T, kq = c / T;
                const float4* src = a.Wq + ((long long)(grp * T + t) * a.KU + u0 + (long long)LX_SLOT(j) * KQ + kq) * 128 + hf * 64 + lane;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + d * 1024), 16, 0, LX_NT);
            }
        };
        for (int j = 0; j < NSW - 1 && j < n; ++j) issue(j);
        for (int i = 0; i < n; ++i) {
            if (i + NSW - 2 < n) VMCNT(PF * (NSW - 2)); else VMCNT(0);
            __builtin_amdgcn_s_barrier();
            if (i + NSW - 1 < n) issue(i + NSW - 1);
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    if (w >= 8 + NLW) {
        // ---------------------------------------------------------------- activation loaders
        const int lx = w - 8 - NLW;
        constexpr int PF = 6 * KQ / NLX;
        auto issue = [&](int j) {
            if (LX_ABL & 4) return;
            char* dst = xring + (j % NSX) * XSLOT;
            const u32x4* src = a.Xq + (long long)(u0 + (long long)LX_SLOT(j) * KQ) * 384 + lane;
#pragma unroll
            for (int e = 0; e < PF; ++e) {
                const int q = e * NLX + lx;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + q * 64), (lptr_t)(dst + q * 1024), 16, 0, 0);
            }
        };
        for (int j = 0; j < NSX - 1 && j < n; ++j) issue(j);
        for (int i = 0; i < n; ++i) {
            if (i + NSX - 2 < n) VMCNT(PF * (NSX - 2)); else VMCNT(0);
            __builtin_amdgcn_s_barrier();
            if (i + NSX - 1 < n) issue(i + NSX - 1);
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    // -------------------------------------------------------------------- consumers
    // Two groups of four (one wave of each group per SIMD): group g takes the slots j = g (mod 2).  A wave reads and splits its slot
    // between barrier j and barrier j+1 and issues the slot's MFMAs AFTER barrier j+1 -- while the other group's wave on the same
    // SIMD reads and splits slot j+1: the matrix pipe of a SIMD alternates between its two waves, no software pipelining needed.
    const int g = w >> 2, c = w & 3;
    const int t = c % T, kq = c / T;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int j = 0;
    if (g == 1) { __builtin_amdgcn_s_barrier(); j = 1; }
    for (; j < n; j += 2) {
        __builtin_amdgcn_s_barrier();                                   // barrier j: slot j has landed
        f32x4 wr[2];
        u32x4 xr[6];
        const char* wp_ = wring + (j % NSW) * WSLOT + c * 2048 + lane * 16;
        wr[0] = *(const f32x4*)wp_; wr[1] = *(const f32x4*)(wp_ + 1024);
        const char* xp_ = xring + (j % NSX) * XSLOT + kq * 6144 + lane * 16;
#pragma unroll
        for (int q = 0; q < 6; ++q) xr[q] = *(const u32x4*)(xp_ + q * 1024);
        bf16x8 ph, pm, pl;
        split8(wr[0], wr[1], ph, pm, pl);
        bf16x8 x[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = __builtin_bit_cast(bf16x8, xr[q]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                   // barrier j+1: the slot is drained (and slot j+1 has landed)
        __builtin_amdgcn_sched_barrier(0);
        if (!(LX_ABL & 1)) {
            MFMA(pl, x[0], acc[0]); MFMA(pl, x[3], acc[1]);
            MFMA(ph, x[2], acc[0]); MFMA(ph, x[5], acc[1]);
            MFMA(pm, x[1], acc[0]); MFMA(pm, x[4], acc[1]);
            MFMA(pm, x[0], acc[0]); MFMA(pm, x[3], acc[1]);
            MFMA(ph, x[1], acc[0]); MFMA(ph, x[4], acc[1]);
            MFMA(ph, x[0], acc[0]); MFMA(ph, x[3], acc[1]);
        } else {
            acc[0][0] += (float)ph[0] + (float)pm[1] + (float)pl[2] + (float)x[0][0] + (float)x[5][1];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (((n & 1) == 0) == (g == 0)) __builtin_amdgcn_s_barrier();      // every wave passes n + 1 barriers; after the last all slots are drained
    // consumers of one tile (2 KQ of them) meet in LDS, fixed order
    float4* red = (float4*)smem;           // [8][8][64] float4 = 64 KiB
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            red[(w * 8 + mt * 4 + q) * 64 + lane] = make_float4(acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
    __builtin_amdgcn_s_barrier();          // the loaders have left: a barrier of the eight consumers
    float4* out = a.out + (long long)ks * a.slab_stride;
    // the 2 KQ waves of tile t share its 8 (row tile, register group) rows
    for (int r = kq * 2 + g; r < 8; r += 2 * KQ) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < 2 * KQ; ++o) {            // contributor o: group o % 2, k-step position o / 2
            const float4 q = red[(((o & 1) * 4 + t + (o >> 1) * T) * 8 + r) * 64 + lane];
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        const int mt = r >> 2, q4 = r & 3;
        out[((long long)((grp * T + t) * 4 + q4) * 2 + mt) * 64 + lane] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// k_sx: the consumers stream their own weights (plain non-temporal loads into a register ring, R slots ahead, independent of the
// workgroup's barriers); only the activation pieces go through LDS (one DMA loader wave, ring of NSX slots, one barrier per slot).
template <int T, int NSX, int R, int LX_ABL>
__global__ __launch_bounds__(576) void k_sx(LxArgs a) {
    constexpr int KQ = 4 / T;
    constexpr int XSLOT = KQ * 6144;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xring = smem;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = (int)blockIdx.x / a.S, ks = (int)blockIdx.x % a.S;
    const int nsl = a.KU / KQ;
    const int s0 = (int)((long long)ks * nsl / a.S), s1 = (int)((long long)(ks + 1) * nsl / a.S);
    const int n = s1 - s0;
    const int u0 = s0 * KQ;
    const int rot = LX_ROT ? (int)((grp * 5u + ks * 3u) % (unsigned)n) : 0;
    if (w == 8) {
        constexpr int PF = 6 * KQ;
        auto issue = [&](int j) {
            if (LX_ABL & 4) return;
            char* dst = xring + (j % NSX) * XSLOT;
            const u32x4* src = a.Xq + (long long)(u0 + (long long)LX_SLOT(j) * KQ) * 384 + lane;
#pragma unroll
            for (int q = 0; q < PF; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(src + q * 64), (lptr_t)(dst + q * 1024), 16, 0, 0);
        };
        for (int j = 0; j < NSX - 1 && j < n; ++j) issue(j);
        for (int i = 0; i < n; ++i) {
            if (i + NSX - 2 < n) VMCNT(PF * (NSX - 2)); else VMCNT(0);
            __builtin_amdgcn_s_barrier();
            if (i + NSX - 1 < n) issue(i + NSX - 1);
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    const int g = w >> 2, c = w & 3;
    const int t = c % T, kq = c / T;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // this wave's slots: j = g, g + 2, ...; its weights of slot j: k-step u0 + LX_SLOT(j) * KQ + kq of tile grp * T + t
    const float4* wbase = a.Wq + ((long long)(grp * T + t) * a.KU + u0 + kq) * 128 + lane;
    f32x4 wr[R][2];
#define SX_LOADW(SLOT_, J)                                                                          \
    {                                                                                               \
        const int jj_ = (J) < n ? (J) : g;      /* past the end: a harmless re-read of the first slot */ \
        const f32x4* p_ = (const f32x4*)(wbase + (long long)LX_SLOT(jj_) * KQ * 128);               \
        if (!(LX_ABL & 2)) { wr[SLOT_][0] = __builtin_nontemporal_load(p_); wr[SLOT_][1] = __builtin_nontemporal_load(p_ + 64); } \
        else { wr[SLOT_][0] = f32x4{1.f, 2.f, 3.f, (float)jj_}; wr[SLOT_][1] = wr[SLOT_][0]; }     \
    }
#pragma unroll
    for (int r = 0; r < R; ++r) SX_LOADW(r, g + 2 * r)
    __builtin_amdgcn_sched_barrier(0);
    int j = g;
    if (g == 1) __builtin_amdgcn_s_barrier();
#ifdef SX_TRACE
#define SX_T(I) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tacc[I] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); }
#define SX_TV(I) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (R - 1)) : "memory"); SX_T(I) }
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tlast;
#else
#define SX_T(I)
#define SX_TV(I)
#endif
#define SX_STEP(SLOT_, LOAD_)                                                                       \
    {                                                                                               \
        SX_T(0)                                                                                     \
        __builtin_amdgcn_s_barrier();                                                               \
        SX_T(1)                                                                                     \
        if (LOAD_) { SX_TV(2) }                                                               \
        u32x4 xr[6];                                                                                \
        const char* xp_ = xring + (j % NSX) * XSLOT + kq * 6144 + lane * 16;                        \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) xr[q] = *(const u32x4*)(xp_ + q * 1024);      \
        bf16x8 ph, pm, pl;                                                                          \
        split8(wr[SLOT_][0], wr[SLOT_][1], ph, pm, pl);                                             \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        SX_T(3)                                                                                     \
        if (LOAD_) SX_LOADW(SLOT_, j + 2 * R)                                                       \
        SX_T(4)                                                                                     \
        bf16x8 x[6];                                                                                \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) x[q] = __builtin_bit_cast(bf16x8, xr[q]);     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        SX_T(5)                                                                                     \
        __builtin_amdgcn_s_barrier();                                                               \
        SX_T(6)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        if (!(LX_ABL & 1)) {                                                                        \
            MFMA(pl, x[0], acc[0]); MFMA(pl, x[3], acc[1]);                                         \
            MFMA(ph, x[2], acc[0]); MFMA(ph, x[5], acc[1]);                                         \
            MFMA(pm, x[1], acc[0]); MFMA(pm, x[4], acc[1]);                                         \
            MFMA(pm, x[0], acc[0]); MFMA(pm, x[3], acc[1]);                                         \
            MFMA(ph, x[1], acc[0]); MFMA(ph, x[4], acc[1]);                                         \
            MFMA(ph, x[0], acc[0]); MFMA(ph, x[3], acc[1]);                                         \
        } else {                                                                                    \
            acc[0][0] += (float)ph[0] + (float)pm[1] + (float)pl[2] + (float)x[0][0] + (float)x[5][1]; \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        j += 2;                                                                                     \
    }
    // whole rounds of R steps: no branches inside (hipcc waits for ALL outstanding loads at a control-flow merge)
    while (j + 2 * (R - 1) < n) {
        SX_STEP(0, true)
        if (R > 1) { SX_STEP(1 % R, true) }
        if (R > 2) { SX_STEP(2 % R, true) }
        if (R > 3) { SX_STEP(3 % R, true) }
        if (R > 4) { SX_STEP(4 % R, true) }
        if (R > 5) { SX_STEP(5 % R, true) }
    }
    if (j < n) { SX_STEP(0, false) }
    if (R > 2 && j < n) { SX_STEP(1 % R, false) }
    if (R > 3 && j < n) { SX_STEP(2 % R, false) }
    if (R > 4 && j < n) { SX_STEP(3 % R, false) }
    if (R > 5 && j < n) { SX_STEP(4 % R, false) }
#ifdef SX_TRACE
    SX_T(7)
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((long long)blockIdx.x * 8 + w) * 10;
        for (int i = 0; i < 8; ++i) tr[i] = tacc[i];
        tr[8] = tstart; tr[9] = tlast;
    }
#endif
    if (((n & 1) == 0) == (g == 0)) __builtin_amdgcn_s_barrier();
    float4* red = (float4*)smem;           // [8][8][64] float4 = 64 KiB
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            red[(w * 8 + mt * 4 + q) * 64 + lane] = make_float4(acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
    __builtin_amdgcn_s_barrier();
    float4* out = a.out + (long long)ks * a.slab_stride;
    for (int r = kq * 2 + g; r < 8; r += 2 * KQ) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < 2 * KQ; ++o) {
            const float4 q = red[(((o & 1) * 4 + t + (o >> 1) * T) * 8 + r) * 64 + lane];
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        const int mt = r >> 2, q4 = r & 3;
        out[((long long)((grp * T + t) * 4 + q4) * 2 + mt) * 64 + lane] = v;
    }
}
#define LX_KERNEL_SX 1

// ---------------------------------------------------------------------------------------------------------------------------------
// k_rx: the workgroup's whole activation slice is RESIDENT in LDS (DMA'd first thing, ahead of every weight request: the CU's vector
// memory pipeline returns data in request order, so an L2 hit queued behind HBM misses takes HBM latency).  After one barrier the
// eight consumer waves are independent streams: own k-steps, own weight ring in registers, activation operands read from LDS.
// No barrier in the main loop; the two waves of a SIMD interleave their split (VALU) and MFMA phases by themselves.
template <int T, int R, int LX_ABL>
__global__ __launch_bounds__(512) void k_rx(LxArgs a) {
    constexpr int NQ = 8 / T;               // waves per column tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = (int)blockIdx.x / a.S, ks = (int)blockIdx.x % a.S;
    const int u0 = (int)((long long)ks * a.KU / a.S), u1 = (int)((long long)(ks + 1) * a.KU / a.S);
    const int nk = u1 - u0;                 // k-steps of the slice
    // 1. the activation slice: 6 nk pieces of 1 KiB, piece p by wave p % 8
    if (!(LX_ABL & 4)) {
        const u32x4* src = a.Xq + (long long)u0 * 384 + lane;
        for (int p = w; p < 6 * nk; p += 8)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (long long)p * 64), (lptr_t)(smem + p * 1024), 16, 0, 0);
    }
    const int t = w % T, q = w / T;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float dummy = 0.f;
    // this wave's k-steps: q, q + NQ, ... (walk rotated per workgroup)
    const int mine = (nk - q + NQ - 1) / NQ;
    const float4* wbase = a.Wq + ((long long)(grp * T + t) * a.KU + u0) * 128 + lane;
    f32x4 wr[R][2];
#define RX_LOADW(SLOT_, I)                                                                          \
    {                                                                                               \
        const int ii_ = (I) < mine ? (I) : 0;                                                       \
        const f32x4* p_ = (const f32x4*)(wbase + (long long)(q + ii_ * NQ) * 128);                  \
        if (!(LX_ABL & 2)) { wr[SLOT_][0] = __builtin_nontemporal_load(p_); wr[SLOT_][1] = __builtin_nontemporal_load(p_ + 64); } \
        else { wr[SLOT_][0] = f32x4{0.37f * lane + 0.11f, -1.3f * lane + 0.7f, 0.013f * lane - 2.1f, 3.3f - 0.21f * lane + ii_}; wr[SLOT_][1] = wr[SLOT_][0] * 1.7f; }     \
    }
#pragma unroll
    for (int r = 0; r < R; ++r) RX_LOADW(r, r)
    __builtin_amdgcn_sched_barrier(0);
#ifdef SX_TRACE
    unsigned long long racc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long rstart = __builtin_amdgcn_s_memtime();
#define RX_T(I) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); racc[I] += t_ - rlast; rlast = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define RX_T(I)
#endif
    __builtin_amdgcn_s_waitcnt((15 << 8) | (7 << 4) | ((2 * R) & 15) | (((2 * R) >> 4) << 14));      // vmcnt(2R): the DMAs were issued first, they have landed
    __builtin_amdgcn_s_barrier();
#ifdef SX_TRACE
    unsigned long long rlast = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    racc[5] = rlast - rstart;
#endif
    int i = 0;
#define RX_STEP(SLOT_, LOAD_)                                                                       \
    {                                                                                               \
        if (LOAD_) { RX_T(0) __builtin_amdgcn_s_waitcnt((15 << 8) | (7 << 4) | ((2 * (R - 1)) & 15)); RX_T(1) }  \
        u32x4 xr[6];                                                                                \
        const char* xp_ = smem + (q + i * NQ) * 6144 + lane * 16;                                   \
        _Pragma("unroll") for (int e = 0; e < 6; ++e) xr[e] = *(const u32x4*)(xp_ + e * 1024);      \
        bf16x8 ph, pm, pl;                                                                          \
        split8(wr[SLOT_][0], wr[SLOT_][1], ph, pm, pl);                                             \
        if (LX_ABL & 8) {      /* loads are waited for and consumed, but the MFMAs run on lane-dependent constants */ \
            dummy += wr[SLOT_][0].x + wr[SLOT_][1].w;                                               \
            const f32x4 ca = {0.37f * lane + 0.11f, -1.3f * lane + 0.7f, 0.013f * lane - 2.1f, 3.3f - 0.21f * lane}; \
            const f32x4 cb = {1.37f * lane + 0.31f, -0.3f * lane + 1.7f, 0.113f * lane - 0.1f, 1.3f - 0.71f * lane}; \
            split8(ca, cb, ph, pm, pl);                                                             \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        RX_T(2)                                                                                     \
        if (LOAD_) RX_LOADW(SLOT_, i + R)                                                           \
        RX_T(3)                                                                                     \
        bf16x8 x[6];                                                                                \
        _Pragma("unroll") for (int e = 0; e < 6; ++e) x[e] = __builtin_bit_cast(bf16x8, xr[e]);     \
        if (!(LX_ABL & 1)) {                                                                        \
            MFMA(pl, x[0], acc[0]); MFMA(pl, x[3], acc[1]);                                         \
            MFMA(ph, x[2], acc[0]); MFMA(ph, x[5], acc[1]);                                         \
            MFMA(pm, x[1], acc[0]); MFMA(pm, x[4], acc[1]);                                         \
            MFMA(pm, x[0], acc[0]); MFMA(pm, x[3], acc[1]);                                         \
            MFMA(ph, x[1], acc[0]); MFMA(ph, x[4], acc[1]);                                         \
            MFMA(ph, x[0], acc[0]); MFMA(ph, x[3], acc[1]);                                         \
        } else {                                                                                    \
            acc[0][0] += (float)ph[0] + (float)pm[1] + (float)pl[2] + (float)x[0][0] + (float)x[5][1]; \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        ++i;                                                                                        \
    }
    while (i + R - 1 < mine) {
        RX_STEP(0, true)
        if (R > 1) { RX_STEP(1 % R, true) }
        if (R > 2) { RX_STEP(2 % R, true) }
        if (R > 3) { RX_STEP(3 % R, true) }
        if (R > 4) { RX_STEP(4 % R, true) }
        if (R > 5) { RX_STEP(5 % R, true) }
        if (R > 6) { RX_STEP(6 % R, true) }
        if (R > 7) { RX_STEP(7 % R, true) }
    }
    if (i < mine) { RX_STEP(0, false) }
    if (R > 2 && i < mine) { RX_STEP(1 % R, false) }
    if (R > 3 && i < mine) { RX_STEP(2 % R, false) }
    if (R > 4 && i < mine) { RX_STEP(3 % R, false) }
    if (R > 5 && i < mine) { RX_STEP(4 % R, false) }
    if (R > 6 && i < mine) { RX_STEP(5 % R, false) }
    if (R > 7 && i < mine) { RX_STEP(6 % R, false) }
    if (dummy == 12345.678f) acc[0][0] += 1.f;
#ifdef SX_TRACE
    RX_T(4)
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((long long)blockIdx.x * 8 + w) * 10;
        for (int e = 0; e < 6; ++e) tr[e] = racc[e];
        tr[8] = rstart; tr[9] = rlast;
    }
#endif
    __builtin_amdgcn_s_barrier();          // every wave is done with the activation slice: LDS becomes reduction scratch
    float4* red = (float4*)smem;           // [8][8][64] float4 = 64 KiB
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            red[(w * 8 + mt * 4 + e) * 64 + lane] = make_float4(acc[mt][4 * e], acc[mt][4 * e + 1], acc[mt][4 * e + 2], acc[mt][4 * e + 3]);
    __builtin_amdgcn_s_barrier();
    float4* out = a.out + (long long)ks * a.slab_stride;
    for (int r = q; r < 8; r += NQ) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < NQ; ++o) {
            const float4 e = red[((t + o * T) * 8 + r) * 64 + lane];
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        }
        const int mt = r >> 2, q4 = r & 3;
        out[((long long)((grp * T + t) * 4 + q4) * 2 + mt) * 64 + lane] = v;
    }
}
template <int T, int NSW, int NSX, int ABL = 0, int NLW = 1, int NLX = 1>
static void run(const char* name, int N, int K, int S, hipStream_t st) {
    const int NL = 12;
    constexpr int KQ = 4 / T;
    size_t lds = (size_t)NSW * 8192 + (size_t)NSX * KQ * 6144;
    if (NLW == 0) { lds = (size_t)NSX * KQ * 6144; if (lds < 65536) lds = 65536; }
    if (NLW == 0 && NLX == 0) { lds = (size_t)((K / 16 + S - 1) / S + 1) * 6144; if (lds < 65536) lds = 65536; }
    std::vector<float> hW((size_t)N * K), hX((size_t)64 * K);
    srand(1);
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.08f;
    for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    float *W, *X; float4* out; u32x4* Xq;
    (void)hipMalloc(&W, hW.size() * 4); (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&X, hX.size() * 4); (void)hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&Xq, (size_t)K / 16 * 6 * 64 * 16);
    const size_t slab = (size_t)N * 64 / 4;
    (void)hipMalloc(&out, slab * S * 16);
    hipLaunchKernelGGL(k_pack_x, dim3((K / 16 * 128 + 255) / 256), dim3(256), 0, st, X, Xq, K);
    std::vector<float4*> Wq(NL);
    for (int l = 0; l < NL; ++l) {
        (void)hipMalloc(&Wq[l], (size_t)N * K * 4);
        const long long total = (long long)(N / 32) * (K / 16) * 128;
        hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, Wq[l], N, K);
    }
    (void)hipStreamSynchronize(st);
    if constexpr (NLW == 0 && NLX == 0) (void)hipFuncSetAttribute((const void*)k_rx<T, NSW, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else if constexpr (NLW == 0) (void)hipFuncSetAttribute((const void*)k_sx<T, NSX, NSW, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else (void)hipFuncSetAttribute((const void*)k_lx<T, NSW, NSX, ABL, NLW, NLX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = N / 32 / T * S;
    unsigned long long* tr; (void)hipMalloc(&tr, 320 * 8 * 10 * 8); (void)hipMemset(tr, 0, 320 * 8 * 10 * 8);
    LxArgs a{}; a.Xq = Xq; a.out = out; a.KU = K / 16; a.S = S; a.slab_stride = slab; a.trace = tr;
    hipGraph_t graph; hipGraphExec_t exec;
    hipStream_t cs; (void)hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    (void)hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    for (int rep = 0; rep < 4; ++rep)
        for (int l = 0; l < NL; ++l) { a.Wq = Wq[l];
            if constexpr (NLW == 0 && NLX == 0) hipLaunchKernelGGL((k_rx<T, NSW, ABL>), dim3(grid), dim3(512), lds, cs, a);
            else if constexpr (NLW == 0) hipLaunchKernelGGL((k_sx<T, NSX, NSW, ABL>), dim3(grid), dim3(576), lds, cs, a);
            else hipLaunchKernelGGL((k_lx<T, NSW, NSX, ABL, NLW, NLX>), dim3(grid), dim3((8 + NLW + NLX) * 64), lds, cs, a); }
    (void)hipStreamEndCapture(cs, &graph);
    (void)hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, cs);
        (void)hipGraphLaunch(exec, cs);
        (void)hipEventRecord(e1, cs); (void)hipStreamSynchronize(cs);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double us = best * 1000.0 / (4 * NL);
    printf("%-20s abl %d ld %d+%d N=%d K=%d: %3d workgroups (%d tiles x %d k), rings %d/%d (%zu KiB LDS): %6.2f us per launch (%.1f MB -> %.2f TB/s) %s\n", name, ABL, NLW, NLX, N, K, grid, T,
           K / S, NSW, NSX, lds / 1024, us, N * (double)K * 4 / 1e6, N * (double)K * 4 / us / 1e6, hipGetErrorString(hipGetLastError()));
#ifdef SX_TRACE
    if (NLW == 0 && NLX == 0) {
        std::vector<unsigned long long> h((size_t)grid * 8 * 10); (void)hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
        double av[10] = {0}; double span = 0;
        for (int i = 0; i < grid * 8; ++i) { for (int k = 0; k < 6; ++k) av[k] += (double)h[(size_t)i * 10 + k] / (grid * 8); span += (double)(h[(size_t)i * 10 + 9] - h[(size_t)i * 10 + 8]) / (grid * 8); }
        printf("    ticks per wave: until X landed + barrier %.0f | mfma issue (prev step) %.0f | W wait %.0f | ds_read+split %.0f | load issue %.0f | tail %.0f | span %.0f\n",
               av[5], av[0], av[1], av[2], av[3], av[4], span);
    } else if (NLW == 0) {
        std::vector<unsigned long long> h((size_t)grid * 8 * 10); (void)hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
        double av[10] = {0}; double span = 0;
        for (int i = 0; i < grid * 8; ++i) { for (int k = 0; k < 8; ++k) av[k] += (double)h[(size_t)i * 10 + k] / (grid * 8); span += (double)(h[(size_t)i * 10 + 9] - h[(size_t)i * 10 + 8]) / (grid * 8); }
        printf("    ticks per wave (100 MHz?): pre-barrier1 %.0f | barrier1 %.0f | vmcnt wait %.0f | ds_read+split %.0f | load issue %.0f | lgkm wait %.0f | barrier2 %.0f | mfma+tail %.0f | span %.0f\n",
               av[0], av[1], av[2], av[3], av[4], av[5], av[6], av[7], span);
    }
#endif
    std::vector<float> o(slab * 4 * S);
    (void)hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    double e_bx = 0, e_f32 = 0, mag = 0;
    for (int nn = 0; nn < N; nn += 13)
        for (int m = 0; m < 64; ++m) {
            double r = 0; float f = 0.f;
            for (int k = 0; k < K; ++k) { r += (double)hX[(size_t)m * K + k] * hW[(size_t)nn * K + k]; f = fmaf(hX[(size_t)m * K + k], hW[(size_t)nn * K + k], f); }
            float got = 0.f;
            for (int s = 0; s < S; ++s) got += o[s * slab * 4 + (((size_t)(nn >> 3) * 2 + (m >> 5)) * 64 + (m & 31) + 32 * ((nn >> 2) & 1)) * 4 + (nn & 3)];
            e_bx = fmax(e_bx, fabs(got - r)); e_f32 = fmax(e_f32, fabs(f - r)); mag = fmax(mag, fabs(r));
        }
    if (!ABL) printf("    max |lx - fp64| = %.3e, max |fp32 fma chain - fp64| = %.3e, max |value| = %.3f\n", e_bx, e_f32, mag);
    for (auto p : Wq) (void)hipFree(p);
    (void)hipFree(W); (void)hipFree(X); (void)hipFree(Xq); (void)hipFree(out);
    (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph); (void)hipStreamDestroy(cs);
}

#define RUNR(T, R, NAME, N, K, S) run<T, R, 0, 0, 0, 0>(NAME, N, K, S, st); run<T, R, 0, 1, 0, 0>(NAME, N, K, S, st); run<T, R, 0, 2, 0, 0>(NAME, N, K, S, st); run<T, R, 0, 4, 0, 0>(NAME, N, K, S, st); run<T, R, 0, 6, 0, 0>(NAME, N, K, S, st); run<T, R, 0, 8, 0, 0>(NAME, N, K, S, st); run<T, R, 0, 12, 0, 0>(NAME, N, K, S, st);
int main() {
    hipStream_t st; (void)hipStreamCreate(&st);
    RUNR(4, 4, "rx qkv n128 S=7", 4608, 1536, 7)
    RUNR(4, 4, "rx fc2 n128 S=21", 1536, 6144, 21)
    return 0;
}
