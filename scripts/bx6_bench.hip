// Dev tool: fp32 GEMM on the bf16 matrix pipe.  Every fp32 operand is split EXACTLY into three bf16 pieces (w = h + m + l, 8 + 8 + 8
// significand bits); the six products of combined order <= 2 (hh, hm, mh, hl, lh, mm) are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (32 cycles for 32x32x16, against 8 x 64 for the fp32-input MFMA).  The dropped products (ml, lm, ll) are
// at most 2^-24 of |x w| (worst case; typically 2^-27): no more than the fp32 rounding of the product itself.  Weights stay fp32 in HBM (no extra bytes) and are split in
// registers; the activation arrives pre-split from its producer.
// Shape: out[64 x N] = X[64 x K] W[N x K]^T, N = 6144, K = 1536 (FC1 of the Taming GPT), raw sums, against an fp64 host reference.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef BX_ABL
#define BX_ABL 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef BX_SCHED
#define BX_SCHED 1
#endif
#ifndef BX_ROT
#define BX_ROT 0
#endif
#ifndef BX_NT
#define BX_NT 1
#endif
#ifndef BX_STAMP
#define BX_STAMP 0
#endif
#ifndef BX_LAST
#define BX_LAST 0
#endif
#ifndef BX_OCC
#define BX_OCC 1
#endif
#ifndef BX_WR
#define BX_WR 1
#endif
#ifndef BX_XR
#define BX_XR 1
#endif
#define BX_RING (BX_WR * 10 + BX_XR)

__device__ __forceinline__ float4 ld_nt4(const float4* p) { const f32x4 v = BX_NT ? __builtin_nontemporal_load((const f32x4*)p) : *(const f32x4*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
// plain v_sub_f32: the SLP vectoriser would pair these into v_pk_add_f32, which costs more beside MFMAs
__device__ __forceinline__ float fsub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// two fp32 -> packed (h, m, l) bf16 pairs, exact: a = h + m + l
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(a, b);
    const float ra = fsub(a, __uint_as_float(h << 16)), rb = fsub(b, __uint_as_float(h & 0xffff0000u));
    m = pk_bf16(ra, rb);
    const float sa = fsub(ra, __uint_as_float(m << 16)), sb = fsub(rb, __uint_as_float(m & 0xffff0000u));
    l = pk_bf16(sa, sb);
}
__device__ __forceinline__ void split8(const float4 a, const float4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
#if BX_ABL & 16
    const u32x4 q = {__float_as_uint(a.x) ^ __float_as_uint(a.y), __float_as_uint(a.z) ^ __float_as_uint(a.w), __float_as_uint(b.x) ^ __float_as_uint(b.y), __float_as_uint(b.z) ^ __float_as_uint(b.w)};
    h = __builtin_bit_cast(bf16x8, q); m = h; l = h; return;
#endif
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
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // over (K/16) * 2 * 64
    if (idx >= K / 16 * 128) return;
    const int lane = idx & 63, mt = (idx >> 6) & 1, ku = idx >> 7;
    const float* p = X + (long long)(mt * 32 + (lane & 31)) * K + ku * 16 + 8 * (lane >> 5);
    bf16x8 h, m, l;
    split8(make_float4(p[0], p[1], p[2], p[3]), make_float4(p[4], p[5], p[6], p[7]), h, m, l);
    u32x4* o = Xq + ((long long)(ku * 2 + mt) * 3) * 64 + lane;
    o[0] = __builtin_bit_cast(u32x4, h); o[64] = __builtin_bit_cast(u32x4, m); o[128] = __builtin_bit_cast(u32x4, l);
}

struct BxArgs {
    const float4* Wq; const u32x4* Xq; float4* out; int KU; int S; long long slab_stride; unsigned long long* trace;
    unsigned* cnt; float4* fin;   // BX_LAST: arrival counters per column group, final sums
};

#if BX_ABL & 16
#define MFMA(A, B, C) C[0] += (float)A[0] + (float)B[0]
#else
#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#endif

// Workgroup = (NT column tiles of 32) x (K slice blockIdx % S); its four waves split the slice's 16-k steps; slab s = raw partial sums.
template <int NT, int PER, int NW = 4>
__global__ __launch_bounds__(NW * 64, BX_OCC) void k_bx6(BxArgs a) {
    __shared__ __attribute__((aligned(16))) float4 red[NW][NT * 8][64];
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = (int)blockIdx.x / a.S, ks = (int)blockIdx.x % a.S;
    constexpr int per = PER;             // host: K / 16 / S / 4 == PER
    const int u0 = ks * (a.KU / a.S) + w * per;
    f32x16 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
#define BX_U(I) (I)
    const float4* wp = a.Wq + ((long long)grp * NT * a.KU + u0) * 128 + lane;
    const long long wt = (long long)a.KU * 128;     // next column tile
    const u32x4* xp = a.Xq + (long long)u0 * 6 * 64 + lane;
    unsigned long long ts[3 * PER + 1];
#define BX_T(I) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ts[I]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
    float4 wr[BX_WR][NT][2];
    u32x4 xr[BX_XR][6];
#define BX_LOADW(S_, U)                                                                            \
    { _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                             \
        wr[S_][t][0] = ld_nt4(wp + t * wt + (long long)(U) * 128);                                 \
        wr[S_][t][1] = ld_nt4(wp + t * wt + (long long)(U) * 128 + 64); } }
#define BX_LOADX(S_, U)                                                                            \
    { _Pragma("unroll") for (int q = 0; q < 6; ++q) xr[S_][q] = xp[(long long)(U) * 384 + q * 64]; }
    // step j's weights live in slot j % WR, its activation pieces in slot j % XR
#pragma unroll
    for (int j = 0; j < (BX_WR > BX_XR ? BX_WR : BX_XR); ++j) {
        if (j < BX_WR && j < per) BX_LOADW(j % BX_WR, j);
        if (j < BX_XR && j < per) BX_LOADX(j % BX_XR, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 ph[2][NT], pm[2][NT], pl[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) split8(wr[0][t][0], wr[0][t][1], ph[0][t], pm[0][t], pl[0][t]);
    if (BX_WR < per) BX_LOADW(0, BX_WR);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < per; ++j) {
        const int c = j & 1, n = c ^ 1;
#if BX_STAMP
        BX_T(3 * j);
#endif
        if (j + 1 < per) {
#pragma unroll
            for (int t = 0; t < NT; ++t) split8(wr[(j + 1) % BX_WR][t][0], wr[(j + 1) % BX_WR][t][1], ph[n][t], pm[n][t], pl[n][t]);
            if (j + 1 + BX_WR < per) BX_LOADW((j + 1) % BX_WR, j + 1 + BX_WR);
        }
#if BX_STAMP
        BX_T(3 * j + 1);
#endif
        bf16x8 x[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = __builtin_bit_cast(bf16x8, xr[j % BX_XR][q]);
        // x[3 mt + piece]; small products first
#pragma unroll
        for (int t = 0; t < NT; ++t) { MFMA(pl[c][t], x[0], acc[t][0]); MFMA(pl[c][t], x[3], acc[t][1]); }
#pragma unroll
        for (int t = 0; t < NT; ++t) { MFMA(ph[c][t], x[2], acc[t][0]); MFMA(ph[c][t], x[5], acc[t][1]); }
#pragma unroll
        for (int t = 0; t < NT; ++t) { MFMA(pm[c][t], x[1], acc[t][0]); MFMA(pm[c][t], x[4], acc[t][1]); }
#pragma unroll
        for (int t = 0; t < NT; ++t) { MFMA(pm[c][t], x[0], acc[t][0]); MFMA(pm[c][t], x[3], acc[t][1]); }
#pragma unroll
        for (int t = 0; t < NT; ++t) { MFMA(ph[c][t], x[1], acc[t][0]); MFMA(ph[c][t], x[4], acc[t][1]); }
#pragma unroll
        for (int t = 0; t < NT; ++t) { MFMA(ph[c][t], x[0], acc[t][0]); MFMA(ph[c][t], x[3], acc[t][1]); }
        if (j + BX_XR < per) BX_LOADX(j % BX_XR, j + BX_XR);
#if BX_SCHED
#pragma unroll
        for (int i = 0; i < 12 * NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            if (i % 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
#if BX_STAMP
        BX_T(3 * j + 2);
#endif
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                red[w][(t * 2 + i) * 4 + g][lane] = make_float4(acc[t][i][4 * g], acc[t][i][4 * g + 1], acc[t][i][4 * g + 2], acc[t][i][4 * g + 3]);
    __syncthreads();
    // wave w sums rows w, w + NW, ... of the NT * 8 (tile, row tile, register group) rows over the NW K parts, in fixed order
    float4* out = a.out + (long long)ks * a.slab_stride;
#pragma unroll
    for (int r = 0; r < NT * 8 / NW; ++r) {
        const int row = r * NW + w, t = row >> 3, i = (row >> 2) & 1, g = row & 3;
        float4 v = red[0][row][lane];
#pragma unroll
        for (int o = 1; o < NW; ++o) { const float4 q = red[o][row][lane]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        out[((long long)((grp * NT + t) * 4 + g) * 2 + i) * 64 + lane] = v;
    }
#if BX_LAST
    // the last workgroup of a column group to arrive sums the S slabs in slab order (the same result whoever is last)
    __shared__ unsigned last_flag;
#if BX_LAST == 2
    // slabs and counters live in UNCACHED (MTYPE_UC) device memory: stores are acknowledged by the memory side, so waiting for
    // this wave's stores + a relaxed device-scope atomic is the release; the acquire is an L1/L2 invalidate of non-local lines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    __threadfence();
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
#if BX_LAST == 2
        const unsigned old = __hip_atomic_fetch_add(a.cnt + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == (unsigned)a.S - 1;
        if (last_flag) __hip_atomic_store(a.cnt + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        const unsigned old = atomicAdd(a.cnt + grp, 1u);
        last_flag = old == (unsigned)a.S - 1;
        if (last_flag) a.cnt[grp] = 0;
#endif
    }
    __syncthreads();
    if (last_flag) {
#if BX_LAST == 2
        asm volatile("buffer_inv sc1" ::: "memory");
#else
        __threadfence();
#endif
#pragma unroll
        for (int r = 0; r < NT * 2; ++r) {
            const int row = r * 4 + w, t = row >> 3, i = (row >> 2) & 1, g = row & 3;
            const long long o = ((long long)((grp * NT + t) * 4 + g) * 2 + i) * 64 + lane;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s0 = 0; s0 < a.S; s0 += 4) {
                float4 q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = a.out[(long long)min(s0 + e, a.S - 1) * a.slab_stride + o];
#pragma unroll
                for (int e = 0; e < 4; ++e) if (s0 + e < a.S) { v.x += q[e].x; v.y += q[e].y; v.z += q[e].z; v.w += q[e].w; }
            }
            a.fin[o] = v;
        }
    }
#endif
#if BX_STAMP
    if (w == 0 && a.trace && lane == 0) { for (int i = 0; i < 3 * PER; ++i) a.trace[blockIdx.x * 32 + 4 + i] = ts[i] - te; }
#endif
    if (w == 0 && a.trace && lane == 0) { a.trace[blockIdx.x * 32] = te; a.trace[blockIdx.x * 32 + 1] = t0; a.trace[blockIdx.x * 32 + 2] = t1; a.trace[blockIdx.x * 32 + 3] = __builtin_amdgcn_s_memtime(); }
}

template <int NT, int PER, int NW = 4>
static void run(const char* name, int N, int K, int S, hipStream_t st) {
    const int NL = 12;
    std::vector<float> hW((size_t)N * K), hX((size_t)64 * K);
    srand(1);
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.08f;
    for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    float *W, *X; float4* out; u32x4* Xq;
    (void)hipMalloc(&W, hW.size() * 4); (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&X, hX.size() * 4); (void)hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&Xq, (size_t)K / 16 * 6 * 64 * 16);
    const size_t slab = (size_t)N * 64 / 4;
#if BX_LAST == 2
    if (hipExtMallocWithFlags((void**)&out, slab * S * 16, hipDeviceMallocUncached) != hipSuccess) { printf("uncached alloc failed\n"); return; }
#else
    (void)hipMalloc(&out, slab * S * 16);
#endif
    hipLaunchKernelGGL(k_pack_x, dim3((K / 16 * 128 + 255) / 256), dim3(256), 0, st, X, Xq, K);
    std::vector<float4*> Wq(NL);
    for (int l = 0; l < NL; ++l) {
        (void)hipMalloc(&Wq[l], (size_t)N * K * 4);
        const long long total = (long long)(N / 32) * (K / 16) * 128;
        hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, Wq[l], N, K);
    }
    const int grid = N / 32 / NT * S;
    unsigned long long* tr; (void)hipMalloc(&tr, 1024 * 32 * 8); (void)hipMemset(tr, 0, 1024 * 32 * 8);
    unsigned* cnt;
#if BX_LAST == 2
    (void)hipExtMallocWithFlags((void**)&cnt, 4096, hipDeviceMallocUncached);
#else
    (void)hipMalloc(&cnt, 4096);
#endif
    (void)hipMemset(cnt, 0, 4096);
    float4* fin; (void)hipMalloc(&fin, slab * 16);
    BxArgs a{}; a.Xq = Xq; a.out = out; a.KU = K / 16; a.S = S; a.slab_stride = slab; a.trace = tr; a.cnt = cnt; a.fin = fin;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, st);
        for (int l = 0; l < NL; ++l) { a.Wq = Wq[l]; hipLaunchKernelGGL((k_bx6<NT, PER, NW>), dim3(grid), dim3(NW * 64), 0, st, a); }
        (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 3) printf("%s [%d waves] N=%d K=%d: %d workgroups of %d columns x %d k, ring %d: %.2f us per launch (%.1f MB -> %.2f TB/s)\n", name, NW, N, K, grid, 32 * NT, K / S, BX_RING, ms * 1000.f / NL, N * (double)K * 4 / 1e6, N * (double)K * 4 / (ms * 1e-3 / NL) / 1e12);
    }
    { std::vector<unsigned long long> h(grid * 32); (void)hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
      double av[3] = {0,0,0};
      for (int i = 0; i < grid; ++i) for (int j = 0; j < 3; ++j) av[j] += (double)(h[i*32+j+1] - h[i*32+j]) / grid;
      printf("  wave 0 ticks: prologue %.0f, main loop %.0f, epilogue %.0f\n", av[0], av[1], av[2]);
#if BX_STAMP
      if (PER <= 8) for (int j = 0; j < PER; ++j) {
          double s0 = 0, s1 = 0, s2 = 0;
          for (int i = 0; i < grid; ++i) { s0 += (double)h[i*32+4+3*j] / grid; s1 += (double)h[i*32+5+3*j] / grid; s2 += (double)h[i*32+6+3*j] / grid; }
          printf("    step %d: begins %.0f, split+W reload issued %.0f (+%.0f), MFMAs + X reload issued %.0f (+%.0f)\n", j, s0, s1, s1 - s0, s2, s2 - s1);
      }
#endif
    }
    std::vector<float> o(slab * 4 * S);
#if BX_LAST
    // visibility check: poison slabs and result, run once more, the result must be complete
    (void)hipMemsetAsync(out, 0xff, slab * S * 16, st); (void)hipMemsetAsync(fin, 0xff, slab * 16, st);
    a.Wq = Wq[0]; hipLaunchKernelGGL((k_bx6<NT, PER>), dim3(grid), dim3(256), 0, st, a); (void)hipStreamSynchronize(st);
    (void)hipMemcpy(o.data(), fin, slab * 16, hipMemcpyDeviceToHost);
    const int S_host = 1;
#else
    const int S_host = S;
    (void)hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
#endif
    double e_bx = 0, e_f32 = 0, mag = 0;
    for (int n = 0; n < N; n += 13)
        for (int m = 0; m < 64; ++m) {
            double r = 0; float f = 0.f;
            for (int k = 0; k < K; ++k) { r += (double)hX[(size_t)m * K + k] * hW[(size_t)n * K + k]; f = fmaf(hX[(size_t)m * K + k], hW[(size_t)n * K + k], f); }
            float got = 0.f;
            for (int s = 0; s < S_host; ++s) got += o[s * slab * 4 + (((size_t)(n >> 3) * 2 + (m >> 5)) * 64 + (m & 31) + 32 * ((n >> 2) & 1)) * 4 + (n & 3)];
            e_bx = fmax(e_bx, fabs(got - r)); e_f32 = fmax(e_f32, fabs(f - r)); mag = fmax(mag, fabs(r));
        }
    printf("  max |bx6 - fp64| = %.3e, max |fp32 fma chain - fp64| = %.3e, max |value| = %.3f; %s\n", e_bx, e_f32, mag, hipGetErrorString(hipGetLastError()));
    for (auto p : Wq) (void)hipFree(p);
    (void)hipFree(W); (void)hipFree(X); (void)hipFree(Xq); (void)hipFree(out); (void)hipFree(tr);
}

int main() {
    hipStream_t st; (void)hipStreamCreate(&st);
#ifdef BX_NW8
    run<1, 24, 4>("fc1 n32 full K", 6144, 1536, 1, st);
    run<1, 12, 8>("fc1 n32 full K", 6144, 1536, 1, st);
    run<1, 6, 4>("proj", 1536, 1536, 4, st);
    run<1, 3, 8>("proj", 1536, 1536, 4, st);
    run<1, 4, 4>("proj", 1536, 1536, 6, st);
    run<1, 2, 8>("proj", 1536, 1536, 6, st);
    run<1, 24, 4>("fc2 n32 k1536", 1536, 6144, 4, st);
    run<1, 12, 8>("fc2 n32 k1536", 1536, 6144, 4, st);
    run<1, 16, 4>("fc2 n32 k1024", 1536, 6144, 6, st);
    run<1, 8, 8>("fc2 n32 k1024", 1536, 6144, 6, st);
    run<2, 12, 4>("fc2 n64 k768", 1536, 6144, 8, st);
    run<2, 6, 8>("fc2 n64 k768", 1536, 6144, 8, st);
    run<2, 6, 4>("qkv n64 k384", 4608, 1536, 4, st);
    run<2, 3, 8>("qkv n64 k384", 4608, 1536, 4, st);
    run<2, 12, 4>("qkv n64 k768", 4608, 1536, 2, st);
    run<2, 6, 8>("qkv n64 k768", 4608, 1536, 2, st);
    run<1, 12, 8>("qkv n32 whole K", 4608, 1536, 1, st);
    return 0;
#endif
    run<3, 6>("fc1", 6144, 1536, 4, st);
    run<3, 6>("fc2", 1536, 6144, 16, st);
    run<3, 6>("qkv", 4608, 1536, 4, st);
    run<1, 6>("proj", 1536, 1536, 4, st);
    run<2, 12>("fc2 n64 k768", 1536, 6144, 8, st);
    run<2, 6>("fc1 n64 k384", 6144, 1536, 4, st);
    run<2, 6>("fc2 n64 k384", 1536, 6144, 16, st);
    run<2, 6>("qkv n64 k384", 4608, 1536, 4, st);
    run<1, 6>("fc1 n32 k384", 6144, 1536, 4, st);
    run<1, 24>("fc1 n32 full K", 6144, 1536, 1, st);
    return 0;
}
