// Minimal reproducer attempt for the round-3 finding (csrc/common.h prod_f64): a workgroup of eight waves, four of
// them streaming weights into v_mfma_f32_32x32x16_bf16, four of them ("stagers") folding fp32 rows and accumulating fp64 row sums --
// sum (v_add_f64 chain) and sum of squares (v_mul_f64 + dependent v_fmac_f64 chain, the compiler's contraction of x*x + y*y + z*z + w*w).
// Every launch runs on the same inputs; the host compares each launch's (sum, sumsq) with the first launch's, bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -DPAD=<n> [-DNOFMA] scripts/fmac64_repro.hip -o scripts/fmac64_repro.bin ; ./fmac64_repro.bin [launches]
// PAD: s_nop instructions in front of the kernel body (moves the code address);  NOFMA: products kept out of the fused chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef PAD
#define PAD 0
#endif
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
constexpr int KB = 28, NSLAB = 6, CK = 4;       // k-blocks per workgroup slice, slabs folded, k-blocks per chunk (one per stager wave)

#define STR2(x) #x
#define STR(x) STR2(x)
__device__ __forceinline__ double sq4(const float4 r) {
#ifdef NOFMA
#pragma clang fp contract(off)
    double a = (double)r.x * (double)r.x, b = (double)r.y * (double)r.y, c = (double)r.z * (double)r.z, d = (double)r.w * (double)r.w;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    return ((a + b) + c) + d;
#else
    return (double)r.x * r.x + (double)r.y * r.y + (double)r.z * r.z + (double)r.w * r.w;
#endif
}

__global__ __launch_bounds__(512) void k_repro(const float4* __restrict__ x, const float4* __restrict__ slabs, long long slab_stride,
                                               const float4* __restrict__ W, double2* __restrict__ out, float* __restrict__ sink, int mode, unsigned int* __restrict__ bad) {
    if (PAD > 0) asm volatile(".rept " STR(PAD) "\n s_nop 0\n .endr" ::: "memory");
    __shared__ __attribute__((aligned(16))) u32x4 xq[2][CK][2][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kb0 = (blockIdx.x % 7) * KB;
    constexpr int NCH = KB / CK;
    if (w >= 4) {
        const int sw = w - 4;
        double sum[2] = {0, 0}, sq[2] = {0, 0};
        for (int c = 0; c < NCH; ++c) {
            const int kb = kb0 + c * CK + sw;
            float4 r[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long long idx = ((long long)kb * 2 + i) * 64 + lane;
                float4 v = x[idx], t = slabs[idx];
#pragma unroll
                for (int s = 1; s < NSLAB; ++s) { const float4 q = slabs[s * slab_stride + idx]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
                r[i] = make_float4(v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xq[c & 1][sw][i][lane] = u32x4{__float_as_uint(r[i].x), __float_as_uint(r[i].y), __float_as_uint(r[i].z), __float_as_uint(r[i].w)};
                sum[i] += (double)r[i].x + (double)r[i].y + (double)r[i].z + (double)r[i].w;
                sq[i] += sq4(r[i]);
            }
            __syncthreads();
        }
        double2* o = out + ((long long)blockIdx.x * 4 + sw) * 128;
        if (mode == 0) {                 // first launch: the reference values
            o[lane] = make_double2(sum[0], sq[0]);
            o[64 + lane] = make_double2(sum[1], sq[1]);
        } else {                         // every later launch compares in place (no extra launch, no host copy: both hid the original effect)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double2 want = o[64 * i + lane];
                if (want.x != sum[i] || want.y != sq[i]) {
                    const unsigned int n = atomicAdd(bad, 1u);
                    if (n < 16) { bad[4 + 4 * n] = blockIdx.x; bad[5 + 4 * n] = (sw << 8) | (i << 7) | lane; bad[6 + 4 * n] = want.x != sum[i]; bad[7 + 4 * n] = want.y != sq[i]; }
                }
            }
        }
        return;
    }
    // multiplying waves: stream fp32 "weights", convert, 12 bf16 MFMAs per chunk on operands read from LDS
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float4* wp = W + ((long long)blockIdx.x * 4 + w) * NCH * 4 * 64 + lane;
    for (int c = 0; c < NCH; ++c) {
        float4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 t = __builtin_nontemporal_load((const f4*)(wp + ((long long)c * 4 + u) * 64));
            wv[u] = make_float4(t.x, t.y, t.z, t.w);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32x4 xa = xq[c & 1][u][0][lane], xb = xq[c & 1][u][1][lane];
            const u32x4 wa = {__float_as_uint(wv[u].x), __float_as_uint(wv[u].y), __float_as_uint(wv[u].z), __float_as_uint(wv[u].w)};
            const bf16x8 A = __builtin_bit_cast(bf16x8, wa);
#pragma unroll
            for (int rep = 0; rep < 3; ++rep) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(bf16x8, xa), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(bf16x8, xb), acc[1], 0, 0, 0);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    if (s == 1234.5f) sink[threadIdx.x] = s;
}

// 80 KB of straight-line code on every CU: evicts the instruction cache between two launches of k_repro (-DTHRASH), the way the other
// kernels of a decode step do (the original effect hit the first launch of the kernel in a step five times out of seven)
__global__ __launch_bounds__(64) void k_thrash(float* sink) {
    float v = threadIdx.x;
    asm volatile(".rept 20000\n v_add_f32 %0, %0, %0\n .endr" : "+v"(v));
    if (v == 1234.5f) sink[0] = v;
}

#define CK_(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20000;
    const int G = 252;                                   // 36 column groups x 7 slices, like the real launch
    const long long act = 7LL * KB * 2 * 64;             // float4 per activation buffer
    std::vector<float> hx(act * 4), hs(act * 4 * NSLAB);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f * 2.f - 1.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hs) v = 0.3f * rnd();
    float4 *x, *slabs, *W; double2* out; float* sink;
    const long long wN = (long long)G * 4 * (KB / CK) * 4 * 64;
    CK_(hipMalloc(&x, act * 16)); CK_(hipMalloc(&slabs, act * 16 * NSLAB)); CK_(hipMalloc(&W, wN * 16 * 8)); CK_(hipMalloc(&out, (size_t)G * 4 * 128 * 16));
    CK_(hipMalloc(&sink, 4096));
    CK_(hipMemcpy(x, hx.data(), act * 16, hipMemcpyHostToDevice)); CK_(hipMemcpy(slabs, hs.data(), act * 16 * NSLAB, hipMemcpyHostToDevice));
    CK_(hipMemset(W, 0x3c, wN * 16 * 8));
    unsigned int* bad;
    CK_(hipMalloc(&bad, 4096)); CK_(hipMemset(bad, 0, 4096));
    for (int it = 0; it < launches; ++it)       // a different weight buffer each launch (8 of them): the stream comes from HBM like the real per-layer weights
    {
#ifdef THRASH
        hipLaunchKernelGGL(k_thrash, dim3(512), dim3(64), 0, 0, sink);
#endif
        hipLaunchKernelGGL(k_repro, dim3(G), dim3(512), 0, 0, x, slabs, act, W + (it & 7) * wN, out, sink, it == 0 ? 0 : 1, bad);
    }
    CK_(hipDeviceSynchronize());
    unsigned int hb[128];
    CK_(hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost));
    for (unsigned i = 0; i < hb[0] && i < 16; ++i)
        printf("  workgroup %u stager %u tile %u lane %u: sum %s, sumsq %s\n", hb[4 + 4 * i], hb[5 + 4 * i] >> 8, (hb[5 + 4 * i] >> 7) & 1, hb[5 + 4 * i] & 127,
               hb[6 + 4 * i] ? "DIFFERS" : "equal", hb[7 + 4 * i] ? "DIFFERS" : "equal");
    printf("PAD %d%s: %d launches, all compared in-kernel: %u differing (lane, tile) results\n", PAD,
#ifdef NOFMA
           " NOFMA",
#else
           "",
#endif
           launches, hb[0]);
    return 0;
}
