// Dev microbenchmark (round 5): a HIERARCHICAL device-wide barrier inside one persistent kernel on gfx950 -- arrival and release inside
// each XCD through its own L2 (the atomics of k_bx_xr: no sc1, they never leave the XCD), one device-scope atomic per XCD between them --
// against the naive barrier of scripts/grid_barrier_bench.hip (17 us: __threadfence + 256 workgroups polling one word) and a kernel
// boundary inside a hipGraph (~2 us).  Data crosses the barrier the way the decoder kernels pass it: write-through (sc1) stores, agent-scope
// (sc1) loads.  256 workgroups (one per CU), workgroups with equal blockIdx % 8 share an XCD (checked with HW_REG_XCC_ID).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/grid_barrier2_bench.bin scripts/grid_barrier2_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned l2_read(unsigned* p) {
    unsigned v; const unsigned z = 0;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(z) : "memory");
    return v;
}
__device__ __forceinline__ unsigned l2_add(unsigned* p, unsigned v) {
    unsigned o;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(o) : "v"(p), "v"(v) : "memory");
    return o;
}
__device__ __forceinline__ void l2_add_noret(unsigned* p, unsigned v) { asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void l2_swap_noret(unsigned* p, unsigned v) { asm volatile("global_atomic_swap %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
// device scope: performed at the memory side (sc1), visible to every XCD
__device__ __forceinline__ unsigned dev_add(unsigned* p, unsigned v) {
    unsigned o;
    asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(o) : "v"(p), "v"(v) : "memory");
    return o;
}
__device__ __forceinline__ void dev_add_noret(unsigned* p, unsigned v) { asm volatile("global_atomic_add %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned dev_read(unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ float ld_dev(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 15u; }

// xs: [8][64] words per XCD group (word 0 arrivals, word 32 generation); gs: [64] (word 0 XCD arrivals, word 32 generation)
// MODE 0: hierarchical; MODE 1: XCD-local only (no device phase: NOT a device-wide barrier, the lower bound)
template <int MODE>
__device__ __forceinline__ void grid_barrier2(unsigned* xs, unsigned* gs, unsigned members, unsigned phase, unsigned long long* stamp) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* cnt = xs + (blockIdx.x & 7) * 64;
        unsigned* gen = cnt + 32;
        const unsigned old = l2_add(cnt, 1u);
        if (old == members - 1u) {
            l2_swap_noret(cnt, 0u);
            if (MODE == 0) {
                const unsigned g = dev_add(gs, 1u);
                if (g == 8u * phase + 7u) dev_add_noret(gs + 32, 1u);
                else { int n = 0; while (dev_read(gs + 32) == phase && ++n < (1 << 18)) __builtin_amdgcn_s_sleep(1); }      // bounded: a hang would cost the box
            }
            l2_add_noret(gen, 1u);
        } else {
            { int n = 0; while (l2_read(gen) == phase && ++n < (1 << 18)) __builtin_amdgcn_s_sleep(1); }
        }
        if (stamp) *stamp = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void k_persistent(unsigned* xs, unsigned* gs, float* buf, int phases, int nwg, float* out, unsigned* xcc_seen) {
    float acc = 0.f;
    if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc_id();
    for (int p = 0; p < phases; ++p) {
        if (threadIdx.x == 0) st_wt(buf + (p & 1) * nwg + blockIdx.x, (float)(p + blockIdx.x));
        grid_barrier2<MODE>(xs, gs, (unsigned)(nwg / 8), (unsigned)p, nullptr);
        float s = 0.f;
        if (MODE == 0) { for (int i = threadIdx.x; i < nwg; i += 256) s += ld_dev(buf + (p & 1) * nwg + i); }
        else { for (int i = (blockIdx.x & 7) + 8 * threadIdx.x; i < nwg; i += 8 * 256) s += ld_dev(buf + (p & 1) * nwg + i); }   // own XCD's entries only
        acc += s;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0; for (int i = 0; i < 256; ++i) t += red[i]; out[blockIdx.x] = t; }
}

int main() {
    const int nwg = 256, phases = 2000;
    unsigned *xs, *gs, *xcc; float *buf, *out;
    CK(hipMalloc(&xs, 8 * 64 * 4)); CK(hipMalloc(&gs, 64 * 4)); CK(hipMalloc(&xcc, nwg * 4));
    CK(hipMalloc(&buf, 2 * nwg * 4)); CK(hipMalloc(&out, nwg * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(xs, 0, 8 * 64 * 4)); CK(hipMemset(gs, 0, 64 * 4)); CK(hipMemset(buf, 0, 2 * nwg * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(nwg), dim3(256), 0, 0, xs, gs, buf, phases, nwg, out, xcc);
            else hipLaunchKernelGGL(k_persistent<1>, dim3(nwg), dim3(256), 0, 0, xs, gs, buf, phases, nwg, out, xcc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<float> h(nwg); CK(hipMemcpy(h.data(), out, nwg * 4, hipMemcpyDeviceToHost));
            std::vector<unsigned> xc(nwg); CK(hipMemcpy(xc.data(), xcc, nwg * 4, hipMemcpyDeviceToHost));
            int grouped = 1;
            for (int i = 8; i < nwg; ++i) if (xc[i] != xc[i & 7]) grouped = 0;
            // expected per workgroup (mode 0): sum_p sum_i (p + i) = phases*nwg*(nwg-1)/2 + nwg*phases*(phases-1)/2
            const double expect = (double)phases * nwg * (nwg - 1) / 2 + (double)nwg * phases * (phases - 1) / 2;
            printf("%s: %d phases in %.3f ms = %.3f us per (store + barrier + %s read)   blocks grouped by blockIdx %% 8: %s   check wg0 %.0f%s\n",
                   mode == 0 ? "hierarchical device-wide barrier" : "XCD-local barrier only (lower bound)", phases, ms, ms * 1e3 / phases,
                   mode == 0 ? "256-entry" : "32-entry", grouped ? "yes" : "NO", h[0], mode == 0 ? (fabs(h[0] - expect) <= 1e-3 * expect ? " = expected" : " != expected") : "");
        }
    return 0;
}
