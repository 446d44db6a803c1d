// Dev tool: how fast can ONE compute unit pull L2-resident data (the activation pieces every workgroup of a skinny GEMM re-reads)?
// 256 workgroups; NL loader waves each stream their share of the SAME `bytes` buffer (start rotated per workgroup) REP times,
// as LDS-DMA into a ring (MODE 0) or as plain 16-byte loads into registers (MODE 1), with D one-KiB requests in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")

template <int MODE, int D>
__global__ __launch_bounds__(512) void k_pull(const f32x4* __restrict__ X, int n_kib, int rep, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    // wave w takes KiB pieces w, w + nw, ...; the walk starts at a workgroup-dependent piece
    const int per = n_kib / nw;
    const int rot = (blockIdx.x * 37) % per;
    char* ring = smem + w * D * 1024;
    f32x4 acc = {0, 0, 0, 0};
    f32x4 r[D];
    for (int it = 0; it < rep; ++it) {
        for (int p0 = 0; p0 < per; p0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                int p = p0 + d + rot; if (p >= per) p -= per;
                const f32x4* src = X + ((long long)(p * nw + w) * 64 + lane);
                if (MODE == 0) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + d * 1024), 16, 0, 0);
                else r[d] = *src;
            }
            if (MODE == 0) { VMCNT(D / 2); }      // half of the ring stays in flight
            else {
#pragma unroll
                for (int d = 0; d < D; ++d) acc += r[d];
            }
        }
    }
    if (MODE == 0) VMCNT(0);
    if (acc.x == 123.f) sink[0] = acc.y;
}

template <int MODE, int D>
static void run(int nw, int n_kib, const f32x4* X, float* sink) {
    const int rep = 16;
    const size_t lds = MODE == 0 ? (size_t)nw * D * 1024 : 0;
    (void)hipFuncSetAttribute((const void*)k_pull<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int i = 0; i < 4; ++i) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_pull<MODE, D>), dim3(256), dim3(nw * 64), lds, 0, X, n_kib, rep, sink);
        (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n_kib * 1024 * rep;
    printf("%s  %d waves x %2d KiB in flight, %4d KiB buffer: %7.1f us -> %6.1f GB/s per CU (%5.1f TB/s chip)  %s\n", MODE == 0 ? "lds-dma" : "vgpr   ", nw, D, n_kib,
           best * 1e3, bytes / (best * 1e-3) / 1e9, bytes * 256 / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
    f32x4* X; (void)hipMalloc(&X, 4 << 20); (void)hipMemset(X, 0, 4 << 20);
    float* sink; (void)hipMalloc(&sink, 4);
    for (int kib : {576, 96}) {
        run<0, 8>(1, kib, X, sink); run<0, 16>(1, kib, X, sink); run<0, 32>(1, kib, X, sink); run<0, 48>(1, kib, X, sink);
        run<0, 8>(2, kib, X, sink); run<0, 16>(2, kib, X, sink); run<0, 32>(2, kib, X, sink);
        run<0, 8>(4, kib, X, sink); run<0, 16>(4, kib, X, sink); run<0, 32>(4, kib, X, sink);
        run<1, 8>(1, kib, X, sink); run<1, 16>(1, kib, X, sink);
        run<1, 8>(2, kib, X, sink); run<1, 16>(2, kib, X, sink);
        run<1, 8>(4, kib, X, sink); run<1, 16>(4, kib, X, sink);
        run<1, 8>(8, kib, X, sink); run<1, 16>(8, kib, X, sink);
    }
    return 0;
}
