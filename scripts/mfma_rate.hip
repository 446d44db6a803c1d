// Dev tool: s_memtime ticks per v_mfma_f32_32x32x2_f32 in a register-only loop (1 wave per SIMD, 2 accumulators), and the
// wall-clock rate of the same loop: tells whether s_memtime counts shader cycles and what the MFMA pipe sustains under load.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 a0, a1;
    for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* tk; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&tk, 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1, 256}) for (int iters : {200, 2000}) {
        k<<<grid, 256>>>(out, tk, iters); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<grid, 256>>>(out, tk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, tk, grid * 8, hipMemcpyDeviceToHost);
        double n = 32.0 * iters;
        printf("grid %3d iters %4d: %.1f ticks/MFMA (wg0), wall %.2f us -> %.1f ns/MFMA, ticks/us %.0f, TF/s %.1f\n", grid, iters, h[0] / n,
               ms * 1e3, ms * 1e6 / n, h[0] / (ms * 1e3), grid * 4 * n * 2.0 * 32 * 32 * 2 / (ms * 1e-3) * 1e-12);
    }
    return 0;
}
