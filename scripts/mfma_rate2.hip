// Dev tool: issue rate of the multi-block fp32 MFMAs (16x16x1 4-block, 4x4x1 16-block) alone and interleaved, in s_memtime ticks.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 a16; f32x4 a4a, a4b;
    for (int i = 0; i < 16; ++i) a16[i] = 0.f;
    for (int i = 0; i < 4; ++i) { a4a[i] = 0.f; a4b[i] = 0.f; }
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0 || MODE == 2) a16 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a16, 2, 1, 0);
            if (MODE == 1 || MODE == 2) {
                a4a = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a4a, 4, 3, 0);
                a4b = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, a4b, 4, 9, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += a16[i]; for (int i = 0; i < 4; ++i) s += a4a[i] + a4b[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* tk; (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&tk, 1024 * 8);
    const char* names[3] = {"16x16x1_4b alone", "2 x 4x4x1_16b alone", "16x16x1_4b + 2 x 4x4x1_16b per k (24 columns x 64 rows)"};
    for (int mode = 0; mode < 3; ++mode) {
        const int iters = 1000;
        if (mode == 0) k<0><<<256, 256>>>(out, tk, iters); else if (mode == 1) k<1><<<256, 256>>>(out, tk, iters); else k<2><<<256, 256>>>(out, tk, iters);
        (void)hipDeviceSynchronize();
        unsigned long long h; (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
        printf("%-60s %.1f ticks per k-step\n", names[mode], (double)h / (16.0 * iters));
    }
    return 0;
}
