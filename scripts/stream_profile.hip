// Dev tool: arrival profile of a weight stream.  256 workgroups x 4 waves; every wave issues its 36 one-KiB loads (6 steps x 6) up
// front, then stamps s_memtime as each step's six loads land (s_waitcnt vmcnt counts in issue order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 0xf) | (((N) >> 4) << 14) | (0x7 << 4) | (0xf << 8))
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ W, float* sink, unsigned long long* tr) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const f32x4* p = W + ((long long)blockIdx.x * 4 + w) * 36 * 64 + lane;
    f32x4 r[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) r[i] = __builtin_nontemporal_load(p + i * 64);
    unsigned long long t[7];
    t[0] = __builtin_amdgcn_s_memtime();
    VMCNT(30); t[1] = __builtin_amdgcn_s_memtime();
    VMCNT(24); t[2] = __builtin_amdgcn_s_memtime();
    VMCNT(18); t[3] = __builtin_amdgcn_s_memtime();
    VMCNT(12); t[4] = __builtin_amdgcn_s_memtime();
    VMCNT(6); t[5] = __builtin_amdgcn_s_memtime();
    VMCNT(0); t[6] = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 36; ++i) acc += r[i].x + r[i].y + r[i].z + r[i].w;
    if (acc == 123.456f) sink[0] = acc;
    if (lane == 0) for (int i = 0; i < 7; ++i) tr[((long long)blockIdx.x * 4 + w) * 7 + i] = t[i] - t0;
}
int main() {
    const int G = 256, NL = 12;
    const size_t bytes = (size_t)G * 4 * 36 * 1024;
    std::vector<f32x4*> W(NL);
    for (auto& p : W) { (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); }
    float* sink; (void)hipMalloc(&sink, 4);
    unsigned long long* tr; (void)hipMalloc(&tr, G * 4 * 7 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(k_stream, dim3(G), dim3(256), 0, 0, W[l], sink, tr);
        (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%.1f MB per launch, %.2f us -> %.2f TB/s\n", bytes / 1e6, ms * 1e3 / NL, bytes / (ms * 1e-3 / NL) / 1e12);
    }
    std::vector<unsigned long long> h(G * 4 * 7); (void)hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
    const char* nm[7] = {"issued", "step 0", "step 1", "step 2", "step 3", "step 4", "step 5"};
    for (int i = 0; i < 7; ++i) {
        double av = 0, mn = 1e18, mx = 0;
        for (int j = 0; j < G * 4; ++j) { const double v = (double)h[j * 7 + i]; av += v / (G * 4); mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        printf("  %s landed: avg %.0f ticks (min %.0f, max %.0f)\n", nm[i], av, mn, mx);
    }
    return 0;
}
