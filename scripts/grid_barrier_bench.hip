// Dev microbenchmark: cost of a device-wide barrier between phases of ONE persistent kernel on gfx950
// (256 workgroups, one per CU), against the cost of a kernel boundary.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned nwg, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += nwg;
        __threadfence();                                   // release: this workgroup's writes are visible device-wide
        atomicAdd(ctr, 1u);
        while (__atomic_load_n(ctr, __ATOMIC_RELAXED) < epoch) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                   // acquire
    }
    __syncthreads();
}

// each phase: every workgroup writes a value that ALL workgroups read in the next phase (checks visibility)
__global__ __launch_bounds__(256) void k_persistent(unsigned* ctr, float* buf, int phases, int nwg, float* out) {
    unsigned epoch = 0;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        if (threadIdx.x == 0) buf[(p & 1) * nwg + blockIdx.x] = (float)(p + blockIdx.x);
        grid_barrier(ctr, nwg, epoch);
        // read a few entries written by other workgroups in this phase
        float s = 0.f;
        for (int i = threadIdx.x; i < nwg; i += 256) s += __builtin_nontemporal_load(buf + (p & 1) * nwg + i);
        acc += s;
    }
    // reduce acc over the block
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0; for (int i = 0; i < 256; ++i) t += red[i]; out[blockIdx.x] = t; }
}

__global__ __launch_bounds__(256) void k_phase(float* buf, int p, int nwg, float* out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nwg; i += 256) s += buf[((p + 1) & 1) * nwg + i];
    if (threadIdx.x == 0) { buf[(p & 1) * nwg + blockIdx.x] = (float)(p + blockIdx.x); out[blockIdx.x] += s; }
}

int main() {
    const int nwg = 256, phases = 2000;
    unsigned* ctr; float *buf, *out;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, 2 * nwg * 4)); CK(hipMalloc(&out, nwg * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 4)); CK(hipMemset(buf, 0, 2 * nwg * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_persistent, dim3(nwg), dim3(256), 0, 0, ctr, buf, phases, nwg, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> h(nwg); CK(hipMemcpy(h.data(), out, nwg * 4, hipMemcpyDeviceToHost));
        // expected per workgroup: sum_p sum_i (p + i) = phases*nwg*(nwg-1)/2 + nwg*phases*(phases-1)/2
        double expect = (double)phases * nwg * (nwg - 1) / 2 + (double)nwg * phases * (phases - 1) / 2;
        printf("persistent: %d phases in %.3f ms = %.3f us per barrier+phase   (check wg0 %.0f vs %.0f)\n", phases, ms, ms * 1e3 / phases, h[0], expect);
    }
    // kernel-boundary version inside a graph
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < 200; ++p) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(256), 0, st, buf, p, nwg, out);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph of kernels: 2000 phases in %.3f ms = %.3f us per kernel\n", ms, ms * 1e3 / 2000);
    }
    return 0;
}
