// Dev tool: what a kernel boundary between two weight-streaming kernels costs on MI355X, and what a predecessor can do about it.
// A chain of launches (captured in a hipGraph), each streaming its own `bytes` of weights from HBM with G workgroups x NW waves
// (non-temporal 1-KiB loads, 16 KiB in flight per wave), optionally
//   pf = 1: an extra wave per workgroup touches, at kernel START, the first P KiB that the same-numbered workgroup of the NEXT launch
//           will stream (one dword per 128-byte line: the lines land in this XCD's L2 -- block b of both launches runs on XCD b % 8),
//   pf = 2: the same touches issued by wave 0 AFTER its own stream (kernel tail),
//   pf = 3: the extra wave touches the workgroup's share of ALL of the next launch's bytes (memory-side cache),
//   warm  : every launch streams the SAME buffer (fits the 256 MB memory-side cache).
// Prints microseconds per launch and TB/s for a sweep of shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const f32x4* W;            // this launch's weights
    const char* Wnext;         // next launch's weights (prefetch target)
    long long per_wave_f4;     // float4 x 64 lanes units per wave = 1-KiB loads per wave
    long long per_wg_bytes;
    float* sink;
    int pf, pf_kib;
};

__device__ __forceinline__ void touch(const char* p, long long bytes, int lane) {
    // one dword per 128-byte line, 64 lines (8 KiB) per wave instruction; results are discarded
    unsigned acc = 0;
    for (long long o = 0; o < bytes; o += 8192 * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long off = o + u * 8192 + lane * 128;
            if (off < bytes) acc += *(const volatile unsigned*)(p + off);
        }
    }
    if (acc == 0x12345678u) *(volatile unsigned*)p = acc;
}

template <int NW>
__global__ __launch_bounds__((NW + 1) * 64) void k_stream(Args a) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w == NW) {       // the prefetch wave
        if (a.pf == 1) touch(a.Wnext + (long long)blockIdx.x * a.per_wg_bytes, (long long)a.pf_kib * 1024, lane);
        if (a.pf == 3) touch(a.Wnext + (long long)blockIdx.x * a.per_wg_bytes, a.per_wg_bytes, lane);
        return;
    }
    const f32x4* p = a.W + ((long long)blockIdx.x * NW + w) * a.per_wave_f4 * 64 + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 rA[8], rB[8];
    const long long n16 = a.per_wave_f4 / 16 * 16, n = a.per_wave_f4;     // n % 4 == 0
    if (n16 > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rA[i] = __builtin_nontemporal_load(p + (long long)i * 64);
        for (long long k = 0; k < n16; k += 16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rB[i] = __builtin_nontemporal_load(p + (k + 8 + i) * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += rA[i];
            __builtin_amdgcn_sched_barrier(0);
            const long long k2 = k + 16 < n ? k + 16 : 0;       // the tail's first loads (or a harmless re-read)
#pragma unroll
            for (int i = 0; i < 8; ++i) rA[i] = __builtin_nontemporal_load(p + (k2 + (i < n - n16 || n == n16 ? i : 0)) * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += rB[i];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += rA[i];
        for (long long k = n16 + 8; k < n; k += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += __builtin_nontemporal_load(p + (k + i) * 64);
        }
    } else {
        for (long long k = 0; k < n; k += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rA[i] = __builtin_nontemporal_load(p + (k + i) * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += rA[i];
        }
    }
    if (a.pf == 2 && w == 0) touch(a.Wnext + (long long)blockIdx.x * a.per_wg_bytes, (long long)a.pf_kib * 1024, lane);
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) a.sink[0] = acc.x;
}

static double run(int G, int NW, long long per_wave_kib, int pf, int pf_kib, bool warm, int NL, std::vector<f32x4*>& bufs, float* sink) {
    const long long per_wg_bytes = per_wave_kib * 1024 * NW;
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipGraph_t graph; hipGraphExec_t exec;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int l = 0; l < NL; ++l) {
        Args a{};
        a.W = warm ? bufs[0] : bufs[l % bufs.size()];
        a.Wnext = (const char*)(warm ? bufs[0] : bufs[(l + 1) % bufs.size()]);
        a.per_wave_f4 = per_wave_kib; a.per_wg_bytes = per_wg_bytes; a.sink = sink; a.pf = pf; a.pf_kib = pf_kib;
        if (NW == 4) hipLaunchKernelGGL(k_stream<4>, dim3(G), dim3(320), 0, st, a);
        else hipLaunchKernelGGL(k_stream<8>, dim3(G), dim3(576), 0, st, a);
    }
    (void)hipStreamEndCapture(st, &graph);
    (void)hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, st);
        (void)hipGraphLaunch(exec, st);
        (void)hipEventRecord(e1, st);
        (void)hipStreamSynchronize(st);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph); (void)hipStreamDestroy(st);
    return best * 1e3 / NL;
}

int main() {
    const int NL = 96;
    const size_t cap = (size_t)48 << 20;         // bytes per buffer: largest launch (37.7 MB + slack)
    std::vector<f32x4*> bufs(24);                // 24 x 48 MB = 1.15 GB >> the 256 MB memory-side cache
    for (auto& p : bufs) { (void)hipMalloc(&p, cap); (void)hipMemset(p, 0x11, cap); }
    float* sink; (void)hipMalloc(&sink, 4);
    (void)hipDeviceSynchronize();
    struct Shape { const char* name; int G, NW; long long kib; };
    const Shape shapes[] = {
        {"fc1/fc2 37.7MB 256x4", 256, 4, 36}, {"fc1 37.7MB 192x4", 192, 4, 48}, {"fc1 37.7MB 192x8", 192, 8, 24},
        {"qkv 28.3MB 252x4", 252, 4, 28}, {"qkv 28.3MB 144x8", 144, 8, 24},
        {"proj 9.4MB 192x4", 192, 4, 12}, {"proj 9.4MB 96x8", 96, 8, 12}, {"proj 9.4MB 48x8", 48, 8, 24}, {"proj 9.4MB 256x4", 256, 4, 9},
        {"tiny 1MB 256x4", 256, 4, 1},
    };
    printf("%-24s %8s %8s | %8s %8s %8s %8s %8s %8s | %8s\n", "shape", "us", "TB/s", "pf1:16K", "pf1:64K", "pf2:16K", "pf2:64K", "pf3:all", "pf3 TB/s", "warm us");
    for (const Shape& s : shapes) {
        const double mb = (double)s.G * s.NW * s.kib * 1024 / 1e6;
        const long long kib = s.kib;       // 1-KiB loads per wave, a multiple of 4
        const double mbr = (double)s.G * s.NW * kib * 1024 / 1e6;
        const double base = run(s.G, s.NW, kib, 0, 0, false, NL, bufs, sink);
        const double p1a = run(s.G, s.NW, kib, 1, 16, false, NL, bufs, sink);
        const double p1b = run(s.G, s.NW, kib, 1, 64, false, NL, bufs, sink);
        const double p2a = run(s.G, s.NW, kib, 2, 16, false, NL, bufs, sink);
        const double p2b = run(s.G, s.NW, kib, 2, 64, false, NL, bufs, sink);
        const double p3 = run(s.G, s.NW, kib, 3, 0, false, NL, bufs, sink);
        const double wm = run(s.G, s.NW, kib, 0, 0, true, NL, bufs, sink);
        printf("%-24s %8.2f %8.2f | %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f | %8.2f   (%.1f MB asked, %.1f MB streamed)\n", s.name, base, mbr / base * 1e-6,
               p1a, p1b, p2a, p2b, p3, mbr / p3 * 1e-6, wm, mb, mbr);
        fflush(stdout);
    }
    return 0;
}
