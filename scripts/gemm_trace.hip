// Dev tool: per-workgroup timeline of k_gemm (s_memtime stamps): start, first operands landed,
// main loop done, end.   hipcc -DWMAR_GEMM_TRACE ...
#define WMAR_GEMM_TRACE 1
#include "../wmar_amd/csrc/gpt.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace wmar;

int main() {
    const int B = 64, MT = 2, D = 1536;
    hipStream_t st; hipStreamCreate(&st);
    const int N = 4 * D, K = D;
    float4 *W, *X, *out; double* stats; float* bias; unsigned long long* tr;
    hipMalloc(&W, (size_t)N * K * 4); hipMemset(W, 0x3c, (size_t)N * K * 4);
    hipMalloc(&X, (size_t)64 * 4 * D * 4); hipMemset(X, 0x3c, (size_t)64 * 4 * D * 4);
    hipMalloc(&out, (size_t)8 * 64 * 4 * D * 4);
    hipMalloc(&stats, 64 * 64 * 2 * 8); hipMemset(stats, 0, 64 * 64 * 2 * 8);
    hipMalloc(&bias, 16384 * 4); hipMemset(bias, 0, 16384 * 4);
    const int NWG = 4096;
    hipMalloc(&tr, NWG * 4 * 8);
    GemmArgs a{};
    a.Wp = W; a.Xp = X; a.bias = bias; a.c1 = bias; a.KB = K / 8; a.NT = N / 32; a.MT = MT; a.S = 1;
    a.stats = stats; a.n_chunks = 12; a.K = K; a.out_packed = out; a.slab_stride = (long long)N / 8 * MT * 64; a.B = B;
    a.trace = tr;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(tr, 0, NWG * 4 * 8);
        launch_gemm<2, 4, EPI_GELU, true>(a, st);
        launch_gemm<2, 4, EPI_GELU, true>(a, st);
        hipStreamSynchronize(st);
    }
    const int grid = a.NT * (a.MT / 2);
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), tr, grid * 32, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, tend = 0;
    for (int i = 0; i < grid; ++i) { t0 = std::min(t0, h[i * 4]); tend = std::max(tend, h[i * 4 + 3]); }
    printf("grid %d, kernel span %llu ticks\n", grid, tend - t0);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0; unsigned long long maxstart = 0;
    for (int i = 0; i < grid; ++i) {
        s0 += h[i * 4] - t0; s1 += h[i * 4 + 1] - h[i * 4]; s2 += h[i * 4 + 2] - h[i * 4 + 1]; s3 += h[i * 4 + 3] - h[i * 4 + 2];
        maxstart = std::max(maxstart, h[i * 4] - t0);
    }
    printf("avg start offset %.0f (max %llu), first-operands %.0f, main loop %.0f, epilogue %.0f ticks\n", s0 / grid, maxstart,
           s1 / grid, s2 / grid, s3 / grid);
    for (int i = 0; i < grid; i += grid / 8)
        printf("wg %4d: start %6llu  +load %6llu  +loop %6llu  +epi %6llu\n", i, h[i * 4] - t0, h[i * 4 + 1] - h[i * 4],
               h[i * 4 + 2] - h[i * 4 + 1], h[i * 4 + 3] - h[i * 4 + 2]);
    return 0;
}
