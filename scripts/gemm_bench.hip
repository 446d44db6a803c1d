// Dev tool (not part of the product): times k_gemm variants on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/gemm_bench.hip wmar_amd/csrc/keytable.cpp wmar_amd/csrc/watermark.hip -o /tmp/gemm_bench
#include "../wmar_amd/csrc/gpt.hip"

#include <chrono>
#include <cstdio>
#include <vector>

using namespace wmar;

template <int MTW, int NW, int EPI, bool LN, int ABL = 0, int U = 8, bool ROT = true>
float time_gemm(GemmArgs a, int iters, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm<MTW, NW, EPI, LN, ABL, U, ROT>(a, st);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch_gemm<MTW, NW, EPI, LN, ABL, U, ROT>(a, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / iters;
}

int main() {
    const int B = 64, MT = 2, D = 1536;
    hipStream_t st; hipStreamCreate(&st);
    struct Shape { const char* name; int N, K; int S; } shapes[] = {
        {"qkv", 3 * D, D, 1}, {"fc1", 4 * D, D, 1}, {"fc2", D, 4 * D, 4}, {"proj", D, D, 5}};
    // many distinct weight buffers so that weights stream from HBM (not the 256 MB infinity cache)
    const int NBUF = 12;
    size_t wmax = (size_t)16384 * D;
    std::vector<float4*> W(NBUF);
    for (auto& p : W) { hipMalloc(&p, wmax * 4); hipMemset(p, 0x3c, wmax * 4); }
    float4 *X, *out; double* stats; float *bias, *logits, *qbuf, *kc, *vc; int* pos;
    hipMalloc(&X, (size_t)64 * 4 * D * 4); hipMemset(X, 0x3c, (size_t)64 * 4 * D * 4);
    hipMalloc(&out, (size_t)8 * 64 * 4 * D * 4);
    hipMalloc(&stats, 64 * 64 * 2 * 8); hipMemset(stats, 0, 64 * 64 * 2 * 8);
    hipMalloc(&bias, 16384 * 4); hipMemset(bias, 0, 16384 * 4);
    hipMalloc(&logits, (size_t)64 * 16384 * 4);
    hipMalloc(&qbuf, 64 * D * 4);
    hipMalloc(&kc, (size_t)64 * D * 256 * 4); hipMalloc(&vc, (size_t)64 * D * 256 * 4);
    hipMalloc(&pos, 16); hipMemset(pos, 0, 16);
    for (auto& s : shapes) {
        GemmArgs a{};
        a.Xp = X; a.bias = bias; a.c1 = bias; a.KB = s.K / 8; a.NT = s.N / 32; a.MT = MT; a.S = s.S;
        a.stats = stats; a.n_chunks = 8; a.K = s.K; a.out_packed = out; a.slab_stride = (long long)s.N / 8 * MT * 64;
        a.qbuf = qbuf; a.kcache = kc; a.vcache = vc; a.pos_dev = pos; a.D = D; a.H = 24; a.hd = 64; a.Tmax = 256;
        a.logits = logits; a.V = 16384; a.B = B;
        double flops = 2.0 * B * s.N * s.K;
        double bytes = 4.0 * s.N * s.K;
        auto report = [&](const char* var, float us) {
            printf("%-5s %-22s %8.2f us  %6.1f TF/s  %5.2f TB/s(W)\n", s.name, var, us, flops / us * 1e-6, bytes / us * 1e-6);
        };
        // rotate weight buffers per launch to defeat caches: emulate with a loop of launches over NBUF
        auto run = [&](auto fn, const char* var) {
            float tot = 0; int n = 0;
            for (int rep = 0; rep < 3; ++rep)
                for (int b = 0; b < NBUF; ++b) { a.Wp = W[b]; float us = fn(a); if (rep) { tot += us; ++n; } }
            report(var, tot / n);
        };
        a.n_chunks = 12;
        run([&](GemmArgs x) { return time_gemm<2, 4, EPI_PACKED, false, 0, 4, true>(x, 20, st); }, "cold W (rotating buffers)");
        { a.Wp = W[0]; float us = time_gemm<2, 4, EPI_PACKED, false, 0, 4, true>(a, 50, st); report("warm W (one buffer, MALL)", us); }
    }
    return 0;
}
