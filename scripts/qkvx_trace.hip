// Dev tool: per-wave timeline of k_qkvx (s_memtime stamps of the multiplying waves): entry, chunk 0 staged (first MFMA),
// main loop done, pieces stored.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWMAR_QX_TRACE scripts/qkvx_trace.hip \
//     wmar_amd/csrc/keytable.cpp wmar_amd/csrc/watermark.hip -o scripts/qkvx_trace.bin
#define WMAR_QX_TRACE 1
#include "../wmar_amd/csrc/gpt.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace wmar;

int main(int argc, char** argv) {
    const int D = 1536, MT = 2, S = argc > 1 ? atoi(argv[1]) : 7; const int S_IN = argc > 2 ? atoi(argv[2]) : 6;
    const int KB = D / 8, NT = 3 * D / 32;
    hipStream_t st; hipStreamCreate(&st);
    const int NL = argc > 3 ? atoi(argv[3]) : 12;                                           // distinct weight buffers: stream from HBM, not the MALL
    std::vector<float4*> W(NL);
    for (auto& p : W) { hipMalloc(&p, (size_t)3 * D * D * 4); hipMemset(p, 0x3c, (size_t)3 * D * D * 4); }
    float4 *x, *x2, *slabs, *out; float* bias; double* stats; unsigned long long* tr;
    const size_t act = (size_t)KB * MT * 64;
    hipMalloc(&x, act * 16); hipMemset(x, 0x3c, act * 16);
    hipMalloc(&x2, act * 16);
    hipMalloc(&slabs, act * 16 * 8); hipMemset(slabs, 0x3c, act * 16 * 8);
    hipMalloc(&out, act * 16 * 3 * 8);
    hipMalloc(&bias, D * 4); hipMemset(bias, 0, D * 4);
    hipMalloc(&stats, 8 * 64 * 2 * 8);
    const int total = (NT / 4) * S, cap = (total + 7) / 8;
    hipMalloc(&tr, (size_t)total * 4 * 4 * 8);
    unsigned long long* trc; hipMalloc(&trc, (size_t)total * 4 * 8 * 8);
    QkvxArgs q{};
    q.x_in = x; q.x_out = x2; q.slabs = slabs; q.slab_stride = (long long)act; q.n_hi = 16; q.bias = bias; q.stats = stats;
    q.out = out; q.out_stride = 3 * (long long)act; q.KB = KB; q.NT = NT; q.S = S; q.cap = cap; q.trace = tr; q.trace_chunks = trc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(tr, 0, (size_t)total * 4 * 4 * 8);
        hipEventRecord(e0, st);
        for (int l = 0; l < NL; ++l) { q.Wp = W[l]; launch_qkvx(q, MT, S_IN, st); }
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d: %.2f us per launch (%d launches, S=%d, %d workgroups)\n", rep, ms * 1000.f / NL, NL, S, total);
    }
    std::vector<unsigned long long> h((size_t)total * 16);
    hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, tend = 0;
    for (int i = 0; i < total * 4; ++i) { t0 = std::min(t0, h[i * 4]); tend = std::max(tend, h[i * 4 + 3]); }
    printf("last launch: span first entry -> last exit %llu ticks (100 MHz: %.2f us)\n", tend - t0, (tend - t0) / 100.0);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0; unsigned long long mx0 = 0, mx1 = 0, mx2 = 0, mx3 = 0;
    for (int i = 0; i < total * 4; ++i) {
        unsigned long long s0 = h[i * 4] - t0, s1 = h[i * 4 + 1] - h[i * 4], s2 = h[i * 4 + 2] - h[i * 4 + 1], s3 = h[i * 4 + 3] - h[i * 4 + 2];
        a0 += s0; a1 += s1; a2 += s2; a3 += s3;
        mx0 = std::max(mx0, s0); mx1 = std::max(mx1, s1); mx2 = std::max(mx2, s2); mx3 = std::max(mx3, s3);
    }
    const int n = total * 4;
    printf("ticks avg (max): entry offset %.0f (%llu) | to first MFMA %.0f (%llu) | main loop %.0f (%llu) | stores %.0f (%llu)\n",
           a0 / n, mx0, a1 / n, mx1, a2 / n, mx2, a3 / n, mx3);
    for (int j = 0; j < total; j += total / 9)
        printf("wg %3d (s=%d g=%2d) wave0: entry %5llu  +stage %5llu  +loop %5llu  +store %5llu\n", j, j / (NT / 4), j % (NT / 4),
               h[(size_t)j * 16] - t0, h[(size_t)j * 16 + 1] - h[(size_t)j * 16], h[(size_t)j * 16 + 2] - h[(size_t)j * 16 + 1],
               h[(size_t)j * 16 + 3] - h[(size_t)j * 16 + 2]);
    std::vector<unsigned long long> hc((size_t)total * 32);
    hipMemcpy(hc.data(), trc, hc.size() * 8, hipMemcpyDeviceToHost);
    double avg[8] = {0}; int cnt[8] = {0};
    for (int i = 0; i < total * 4; ++i) for (int c = 0; c < 8; ++c) if (c == 0 || hc[(size_t)i * 8 + c]) { avg[c] += hc[(size_t)i * 8 + c]; cnt[c]++; }
    printf("chunk start (ticks after the first barrier), avg over waves:");
    for (int c = 0; c < 8; ++c) printf(" %.0f", cnt[c] ? avg[c] / cnt[c] : 0.0);
    printf("\n");
    return 0;
}
