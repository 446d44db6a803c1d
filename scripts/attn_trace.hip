// Dev tool: per-workgroup timeline of k_attn_decode<64,1> (s_memtime): entry, prologue operands landed, q/k/v finished (barrier),
// cache streamed, exit.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWMAR_ATT_TRACE scripts/attn_trace.hip \
//     wmar_amd/csrc/keytable.cpp wmar_amd/csrc/watermark.hip -o scripts/attn_trace.bin
#define WMAR_ATT_TRACE 1
#include "../wmar_amd/csrc/gpt.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace wmar;

int main(int argc, char** argv) {
    const int D = 1536, H = 24, B = 64, MT = 2, Tmax = 256, S = argc > 2 ? atoi(argv[2]) : 7;
    const int T = argc > 1 ? atoi(argv[1]) : 128;
    const int NL = 8;
    const int RM = argc > 3 ? atoi(argv[3]) : 1;
    hipStream_t st; hipStreamCreate(&st);
    const size_t kv = (size_t)B * H * Tmax * 64;
    std::vector<float*> K(NL), V(NL);
    for (int l = 0; l < NL; ++l) { hipMalloc(&K[l], kv * 4); hipMalloc(&V[l], kv * 4); hipMemset(K[l], 0x3c, kv * 4); hipMemset(V[l], 0x3c, kv * 4); }
    const size_t act3 = (size_t)(3 * D / 8) * MT * 64;
    float4 *pieces, *y; double* stats; float *c1, *bias; int* pos; unsigned long long* tr;
    hipMalloc(&pieces, act3 * 16 * 8); hipMemset(pieces, 0x3c, act3 * 16 * 8);
    hipMalloc(&y, (size_t)D / 8 * MT * 64 * 16);
    hipMalloc(&stats, 8 * 64 * 2 * 8); hipMemset(stats, 0x3c, 8 * 64 * 2 * 8);
    hipMalloc(&c1, 3 * D * 4); hipMemset(c1, 0, 3 * D * 4);
    hipMalloc(&bias, 3 * D * 4); hipMemset(bias, 0, 3 * D * 4);
    hipMalloc(&pos, 16); int hp = T - 1; hipMemcpy(pos, &hp, 4, hipMemcpyHostToDevice);
    const int nwg = B * H;
    hipMalloc(&tr, (size_t)nwg * 5 * 8);
    AttnArgs t{};
    t.qkv_slabs = pieces; t.slab_stride = (long long)act3; t.S = S; t.stats = stats; t.n_chunks = S; t.K = D; t.invK = 1.0 / (double)D; t.c1 = c1; t.bias = bias;
    t.rowmajor = RM; t.y = y; t.pos_dev = pos; t.D = D; t.H = H; t.Tmax = Tmax; t.MT = MT; t.scale = 0.125f; t.trace = tr;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, st);
        for (int l = 0; l < NL; ++l) {
            t.kcache = K[l]; t.vcache = V[l];
            if (RM) hipLaunchKernelGGL((k_attn_decode<64, 1, false, 0, true>), dim3(nwg), dim3(64), 0, st, t);
            else hipLaunchKernelGGL((k_attn_decode<64, 1, false, 0, false>), dim3(nwg), dim3(64), 0, st, t);
        }
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d: %.2f us per launch (T=%d, S=%d)\n", rep, ms * 1000.f / NL, T, S);
    }
    std::vector<unsigned long long> h((size_t)nwg * 5);
    hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
    double a[4] = {0, 0, 0, 0}; unsigned long long mx[4] = {0, 0, 0, 0};
    for (int i = 0; i < nwg; ++i)
        for (int k = 0; k < 4; ++k) {
            unsigned long long d = h[(size_t)i * 5 + k + 1] - h[(size_t)i * 5 + k];
            a[k] += d; mx[k] = std::max(mx[k], d);
        }
    unsigned long long t0 = ~0ull, t0x = 0, t4 = 0, t4n = ~0ull;
    for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, h[(size_t)i * 5]); t0x = std::max(t0x, h[(size_t)i * 5]); t4 = std::max(t4, h[(size_t)i * 5 + 4]); t4n = std::min(t4n, h[(size_t)i * 5 + 4]); }
    printf("rowmajor %d: first wave in at 0, last wave in at %llu, first wave out at %llu, last wave out at %llu cycles\n", RM, t0x - t0, t4n - t0, t4 - t0);
    printf("cycles avg (max): prologue loads %.0f (%llu) | finish q/k/v + barrier %.0f (%llu) | stream %.0f (%llu) | reduce + store %.0f (%llu)\n",
           a[0] / nwg, mx[0], a[1] / nwg, mx[1], a[2] / nwg, mx[2], a[3] / nwg, mx[3]);
    return 0;
}
