// Dev tool: k_fc1x (24-column tiles, 256 workgroups) against k_gemm<2,4,EPI_GELU,LN> (32-column tiles, 192 workgroups) on the
// same inputs: max |difference| of the packed hidden activation and time per launch (weights cycling through 12 buffers).
//#define WMAR_FX_TRACE 1
#include "../wmar_amd/csrc/gpt.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace wmar;
int main() {
    const int D = 1536, N = 4 * D, K = D, MT = 2, NL = 12;
    hipStream_t st; (void)hipStreamCreate(&st);
    std::vector<float> hW((size_t)N * K), hX((size_t)K / 8 * MT * 64 * 4), hb(N), hg(K);
    srand(1);
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.08f;
    for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hg) v = 1.f + (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    float *W, *gamma, *bias, *c1; float4 *X, *o1, *o2; double* stats;
    (void)hipMalloc(&W, hW.size() * 4); (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&gamma, K * 4); (void)hipMemcpy(gamma, hg.data(), K * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&bias, N * 4); (void)hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&c1, N * 4);
    (void)hipMalloc(&X, hX.size() * 4); (void)hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    const size_t hid = (size_t)N / 8 * MT * 64;
    (void)hipMalloc(&o1, hid * 16); (void)hipMalloc(&o2, hid * 16);
    const int nch = 12;
    std::vector<double> hs((size_t)nch * 64 * 2);
    for (int c = 0; c < nch; ++c) for (int m = 0; m < 64; ++m) { hs[((size_t)c * 64 + m) * 2] = 0.01 * (m - 30) * K / nch; hs[((size_t)c * 64 + m) * 2 + 1] = (1.3 + 0.01 * m) * K / nch; }
    (void)hipMalloc(&stats, hs.size() * 8); (void)hipMemcpy(stats, hs.data(), hs.size() * 8, hipMemcpyHostToDevice);
    fold_bias(W, nullptr, gamma, c1, N, K, st);
    std::vector<float4*> Wp(NL), W16(NL); std::vector<float2*> W8(NL);
    for (int l = 0; l < NL; ++l) {
        (void)hipMalloc(&Wp[l], (size_t)N * K * 4); (void)hipMalloc(&W16[l], (size_t)N / 24 * (K / 16) * 64 * 16); (void)hipMalloc(&W8[l], (size_t)N / 24 * (K / 16) * 64 * 8);
        pack(W, Wp[l], N, K, 0, st, gamma);
        const long long total = (long long)(N / 24) * (K / 16) * 64;
        hipLaunchKernelGGL(k_pack_fc1x, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, gamma, W16[l], W8[l], N, K);
    }
    GemmArgs g{};
    g.Xp = X; g.bias = bias; g.c1 = c1; g.KB = K / 8; g.NT = N / 32; g.MT = MT; g.S = 1; g.stats = stats; g.n_chunks = nch; g.K = K;
    g.out_packed = o1; g.B = 64;
    Fc1xArgs f{};
    f.Xp = X; f.bias = bias; f.c1 = c1; f.stats = stats; f.n_chunks = nch; f.K = K; f.out = o2; f.KU = K / 16;
    unsigned long long* tr; (void)hipMalloc(&tr, 256 * 4 * 8); (void)hipMemset(tr, 0, 256 * 4 * 8); f.trace = tr;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, st);
            for (int l = 0; l < NL; ++l) {
                if (which == 0) { g.Wp = Wp[l]; launch_gemm<2, 4, EPI_GELU, true>(g, st); }
                else { f.W16 = W16[l]; f.W8 = W8[l]; hipLaunchKernelGGL(k_fc1x, dim3(N / 24), dim3(256), 0, st, f); }
            }
            (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s: %.2f us per launch\n", which ? "k_fc1x (256 x 24 columns)" : "k_gemm (192 x 32 columns)", ms * 1000.f / NL);
        }
    { std::vector<unsigned long long> h(256 * 4); (void)hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
      double s1 = 0, s2 = 0, s3 = 0; for (int i = 0; i < 256; ++i) { s1 += h[i*4+1]-h[i*4]; s2 += h[i*4+2]-h[i*4+1]; s3 += h[i*4+3]-h[i*4+2]; }
      printf("k_fc1x ticks (wave 0 avg): first operands %.0f, main loop %.0f, epilogue %.0f\n", s1/256, s2/256, s3/256); }
    std::vector<float> a(hid * 4), b(hid * 4);
    (void)hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, mag = 0; size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - b[i]); if (d > mx) mx = d; if (fabs(a[i]) > mag) mag = fabs(a[i]); if (d > 1e-4) ++bad; }
    printf("max |k_fc1x - k_gemm| = %.3e (max |value| %.3f), entries off by > 1e-4: %zu of %zu; hipGetLastError: %s\n", mx, mag, bad, a.size(), hipGetErrorString(hipGetLastError()));
    return 0;
}
