// Dev tool (round 4): 64-row fp32 GEMM on the bf16 matrix pipe with the split-K reduction INSIDE the XCD.
//
// Why: every decode GEMM of rounds 1-3 obeys  t ~ 1 us + (bytes through one CU's vector-memory path) / 33 GB/s  -- L2-hit activation
// bytes cost what HBM weight bytes cost -- and whole-K / deep-K tiles make every CU read most of the 64 x K activation (FC1: 393 KB of
// activation beside 147 KB of weights).  Shallow-K wide tiles (96 columns x 384 k: 98 KB + 147 KB) halve that, but need a split-K
// reduction; a second launch or a fold in the consumer gives the gain back.  Here the reduction stays on chip: block b runs on XCD
// b % 8, an XCD (32 CUs, one L2) owns N / 8 output columns, its CUs split (column group, K slice), park their partial sums in the
// XCD's L2 (plain stores, s_waitcnt vmcnt(0): acknowledged by the L2), meet at an XCD-LOCAL barrier (L2 atomics without sc1: no
// fabric round trip) and then reduce + finish (LayerNorm algebra, bias, GELU / residual) 1/32 of the XCD's columns each.
//
//   MODE 0: one launch (phase 1, XCD barrier, phase 2)      MODE 1 + MODE 2: the two phases as two launches (A/B)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/xg_bench.hip -o scripts/xg_bench.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "xg_kernel.h"

using namespace wmar;

static void pack_w_host(const std::vector<float>& W, std::vector<float>& Wq, int N, int K) {
    const int KU = K / 16;
    Wq.resize((size_t)N * K);
    for (int tile = 0; tile < N / 32; ++tile)
        for (int ku = 0; ku < KU; ++ku)
            for (int hf = 0; hf < 2; ++hf)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = tile * 32 + (lane & 31), k = ku * 16 + 8 * (lane >> 5) + 4 * hf;
                    float* d = &Wq[((((size_t)tile * KU + ku) * 2 + hf) * 64 + lane) * 4];
                    for (int i = 0; i < 4; ++i) d[i] = W[(size_t)n * K + k + i];
                }
}
// Xh[ku][mt][lane][8]: lane holds row 32 mt + lane % 32, features 16 ku + 8 (lane / 32) + 0..7
static void pack_xh_host(const std::vector<float>& X, std::vector<float>& Xh, int K) {
    Xh.resize((size_t)64 * K);
    for (int ku = 0; ku < K / 16; ++ku)
        for (int mt = 0; mt < 2; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i)
                    Xh[((((size_t)ku * 2 + mt) * 64 + lane) * 8) + i] = X[(size_t)(mt * 32 + (lane & 31)) * K + ku * 16 + 8 * (lane >> 5) + i];
}
static unsigned short rne_bf16(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
// planes [ku][mt][piece][lane][8 bf16]
static void pack_xq_host(const std::vector<float>& X, std::vector<unsigned short>& Xq, int K) {
    Xq.resize((size_t)64 * K * 3);
    for (int ku = 0; ku < K / 16; ++ku)
        for (int mt = 0; mt < 2; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const float x = X[(size_t)(mt * 32 + (lane & 31)) * K + ku * 16 + 8 * (lane >> 5) + i];
                    const unsigned short h = rne_bf16(x); const float r = x - bf16_f(h);
                    const unsigned short m = rne_bf16(r); const float q = r - bf16_f(m);
                    const unsigned short l = rne_bf16(q);
                    const unsigned short pc[3] = {h, m, l};
                    for (int p = 0; p < 3; ++p) Xq[((((size_t)(ku * 2 + mt) * 3 + p) * 64 + lane) * 8) + i] = pc[p];
                }
}
static float xh_get(const std::vector<float>& Xh, int m, int n) {
    return Xh[((((size_t)(n >> 4) * 2 + (m >> 5)) * 64 + (m & 31) + 32 * ((n >> 3) & 1)) * 8) + (n & 7)];
}

template <int NT, int PER, int EPI, int S, int NW = 4>
static void run(const char* name, int N, int K, int G, hipStream_t st) {
    const int NL = 12, KU = K / 16, TX = N / 256;
    if (KU != S * NW * PER || TX != G * NT || G * S > 32) { printf("%s: bad shape\n", name); return; }
    std::vector<float> hW((size_t)N * K), hX((size_t)64 * K), hB(N), hC(N), hR((size_t)64 * N);
    srand(1);
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.08f;
    for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f + 0.3f;
    for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto& v : hR) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += hW[(size_t)n * K + k]; hC[n] = (float)s; }
    std::vector<float> hWq, hXh, hRh;
    pack_w_host(hW, hWq, N, K);
    pack_xh_host(hX, hXh, K);
    pack_xh_host(hR, hRh, N);
    std::vector<float4*> Wq(NL);
    for (int l = 0; l < NL; ++l) { (void)hipMalloc(&Wq[l], hWq.size() * 4); (void)hipMemcpy(Wq[l], hWq.data(), hWq.size() * 4, hipMemcpyHostToDevice); }
    float4 *Xh, *Rh, *out, *part; float *bias, *c1; double2* statp; unsigned *sync, *fail;
    (void)hipMalloc(&Xh, hXh.size() * 4); (void)hipMemcpy(Xh, hXh.data(), hXh.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned short> hXq; pack_xq_host(hX, hXq, K);
    u32x4* Xq; (void)hipMalloc(&Xq, hXq.size() * 2); (void)hipMemcpy(Xq, hXq.data(), hXq.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&Rh, hRh.size() * 4); (void)hipMemcpy(Rh, hRh.data(), hRh.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, (size_t)64 * N * 4);
    (void)hipMalloc(&part, (size_t)S * 64 * N * 4);
    (void)hipMalloc(&bias, N * 4); (void)hipMemcpy(bias, hB.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&c1, N * 4); (void)hipMemcpy(c1, hC.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&statp, (size_t)8 * 32 * 64 * 16);
    (void)hipMalloc(&sync, 8 * 64 * 4); (void)hipMemset(sync, 0, 8 * 64 * 4);
    (void)hipMalloc(&fail, 64 * 4); (void)hipMemset(fail, 0, 64 * 4);
    XgArgs a{};
    a.Xh = Xh; a.Xq = Xq; a.part = part; a.statp = statp; a.sync = sync; a.fail = fail; a.KU = KU; a.S = S; a.G = G; a.TX = TX; a.K = K;
    a.bias = bias; a.c1 = c1; a.Xres = Rh; a.out = out; a.invK = 1.0 / K;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float t_fused = 0, t_split = 0, t_p1 = 0;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0, st);
            for (int l = 0; l < NL; ++l) {
                a.Wq = Wq[l];
                if (mode == 0) hipLaunchKernelGGL((k_xg<NT, PER, EPI, 0, S, NW>), dim3(256), dim3(NW * 64), 0, st, a);
                else {
                    hipLaunchKernelGGL((k_xg<NT, PER, EPI, 1, S, NW>), dim3(256), dim3(NW * 64), 0, st, a);
                    if (mode == 1) hipLaunchKernelGGL((k_xg<NT, PER, EPI, 2, S, NW>), dim3(256), dim3(NW * 64), 0, st, a);
                }
            }
            (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 3) (mode == 0 ? t_fused : (mode == 1 ? t_split : t_p1)) = ms * 1000.f / NL;
        }
        if (mode > 1) continue;
        // check (the last launch used Wq[NL-1] == the same weights)
        std::vector<float> o((size_t)64 * N);
        (void)hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, mag = 0;
        for (int n = 0; n < N; n += 7)
            for (int m = 0; m < 64; ++m) {
                double r = 0, sm = 0, sq = 0;
                for (int k = 0; k < K; ++k) { const double x = hX[(size_t)m * K + k]; r += x * hW[(size_t)n * K + k]; sm += x; sq += x * x; }
                double ref;
                if (EPI == XG_EPI_GELU) {
                    const double mean = sm / K, rstd = 1.0 / sqrt(sq / K - mean * mean + 1e-5);
                    const double v = rstd * (r - mean * hC[n]) + hB[n];
                    ref = 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
                } else {
                    ref = hR[(size_t)m * N + n] + hB[n] + r;
                }
                const double got = xh_get(o, m, n);
                emax = fmax(emax, fabs(got - ref)); mag = fmax(mag, fabs(ref));
            }
        printf("  %s mode %d: max |out - fp64| = %.3e (max |value| %.3f)\n", name, mode, emax, mag);
        (void)hipMemset(out, 0xff, (size_t)64 * N * 4);
    }
    float t_bar = 0, t_p2 = 0;
    for (int mode = 0; mode < 2; ++mode) {       // no phase 1 at all: launch + barrier + phase 2 against launch + phase 2
        XgArgs b = a; b.G = 0; b.Wq = Wq[0];
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, st);
            for (int l = 0; l < 50; ++l) {
                if (mode == 0) hipLaunchKernelGGL((k_xg<NT, PER, EPI, 0, S, NW>), dim3(256), dim3(NW * 64), 0, st, b);
                else hipLaunchKernelGGL((k_xg<NT, PER, EPI, 2, S, NW>), dim3(256), dim3(NW * 64), 0, st, b);
            }
            (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (mode == 0 ? t_bar : t_p2) = ms * 1000.f / 50;
        }
    }
#ifdef XG_STAMP
    {
        unsigned long long* tr; (void)hipMalloc(&tr, 256 * 16 * 8); (void)hipMemset(tr, 0, 256 * 16 * 8);
        XgArgs b = a; b.trace = tr;
        for (int l = 0; l < NL; ++l) { b.Wq = Wq[l]; hipLaunchKernelGGL((k_xg<NT, PER, EPI, 1, S, NW>), dim3(256), dim3(NW * 64), 0, st, b); }
        (void)hipStreamSynchronize(st);
        unsigned long long h[256 * 16]; (void)hipMemcpy(h, tr, sizeof h, hipMemcpyDeviceToHost);
        printf("  %s phase-1 timeline of wave 0 (ticks from its start, mean over active workgroups): loads issued, step starts..., loop end, stored\n   ", name);
        for (int i = 1; i < PER + 4; ++i) { double sm = 0; int n = 0; for (int bk = 0; bk < 256; ++bk) if ((bk >> 3) < G * S) { sm += (double)h[bk * 16 + i]; ++n; } printf(" %.0f", sm / n); }
        printf("\n");
        (void)hipFree(tr);
    }
#endif
#ifdef XG_TRACE
    {
        unsigned long long* tr; (void)hipMalloc(&tr, 256 * 6 * 8);
        XgArgs b = a; b.G = 0; b.Wq = Wq[0]; b.trace = tr;
        for (int l = 0; l < 3; ++l) hipLaunchKernelGGL((k_xg<NT, PER, EPI, 0, S, NW>), dim3(256), dim3(NW * 64), 0, st, b);
        (void)hipStreamSynchronize(st);
        unsigned long long h[256 * 6]; (void)hipMemcpy(h, tr, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull; for (int i = 0; i < 256; ++i) if (h[i * 6] < t0) t0 = h[i * 6];
        printf("  barrier trace (XCD group 0, ticks from the first workgroup's arrival): block: enter, before atomic, after atomic, released | old, polls\n");
        for (int i = 0; i < 256; i += 8) printf("   %3d: %6llu %6llu %6llu %6llu | %2llu %llu\n", i, h[i*6] - t0, h[i*6+1] - t0, h[i*6+2] - t0, h[i*6+3] - t0, h[i*6+4], h[i*6+5]);
        (void)hipFree(tr);
    }
#endif
    printf("  %s without phase 1: launch + XCD barrier + phase 2 %.2f us, launch + phase 2 %.2f us\n", name, t_bar, t_p2);
    unsigned hf[64]; (void)hipMemcpy(hf, fail, sizeof hf, hipMemcpyDeviceToHost);
    printf("%s N=%d K=%d S=%d G=%d (%d of 32 CUs per XCD): fused %.2f us, two launches %.2f us, phase 1 alone %.2f us (%.1f MB of weights); "
           "fail flags: placement %u, timeout %u; %s\n", name, N, K, S, G, S * G, t_fused, t_split, t_p1, N * (double)K * 4 / 1e6, hf[0], hf[1],
           hipGetErrorString(hipGetLastError()));
    for (auto p : Wq) (void)hipFree(p);
    (void)hipFree(Xh); (void)hipFree(Rh); (void)hipFree(out); (void)hipFree(part); (void)hipFree(bias); (void)hipFree(c1); (void)hipFree(statp);
    (void)hipFree(sync); (void)hipFree(fail);
}

__global__ void k_probe(unsigned* o) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) o[blockIdx.x] = x & 15;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t st; (void)hipStreamCreate(&st);
    {
        unsigned* o; (void)hipMalloc(&o, 1024 * 4);
        int bad = 0, cnt[16] = {0};
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL(k_probe, dim3(256), dim3(256), 0, st, o);
            unsigned h[256]; (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
            for (int i = 0; i < 256; ++i) { if (h[i] != (unsigned)(i & 7)) ++bad; if (rep == 0) ++cnt[h[i] & 15]; }
        }
        { unsigned h[256]; (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost); printf("xcc of blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", h[i]); printf("\n"); }
        printf("placement probe: %d of %d blocks NOT on XCD blockIdx %% 8; blocks per XCC id:", bad, 20 * 256);
        for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
        printf("\n");
    }
    run<3, 3, XG_EPI_GELU, 4, 8>("fc1 8w", 6144, 1536, 8, st);
    run<3, 3, XG_EPI_RESID, 16, 8>("fc2 8w", 1536, 6144, 2, st);
    run<3, 3, XG_EPI_GELU, 4, 8>("qkv-shape 8w", 4608, 1536, 6, st);
    run<3, 6, XG_EPI_GELU, 4>("fc1", 6144, 1536, 8, st);
    run<3, 6, XG_EPI_RESID, 16>("fc2", 1536, 6144, 2, st);
    run<2, 3, XG_EPI_RESID, 8>("proj", 1536, 1536, 3, st);
    run<3, 6, XG_EPI_GELU, 4>("qkv-shape", 4608, 1536, 6, st);
    return 0;
}
