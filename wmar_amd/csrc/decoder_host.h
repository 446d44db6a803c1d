// Host-side helpers shared by the decode engines (gpt.hip, rar.hip).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "decoder_kernels.h"

namespace wmar {

inline int mt_for(int64_t B) { return B <= 32 ? 1 : (B <= 64 ? 2 : 4); }

struct TensorMap {
    std::map<std::string, const void*> m;
    const float* get(const std::string& k) const {
        auto it = m.find(k);
        return it == m.end() ? nullptr : (const float*)it->second;
    }
};

inline int pack(const float* W, float4* Wp, int N, int K, int nt_off, hipStream_t st,
         const float* gamma = nullptr) {
    long long total = (long long)(N / 32) * (K / 8) * 64;
    hipLaunchKernelGGL(k_pack_linear, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, Wp, N, K, nt_off, K / 8,
                       gamma);
    return launch_status("k_pack_linear");
}

// dst[0..N) = bias + W beta
inline int fold_bias(const float* W, const float* bias, const float* beta, float* dst, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL(k_fold_bias, dim3((unsigned)N), dim3(64), 0, st, W, bias, beta, dst, K);
    return launch_status("k_fold_bias");
}

template <typename Engine>
int copy_vec(Engine* g, float** dst, const float* src, size_t n, hipStream_t st) {
    if (int rc = g->alloc(dst, n)) return rc;
    WMAR_HIP_CHECK(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return WMAR_OK;
}

template <int MTW, int NW, int EPI, bool LN, int ABL = 0, int U = GEMM_STAGE, bool ROT = true, int NTW = 1>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
    const int grid = NTW > 1 ? (a.NT / NTW) * (a.MT / MTW) : a.NT * (a.MT / MTW) * a.S + a.n_hi;
    const size_t lds = (size_t)NW * MTW * NTW * 16 * 64 * sizeof(float);
    if (NTW > 1 && (a.NT % NTW != 0 || a.S != 1 || a.n_hi != 0)) { set_error("k_gemm: %d column tiles per workgroup need S == 1 and NT %% %d == 0", NTW, NTW); return WMAR_EINVAL; }
    if (lds > 64 * 1024) {      // more than the default dynamic LDS limit: opt in once per instantiation
        static bool done = false;
        if (!done) { (void)hipFuncSetAttribute((const void*)k_gemm<MTW, NW, EPI, LN, ABL, U, ROT, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    }
    hipLaunchKernelGGL((k_gemm<MTW, NW, EPI, LN, ABL, U, ROT, NTW>), dim3((unsigned)grid), dim3(NW * 64), lds, st, a);
    return launch_status("k_gemm");
}

// Split-K factor for a GEMM whose partial slabs are folded by a later kernel: pick the S that
// fills the 256 CUs most evenly (whole "rounds" of workgroups), keeping >= 8 k-blocks per wave.
inline int pick_split(int tiles, int KB, int NW) {
    int best = 1;
    double best_eff = 0.0;
    for (int S = 1; S <= MAX_SLABS; ++S) {
        if (KB / (S * NW) < 8 && S > 1) break;
        const int wgs = tiles * S;
        const int rounds = (wgs + 255) / 256;
        const double eff = (double)wgs / (rounds * 256.0);
        if (eff > best_eff + 0.02) { best_eff = eff; best = S; }
    }
    return best;
}

// Row tiles per workgroup: two 32-row tiles share every weight fragment (half the operand
// traffic per MFMA); a single tile when the batch has only one.
template <int EPI, bool LN>
int gemm_dispatch(GemmArgs a, bool allow_split, hipStream_t st) {
    constexpr int NW = 4;
    // four row tiles per workgroup once two would need more than one workgroup per CU: a second
    // workgroup on a CU shares its MFMA pipes, so the launch takes as long as the busiest CU
    if (a.MT % 4 == 0 && a.NT * (a.MT / 2) > 256) {
        a.S = allow_split ? pick_split(a.NT * (a.MT / 4), a.KB, NW) : 1;
        return launch_gemm<4, NW, EPI, LN, 0, 2>(a, st);
    }
    if (a.MT % 2 == 0) {
        a.S = allow_split ? pick_split(a.NT * (a.MT / 2), a.KB, NW) : 1;
        return launch_gemm<2, NW, EPI, LN>(a, st);
    }
    a.S = allow_split ? pick_split(a.NT * a.MT, a.KB, NW) : 1;
    return launch_gemm<1, NW, EPI, LN>(a, st);
}

// split-K GEMM writing partial slabs; reports the S it used
inline int gemm_split(GemmArgs a, int* S_out, hipStream_t st, int force_S = 0) {
    constexpr int NW = 4;
    if (a.MT % 2 == 0) {
        a.S = force_S > 0 ? force_S : pick_split(a.NT * (a.MT / 2), a.KB, NW);
        *S_out = a.S;
        return launch_gemm<2, NW, EPI_PACKED, false>(a, st);
    }
    a.S = force_S > 0 ? force_S : pick_split(a.NT * a.MT, a.KB, NW);
    *S_out = a.S;
    return launch_gemm<1, NW, EPI_PACKED, false>(a, st);
}

// chunks of <= WMAR_STAT_CHUNK k-blocks: one k_resid_stats workgroup (4 waves x up to 4 blocks) per chunk and row tile
#ifndef WMAR_STAT_CHUNK
#define WMAR_STAT_CHUNK 16
#endif
inline int stat_chunks(int KB) { const int n = (KB + WMAR_STAT_CHUNK - 1) / WMAR_STAT_CHUNK; return n <= STAT_CHUNKS_MAX ? n : (KB + 15) / 16; }


// Device allocations of one engine (freed together).
struct DeviceArena {
    std::vector<void*> allocs;
    int64_t bytes = 0;
    template <typename T>
    int alloc(T** p, size_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
        if (e != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
            return WMAR_ENOMEM;
        }
        allocs.push_back(q);
        bytes += (int64_t)(n * sizeof(T));
        *p = (T*)q;
        return WMAR_OK;
    }
    void release() {
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
    }
};

}  // namespace wmar
