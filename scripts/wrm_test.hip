// Dev check (round 6): wave_reduce_many with the gfx950 lane-swap steps against the all-shuffle form, bit for bit, on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I wmar_amd/csrc -o scripts/wrm_test.bin scripts/wrm_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "decode_small.h"
namespace wmar { void set_error(const char*, ...) {} }
using namespace wmar;

template <int R, typename T>
__device__ T reduce_shuffle(T (&v)[R], int lane, int* idx_out) {      // the generic form, every step a select + shuffle
    constexpr int P = R <= 1 ? 1 : R <= 2 ? 2 : R <= 4 ? 4 : R <= 8 ? 8 : R <= 16 ? 16 : R <= 32 ? 32 : 64;
    T t[P];
#pragma unroll
    for (int i = 0; i < P; ++i) t[i] = i < R ? v[i] : (T)0;
    int idx = 0, cnt = P, off = 32, j = 0;
#pragma unroll
    for (int step = 0; step < 6; ++step) {
        if (cnt > 1) {
            const bool up = (lane & off) != 0;
#pragma unroll
            for (int i = 0; i < P / 2; ++i)
                if (i < cnt / 2) { const T keep = up ? t[2 * i + 1] : t[2 * i]; const T send = up ? t[2 * i] : t[2 * i + 1]; t[i] = keep + __shfl_xor(send, off); }
            idx |= (up ? 1 : 0) << j; cnt /= 2; ++j;
        } else t[0] += __shfl_xor(t[0], off);
        off >>= 1;
    }
    *idx_out = idx;
    return t[0];
}

template <int R, typename T>
__global__ void k_test(const T* in, int trials, unsigned long long* bad) {
    const int lane = threadIdx.x;
    for (int tr = blockIdx.x; tr < trials; tr += gridDim.x) {
        T a[R], b[R];
        for (int i = 0; i < R; ++i) a[i] = b[i] = in[((long long)tr * R + i) * 64 + lane];
        int ia, ib;
        const T x = wave_reduce_many<R, T>(a, lane, &ia);
        const T y = reduce_shuffle<R, T>(b, lane, &ib);
        bool same = ia == ib;
        if (sizeof(T) == 4) { float fx = (float)x, fy = (float)y; same = same && __float_as_uint(fx) == __float_as_uint(fy); }
        else { double dx = (double)x, dy = (double)y; same = same && __double_as_longlong(dx) == __double_as_longlong(dy); }
        if (!same) atomicAdd(bad, 1ull);
    }
}

template <int R, typename T>
int run(const char* name) {
    const int trials = 20000;
    std::vector<T> h((size_t)trials * R * 64);
    for (auto& v : h) v = (T)((rand() / (double)RAND_MAX - 0.5) * ((rand() & 7) ? 4.0 : 4e4));
    T* d; unsigned long long* bad; unsigned long long hb = 0;
    hipMalloc(&d, h.size() * sizeof(T)); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_test<R, T>), dim3(256), dim3(64), 0, 0, d, trials, bad);
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    printf("%s R=%d: %llu lanes differ of %d x 64\n", name, R, hb, trials);
    hipFree(d); hipFree(bad);
    return hb != 0;
}
int main() {
    int f = 0;
    f |= run<3, float>("float"); f |= run<15, float>("float"); f |= run<16, float>("float"); f |= run<24, float>("float"); f |= run<32, float>("float");
    f |= run<2, double>("double"); f |= run<10, double>("double"); f |= run<16, double>("double");
    return f;
}
