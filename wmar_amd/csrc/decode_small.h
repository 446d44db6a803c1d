// Small-batch decode step (1..10 sequences) of the Taming minGPT engine: weight STREAMING kernels, no matrix cores.
//
// Reference: deps/taming/modules/transformer/mingpt.py:69-95 (attention), :112-122 (Block), :183-214 (forward_with_past) at the batch
// sizes the reference itself runs (configs/taming_generate.json: batch 5; BASELINE configs[0]: batch 1).
//
// Why a second path.  At <= 12 rows a decode step is the 5.54 GB weight stream and nothing else (0.69 ms at 8 TB/s): every weight
// feeds <= 12 multiply-adds, so a 32-row MFMA tile is 62-97 % padding and the fp32 matrix pipe itself becomes the bound (32x32x2:
// 0.56 ms per step at peak), and the big-batch plan's split-K slabs + fold launches are pure fixed cost (round 5: 2.7 ms per step at
// batch 5 = 0.25 of the stream).  Here a step is FIVE launches per layer and no partial sums ever leave a workgroup:
//
//   k_sgemv<QKV>   LayerNorm 1 + QKV projection + bias; k / v rows go straight into the cache, q into a [rows][D] buffer
//   k_sattn        one workgroup per (sequence, head): softmax(q K^T / 8) V over the cached rows, four waves split the rows
//   k_sgemv<PROJ>  x += y Wp^T + b            (residual stream updated in place: a workgroup owns its columns)
//   k_sgemv<FC1>   LayerNorm 2 + FC1 + bias + exact-erf GELU
//   k_sgemv<FC2>   x += h W2^T + b
//
// k_sgemv: the weights stay in the checkpoint's own row-major [N][K] layout; a wave-wide 16-byte load is 1 KiB = 256 consecutive k of
// ONE output column, so a workgroup can own ANY number of columns and every launch is exactly 256 workgroups (18 / 6 / 24 / 6 / 64
// columns for QKV / proj / FC1 / FC2 / head at n_embd 1536) with whole K inside the workgroup.  K is cut into segments of 768 (three
// loads per lane); a wave owns one segment and a column range, keeps its slice of the <= 12 activation rows in REGISTERS (12 VGPRs per
// row, LayerNorm applied once while they are loaded: every workgroup recomputes the row statistics from the 6 KB rows it reads
// anyway), and turns every weight load into 4 x rows fused multiply-adds on the vector ALU (8 rows: 45 % of the VALU rate at
// the HBM rate a CU can get).  Lane-partial dot products of a group of CG columns x rows meet in ONE transposing butterfly
// (values halve while lane distance halves: ~CG x rows shuffles instead of 6 per value), the segments' partial sums meet in LDS in a
// fixed order -- deterministic, no atomics.  Weight loads are non-temporal and double-buffered by column group.  The two widest steps
// of every butterfly run on gfx950's v_permlane32_swap / v_permlane16_swap (tb_pair below).
#pragma once
#include "decoder_kernels.h"

namespace wmar {

// The arithmetic of this file is PINNED: no implicit fp contraction (the compiler's choice between a fused and an unfused a * b + c
// follows its scheduling context -- round 6: two builds that differed only in how a wave reduction moved data returned LayerNorm
// statistics one ulp apart for one row in ~40, because the sum of squares was fused in one and not in the other); every fused
// multiply-add below is written as one (fmaf / __builtin_elementwise_fma).  Restored at the end of decode_persist.h / this file.
#pragma clang fp contract(off)

enum { SG_QKV = 0, SG_PROJ = 1, SG_FC1 = 2, SG_FC2 = 3, SG_HEAD = 4 };
constexpr int SG_CH = 3;                 // 1-KiB loads per (column, segment): segment = 768 k
constexpr int SG_SEG = SG_CH * 256;
constexpr int SG_MAX_ROWS = 12;              // up to 12 rows (generate.py's default batch size is 10) since the lane-swap butterflies: ~+1 us per row and layer

struct SgArgs {
    const float* W;        // [N][K] row-major (the checkpoint's layout)
    const float* bias;     // [N] or null
    const float* x;        // [rows][K] row-major input rows
    const float* gamma;    // [K] LayerNorm weight in front of the Linear (QKV, FC1, head); `bias` is then b + W beta and
    const float* c1;       // [N] c1[n] = sum_k W[n][k] gamma[k] (the engine's folded vectors, shared with the matrix-core plan)
    float* out;            // PROJ / FC2: residual stream [rows][N], updated in place; FC1: hidden [rows][N]; HEAD: logits [rows][N]; QKV: q [rows][D]
    float* kcache;         // QKV: this layer's [Bmax][H][Tmax][64]
    float* vcache;
    const int* pos_dev;
    int N, K;
    int D, H, Tmax;
};

// Sum of R per-lane values over the 64 lanes of a wave, all R at once: while the lane distance halves the value count halves (the
// lane keeps one of a pair and sends the other), so the whole reduction is P - 1 + (6 - log2 P) shuffles for P = R rounded up to a
// power of two, instead of 6 R.  On return v[0] of lane l is the total of value `idx`, idx = sum_j bit(l, 5 - j) << j over the
// log2 P halving steps (lanes that differ only in the low bits hold copies).  Fixed order: the result is a function of the inputs only.
// The two widest steps of the butterfly on gfx950's lane-swap instructions: v_permlane32_swap exchanges lanes 32..63 of its first
// operand with lanes 0..31 of its second (v_permlane16_swap: the odd 16-lane rows of the first with the even rows of the second), so
// swap(a, b) followed by ONE add leaves a's pair sums in the lower lanes and b's in the upper ones -- the keep / send selects and the
// LDS-crossbar shuffle of the generic step (2 v_cndmask + ds_bpermute + add per pair) become one VALU swap + add.  Same operands,
// same order: bit-identical results.
template <int OFF>
__device__ __forceinline__ float tb_pair(float a, float b) {
    static_assert(OFF == 32 || OFF == 16, "lane-swap steps");
    if (OFF == 32) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int OFF>
__device__ __forceinline__ double tb_pair(double a, double b) {
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    unsigned lo0, lo1, hi0, hi1;
    if (OFF == 32) {
        const auto l = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
        lo0 = l[0]; lo1 = l[1]; hi0 = h[0]; hi1 = h[1];
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
        lo0 = l[0]; lo1 = l[1]; hi0 = h[0]; hi1 = h[1];
    }
    return __longlong_as_double((long long)(((unsigned long long)hi0 << 32) | lo0)) + __longlong_as_double((long long)(((unsigned long long)hi1 << 32) | lo1));
}

template <int R, typename T>
__device__ __forceinline__ T wave_reduce_many(T (&v)[R], int lane, int* idx_out) {
    constexpr int P = R <= 1 ? 1 : R <= 2 ? 2 : R <= 4 ? 4 : R <= 8 ? 8 : R <= 16 ? 16 : R <= 32 ? 32 : 64;
    static_assert(R <= 64, "at most 64 values per lane");
    T t[P];
#pragma unroll
    for (int i = 0; i < P; ++i) t[i] = i < R ? v[i] : (T)0;
    int idx = 0;
    int cnt = P, off = 32, j = 0;
#pragma unroll
    for (int step = 0; step < 6; ++step) {
        if (cnt > 1) {
            const bool up = (lane & off) != 0;
#pragma unroll
            for (int i = 0; i < P / 2; ++i) {
                if (i < cnt / 2) {
#ifndef WMAR_SG_NO_SWAP
                    if (step == 0) { t[i] = tb_pair<32>(t[2 * i], t[2 * i + 1]); continue; }
                    if (step == 1) { t[i] = tb_pair<16>(t[2 * i], t[2 * i + 1]); continue; }
#endif
                    const T keep = up ? t[2 * i + 1] : t[2 * i];
                    const T send = up ? t[2 * i] : t[2 * i + 1];
                    t[i] = keep + __shfl_xor(send, off);
                }
            }
            idx |= (up ? 1 : 0) << j;
            cnt /= 2; ++j;
        } else {
#ifndef WMAR_SG_NO_SWAP
            if (step == 0) t[0] = tb_pair<32>(t[0], t[0]);
            else if (step == 1) t[0] = tb_pair<16>(t[0], t[0]);
            else
#endif
            t[0] += __shfl_xor(t[0], off);
        }
        off >>= 1;
    }
    *idx_out = idx;
    return t[0];
}
template <int R> constexpr int sg_log2p() { return R <= 1 ? 0 : R <= 2 ? 1 : R <= 4 ? 2 : R <= 8 ? 3 : R <= 16 ? 4 : R <= 32 ? 5 : 6; }

// grid = ceil(N / columns per workgroup); block = 256 (512 for FC2: eight K segments).  G column groups of CG columns per wave, at
// compile time: the group loop is unrolled so that exactly the weights that are used are requested (a run-time trip count would need
// either a branch around the prefetch -- hipcc then waits for BOTH buffers at the merge -- or clamped dummy loads, 2 x the bytes at G = 1).
template <int NB, int CG, int G, int ROLE>
__global__ __launch_bounds__(ROLE == SG_FC2 ? 512 : 256) void k_sgemv(SgArgs a) {
    constexpr int CH = SG_CH;
    constexpr bool LN = ROLE == SG_QKV || ROLE == SG_FC1 || ROLE == SG_HEAD;
    constexpr int NW = ROLE == SG_FC2 ? 8 : 4;
    constexpr int NSEG = ROLE == SG_FC2 ? 8 : 2;        // K = NSEG * 768
    constexpr int NSPLIT = NW / NSEG;                     // waves that share a segment split the workgroup's columns
    constexpr int R = CG * NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
    double* lnred = (double*)sg_smem;                     // [NW][2 NB] (LN roles)
    float* part = (float*)(sg_smem + NW * 2 * SG_MAX_ROWS * sizeof(double));   // [NSEG][cols_wg][NB]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = w / NSPLIT, chalf = w % NSPLIT;
    constexpr int cols_wave = CG * G, cols_wg = NSPLIT * cols_wave;
    const int n0 = blockIdx.x * cols_wg + chalf * cols_wave;
    const long long K = a.K;
    const float* wbase = a.W + (long long)seg * SG_SEG + lane * 4;

    // weight groups in flight per wave (9 or 12 KiB each).  TWO at every row count: a third one (round 6, first version) cost registers
    // the row pairs need -- 6 / 7 rows 2.03 / 2.17 -> 1.83 / 1.90 ms per step without it, 1..5 rows equal or 1-2 % better
    constexpr int NBUF = 2;
    float4 wbuf[NBUF][CG * CH];
#define WMAR_SG_LOADW(BUF, GI)                                                                 \
    {                                                                                          \
        _Pragma("unroll") for (int c = 0; c < CG; ++c) {                                       \
            int n_ = n0 + (GI) * CG + c;                                                       \
            n_ = n_ < a.N ? n_ : a.N - 1;                                                      \
            const float4* p_ = (const float4*)(wbase + (long long)n_ * K);                     \
            _Pragma("unroll") for (int ch = 0; ch < CH; ++ch) BUF[c * CH + ch] = ld_nt(p_ + ch * 64); \
        }                                                                                      \
    }

    // the epilogue's operands (bias, the residual value a column owner adds to) are requested here, not behind the reduction: a
    // dependent global round trip at the end of every launch otherwise (~1 us of each launch's ~4 us fixed cost)
    constexpr int EPT = ROLE == SG_HEAD ? 1 : (cols_wg * NB + NW * 64 - 1) / (NW * 64);      // epilogue values per thread (2 for FC1 at 11 / 12 rows)
    float ebias[EPT], eold[EPT], ec1[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int et = threadIdx.x + e * NW * 64;
        const int eb = et / cols_wg, ecol = et - eb * cols_wg;
        const int en = blockIdx.x * cols_wg + ecol;
        const bool eon = ROLE != SG_HEAD && et < cols_wg * NB && en < a.N;
        ebias[e] = 0.f; eold[e] = 0.f; ec1[e] = 0.f;
        if (eon) {
            ebias[e] = a.bias[en];
            if (LN) ec1[e] = a.c1[en];
            if (ROLE == SG_PROJ || ROLE == SG_FC2) eold[e] = a.out[(long long)eb * a.N + en];
        }
    }

    // this wave's slice of the input rows (and of the LayerNorm parameters)
    float4 xr[CH][NB];
    {
        const float* xb = a.x + (long long)seg * SG_SEG + lane * 4;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) xr[ch][b] = *((const float4*)(xb + (long long)b * K) + ch * 64);
    }
    // the rows are on the critical path (statistics, then the first multiply); the first weight groups go out right behind them
    WMAR_SG_LOADW(wbuf[0], 0)
    if (G > 1) WMAR_SG_LOADW(wbuf[1], 1)
    if (G > 2) WMAR_SG_LOADW(wbuf[2], 2)
    // roles without LayerNorm: every request above goes out before any arithmetic below (the scheduler otherwise sinks the weight loads
    // behind the first multiplies to save registers: FC2 9.1 -> 10.9 us per launch at 5 rows); the LayerNorm roles are faster WITHOUT the
    // fence (QKV 8.0 against 8.8, FC1 8.8 against 9.6: their statistics arithmetic interleaves with the issue of the requests)
    if (!LN) __builtin_amdgcn_sched_barrier(0);
    // Rows in PAIRS: x2[ch][component][pair] = (row 2p, row 2p + 1) -- one v_pk_fma_f32 per weight component and row pair, the weight
    // splat over both halves (the multiply-adds are what the vector ALU spends its time on: 4 x rows per 16 weight bytes).
    constexpr int NBP = (NB + 1) / 2;
    f32x2 x2[CH][4][NBP];
    // LayerNorm folded around the dot product (the algebra of the matrix-core plan, k_gemm's LN epilogue):
    //     LN(x) W^T + b = rstd (sum_k W[n][k] gamma[k] x[k] - mean c1[n]) + (b + W beta)[n],   c1[n] = sum_k W[n][k] gamma[k]
    // so the multiply-adds start on gamma * x the moment the rows land; the row statistics -- a reduction over the whole workgroup --
    // are only needed by the epilogue and no longer stand between the loads and the first multiply (before: statistics, a workgroup
    // barrier and a normalisation pass in front of every LN launch, ~0.8 us of its ~4 us fixed cost).
    if (LN) {
        float4 gm[CH];
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) gm[ch] = *((const float4*)(a.gamma + seg * SG_SEG + lane * 4) + ch * 64);
        // per-lane partial sums of its 12 raw elements per row in fp32 (error <= 12 ulp of the partial), across lanes and segments in fp64
        double st[2 * NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const float4 v = xr[ch][b];
                s += (v.x + v.y) + (v.z + v.w);
                ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            st[2 * b] = (double)s; st[2 * b + 1] = (double)ss;
        }
        if (2 * NB <= 16) {
            int sidx;
            const double tot = wave_reduce_many<2 * NB, double>(st, lane, &sidx);
            constexpr int LB = sg_log2p<2 * NB>();
            if ((lane & ((64 >> LB) - 1)) == 0 && sidx < 2 * NB) lnred[w * 2 * SG_MAX_ROWS + sidx] = tot;      // read behind the final barrier
        } else {            // 9 / 10 rows: sums and sums of squares in two butterflies of <= 16 values (registers)
            double sa[NB], sb[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { sa[b] = st[2 * b]; sb[b] = st[2 * b + 1]; }
            constexpr int LB = sg_log2p<NB>();
            int ia, ib;
            const double ta = wave_reduce_many<NB, double>(sa, lane, &ia);
            if ((lane & ((64 >> LB) - 1)) == 0 && ia < NB) lnred[w * 2 * SG_MAX_ROWS + 2 * ia] = ta;
            const double tb = wave_reduce_many<NB, double>(sb, lane, &ib);
            if ((lane & ((64 >> LB) - 1)) == 0 && ib < NB) lnred[w * 2 * SG_MAX_ROWS + 2 * ib + 1] = tb;
        }
#pragma unroll
        for (int ch = 0; ch < CH; ++ch)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                xr[ch][b].x *= gm[ch].x; xr[ch][b].y *= gm[ch].y; xr[ch][b].z *= gm[ch].z; xr[ch][b].w *= gm[ch].w;
            }
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
        for (int bp = 0; bp < NBP; ++bp) {
            const float4 v0 = xr[ch][2 * bp], v1 = 2 * bp + 1 < NB ? xr[ch][2 * bp + 1 < NB ? 2 * bp + 1 : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
            x2[ch][0][bp] = f32x2{v0.x, v1.x}; x2[ch][1][bp] = f32x2{v0.y, v1.y};
            x2[ch][2][bp] = f32x2{v0.z, v1.z}; x2[ch][3][bp] = f32x2{v0.w, v1.w};
        }

#define WMAR_SG_COMPUTE(BUF, GI)                                                               \
    {                                                                                          \
        f32x2 acc2[CG][NBP];                                                                   \
        _Pragma("unroll") for (int c = 0; c < CG; ++c)                                         \
            _Pragma("unroll") for (int bp = 0; bp < NBP; ++bp) acc2[c][bp] = f32x2{0.f, 0.f};  \
        _Pragma("unroll") for (int c = 0; c < CG; ++c)                                         \
            _Pragma("unroll") for (int ch = 0; ch < CH; ++ch) {                                \
                const float4 wv = BUF[c * CH + ch];                                            \
                const f32x2 wx = {wv.x, wv.x}, wy = {wv.y, wv.y}, wz = {wv.z, wv.z}, ww = {wv.w, wv.w}; \
                _Pragma("unroll") for (int bp = 0; bp < NBP; ++bp) {                           \
                    f32x2 s_ = acc2[c][bp];                                                    \
                    s_ = __builtin_elementwise_fma(wx, x2[ch][0][bp], s_); s_ = __builtin_elementwise_fma(wy, x2[ch][1][bp], s_); \
                    s_ = __builtin_elementwise_fma(wz, x2[ch][2][bp], s_); s_ = __builtin_elementwise_fma(ww, x2[ch][3][bp], s_); \
                    acc2[c][bp] = s_;                                                          \
                }                                                                              \
            }                                                                                  \
        float acc[R];                                                                          \
        _Pragma("unroll") for (int c = 0; c < CG; ++c)                                         \
            _Pragma("unroll") for (int b = 0; b < NB; ++b) acc[c * NB + b] = (b & 1) ? acc2[c][b >> 1].y : acc2[c][b >> 1].x; \
        if (R <= 32) {                                                                         \
            int ridx;                                                                          \
            const float tot_ = wave_reduce_many<R, float>(acc, lane, &ridx);                   \
            constexpr int LBR = sg_log2p<R>();                                                 \
            if ((lane & ((64 >> LBR) - 1)) == 0 && ridx < R) {                                 \
                const int c_ = ridx / NB, b_ = ridx - c_ * NB;                                 \
                part[((long long)seg * cols_wg + chalf * cols_wave + (GI) * CG + c_) * NB + b_] = tot_; \
            }                                                                                  \
        } else {       /* more than 32 values (the head at 9 / 10 rows): one butterfly per column */ \
            _Pragma("unroll") for (int c = 0; c < CG; ++c) {                                   \
                float col_[NB];                                                                \
                _Pragma("unroll") for (int b = 0; b < NB; ++b) col_[b] = acc[c * NB + b];      \
                int ridx;                                                                      \
                const float tot_ = wave_reduce_many<NB, float>(col_, lane, &ridx);             \
                constexpr int LBR = sg_log2p<NB>();                                            \
                if ((lane & ((64 >> LBR) - 1)) == 0 && ridx < NB)                              \
                    part[((long long)seg * cols_wg + chalf * cols_wave + (GI) * CG + c) * NB + ridx] = tot_; \
            }                                                                                  \
        }                                                                                      \
    }

    // column groups: NBUF groups in flight; group g + NBUF is requested into the buffer group g has just been multiplied out of
#pragma unroll
    for (int g = 0; g < G; ++g) {
        WMAR_SG_COMPUTE(wbuf[g % NBUF], g)
        if (g + NBUF < G) WMAR_SG_LOADW(wbuf[g % NBUF], g + NBUF)
    }
#undef WMAR_SG_LOADW
#undef WMAR_SG_COMPUTE
    __syncthreads();

    // segment sums in a fixed order + epilogue; column fastest so that the stores of a row are contiguous.  LN roles: the row
    // statistics the waves left in LDS (segment 0 = wave 0, segment 1 = wave NSPLIT) close the folded LayerNorm here.
    const double invK = inv_count_f64((double)a.K);
    if (ROLE == SG_HEAD) {
        for (int t = threadIdx.x; t < cols_wg * NB; t += NW * 64) {
            const int b = t / cols_wg, col = t - b * cols_wg;
            const int n = blockIdx.x * cols_wg + col;
            if (n >= a.N) continue;
            float s = part[(long long)col * NB + b];
#pragma unroll
            for (int sg = 1; sg < NSEG; ++sg) s += part[((long long)sg * cols_wg + col) * NB + b];
            const double mean = (lnred[2 * b] + lnred[NSPLIT * 2 * SG_MAX_ROWS + 2 * b]) * invK;
            const float rstd = rsqrtf((float)var_f64((lnred[2 * b + 1] + lnred[NSPLIT * 2 * SG_MAX_ROWS + 2 * b + 1]) * invK, mean) + 1e-5f);
            a.out[(long long)b * a.N + n] = rstd * (s - (float)mean * a.c1[n]) + (a.bias ? a.bias[n] : 0.f);
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int et = threadIdx.x + e * NW * 64;
            const int eb = et / cols_wg, ecol = et - eb * cols_wg;
            const int en = blockIdx.x * cols_wg + ecol;
            if (!(et < cols_wg * NB && en < a.N)) continue;
            float s = part[(long long)ecol * NB + eb];
#pragma unroll
            for (int sg = 1; sg < NSEG; ++sg) s += part[((long long)sg * cols_wg + ecol) * NB + eb];
            if (LN) {
                const double mean = (lnred[2 * eb] + lnred[NSPLIT * 2 * SG_MAX_ROWS + 2 * eb]) * invK;
                const float rstd = rsqrtf((float)var_f64((lnred[2 * eb + 1] + lnred[NSPLIT * 2 * SG_MAX_ROWS + 2 * eb + 1]) * invK, mean) + 1e-5f);
                s = rstd * (s - (float)mean * ec1[e]);
            }
            s += ebias[e];
            if (ROLE == SG_QKV) {
                const int pos = *a.pos_dev;
                const int which = en / a.D, j = en - which * a.D;
                if (which == 0) a.out[(long long)eb * a.D + j] = s;
                else {
                    const int h = j >> 6, d = j & 63;
                    float* cache = which == 1 ? a.kcache : a.vcache;
                    cache[(((long long)eb * a.H + h) * a.Tmax + pos) * 64 + d] = s;
                }
            } else if (ROLE == SG_PROJ || ROLE == SG_FC2) {
                a.out[(long long)eb * a.N + en] = eold[e] + s;
            } else {
                a.out[(long long)eb * a.N + en] = gelu_erf(s);
            }
        }
    }
}

template <int ROLE> constexpr int sg_nseg() { return ROLE == SG_FC2 ? 8 : 2; }
template <int ROLE> constexpr int sg_nw() { return ROLE == SG_FC2 ? 8 : 4; }

// columns per workgroup
template <int CG, int G, int ROLE>
constexpr int sg_cols_wg() { return (sg_nw<ROLE>() / sg_nseg<ROLE>()) * CG * G; }

template <int NB, int CG, int G, int ROLE>
static int launch_sgemv_nb(const SgArgs& a, hipStream_t st) {
    constexpr int cols_wg = sg_cols_wg<CG, G, ROLE>();
    const int grid = (a.N + cols_wg - 1) / cols_wg;
    const size_t lds = (size_t)sg_nw<ROLE>() * 2 * SG_MAX_ROWS * sizeof(double) + (size_t)sg_nseg<ROLE>() * cols_wg * NB * sizeof(float);
    hipLaunchKernelGGL((k_sgemv<NB, CG, G, ROLE>), dim3((unsigned)grid), dim3(sg_nw<ROLE>() * 64), lds, st, a);
    return launch_status("k_sgemv");
}

template <int CG, int G, int ROLE>
static int launch_sgemv(const SgArgs& a, int rows, hipStream_t st) {
    if (a.K != sg_nseg<ROLE>() * SG_SEG) { set_error("k_sgemv: K = %d is not %d segments of %d", a.K, sg_nseg<ROLE>(), SG_SEG); return WMAR_EINVAL; }
    switch (rows) {
        case 1: return launch_sgemv_nb<1, CG, G, ROLE>(a, st);
        case 2: return launch_sgemv_nb<2, CG, G, ROLE>(a, st);
        case 3: return launch_sgemv_nb<3, CG, G, ROLE>(a, st);
        case 4: return launch_sgemv_nb<4, CG, G, ROLE>(a, st);
        case 5: return launch_sgemv_nb<5, CG, G, ROLE>(a, st);
        case 6: return launch_sgemv_nb<6, CG, G, ROLE>(a, st);
        case 7: return launch_sgemv_nb<7, CG, G, ROLE>(a, st);
        case 8: return launch_sgemv_nb<8, CG, G, ROLE>(a, st);
        case 9: return launch_sgemv_nb<9, CG, G, ROLE>(a, st);
        case 10: return launch_sgemv_nb<10, CG, G, ROLE>(a, st);
        case 11: return launch_sgemv_nb<11, CG, G, ROLE>(a, st);
        case 12: return launch_sgemv_nb<12, CG, G, ROLE>(a, st);
        default: set_error("k_sgemv: %d rows (1..%d)", rows, SG_MAX_ROWS); return WMAR_EINVAL;
    }
}

// ------------------------------------------------------------------------------------------------- attention, head_dim 64
struct SaArgs {
    const float* q;        // [rows][D]
    const float* kcache;   // this layer's [Bmax][H][Tmax][64]; the new row is already appended (k_sgemv<QKV>)
    const float* vcache;
    float* y;              // [rows][D]
    const int* pos_dev;
    int H, Tmax, D;
    float scale;
};

// NU units of 4 cached rows per wave and trip, all requested before the first is used (a cache row is 16 lanes x 16 bytes).  Two
// passes without a branch: all NU scores first (independent 16-lane reductions the compiler interleaves), one maximum, NU independent
// exponentials, then the weighted V sum; the running (max, sum, V sum) is touched once per trip, not once per row.
template <int NU>
__device__ __forceinline__ void sattn_rows(const float4* __restrict__ Kp, const float4* __restrict__ Vp, int T, int u0, int rr,
                                           const float4 q4, float scale, float& m, float& l, float4& o) {
    float4 kb[NU], vb[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        int row = 4 * (u0 + 4 * i) + rr;
        row = row < T ? row : T - 1;
        kb[i] = ld_nt(Kp + (long long)row * 16);
        vb[i] = ld_nt(Vp + (long long)row * 16);
    }
    __builtin_amdgcn_sched_barrier(0);       // all 2 NU requests in flight before the first score (without it the scheduler interleaves
                                             // loads and arithmetic to save registers: 100 instead of 186 VGPRs, 6.0 -> 8.3 us per launch)
    float sc[NU];
    float mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        float s = fmaf(q4.w, kb[i].w, fmaf(q4.z, kb[i].z, fmaf(q4.y, kb[i].y, q4.x * kb[i].x)));
        s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
        s = (4 * (u0 + 4 * i) + rr < T) ? s * scale : -INFINITY;
        sc[i] = s;
        mt = fmaxf(mt, s);
    }
    const float mn = fmaxf(m, mt);
    const float ms = mn == -INFINITY ? 0.f : mn;        // a row group without a valid row so far: every weight below is exp(-inf) = 0
    const float corr = expf(m - ms);
    float lt = 0.f;
    float4 ot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const float pe = expf(sc[i] - ms);
        lt += pe;
        ot.x = fmaf(pe, vb[i].x, ot.x); ot.y = fmaf(pe, vb[i].y, ot.y); ot.z = fmaf(pe, vb[i].z, ot.z); ot.w = fmaf(pe, vb[i].w, ot.w);
    }
    l = fmaf(l, corr, lt);
    o.x = fmaf(o.x, corr, ot.x); o.y = fmaf(o.y, corr, ot.y); o.z = fmaf(o.z, corr, ot.z); o.w = fmaf(o.w, corr, ot.w);
    m = mn;
}

__global__ __launch_bounds__(256) void k_sattn(SaArgs a) {
    __shared__ float sm[16], sl[16];
    __shared__ __attribute__((aligned(16))) float so[16][64];
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, p = lane & 15, rr = lane >> 4;
    const int T = *a.pos_dev + 1;
    const float4 q4 = *((const float4*)(a.q + (long long)b * a.D + h * 64) + p);
    const long long base = ((long long)b * a.H + h) * a.Tmax * 64;
    const float4* Kp = (const float4*)(a.kcache + base) + p;
    const float4* Vp = (const float4*)(a.vcache + base) + p;
    float m = -INFINITY, l = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nu = (T + 3) >> 2;               // units of 4 rows; wave w takes units w, w + 4, ...
    if (nu <= 16) {
        if (w < nu) sattn_rows<4>(Kp, Vp, T, w, rr, q4, a.scale, m, l, o);
    } else {
        for (int u0 = w; u0 < nu; u0 += 64) sattn_rows<16>(Kp, Vp, T, u0, rr, q4, a.scale, m, l, o);
    }
    const int gi = w * 4 + rr;
    if (p == 0) { sm[gi] = m; sl[gi] = l; }
    *((float4*)&so[gi][p * 4]) = o;
    __syncthreads();
    if (threadIdx.x < 64) {
        float M = sm[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) M = fmaxf(M, sm[i]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = expf(sm[i] - M);            // groups without a row: exp(-inf) = 0
            L = fmaf(sl[i], e, L);
            O = fmaf(so[i][threadIdx.x], e, O);
        }
        a.y[(long long)b * a.D + h * 64 + threadIdx.x] = O / L;
    }
}

// ------------------------------------------------------------------------------------------------- token + position embedding
struct SeArgs {
    const float* tok_emb; const float* pos_emb; float* x;
    const long long* tok; long long tok_stride; int tok_use_pos;
    const int* pos_dev; int D;
};
__global__ __launch_bounds__(256) void k_sembed(SeArgs a) {
    const int b = blockIdx.y;
    const int k4 = blockIdx.x * 256 + threadIdx.x;
    if (k4 * 4 >= a.D) return;
    const int pos = *a.pos_dev;
    const long long tk = a.tok[(long long)b * a.tok_stride + (a.tok_use_pos ? pos : 0)];
    const float4 e = *((const float4*)(a.tok_emb + tk * a.D) + k4);
    const float4 pe = *((const float4*)(a.pos_emb + (long long)pos * a.D) + k4);
    *((float4*)(a.x + (long long)b * a.D) + k4) = make_float4(e.x + pe.x, e.y + pe.y, e.z + pe.z, e.w + pe.w);
}

#pragma clang fp contract(fast)

}  // namespace wmar
