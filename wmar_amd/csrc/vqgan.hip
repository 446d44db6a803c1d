// Taming VQGAN encoder / decoder / quantizer for gfx950 (fp32, exact-f32 MFMA).
//
// Reference: deps/taming/modules/diffusionmodules/model.py:30-193 (Normalize, Upsample,
// Downsample, ResnetBlock, AttnBlock), :343-538 (Encoder, Decoder);
// deps/taming/models/vqgan.py:64-73 (encode / decode); deps/taming/modules/vqvae/quantize.py:272-331.
//
// Data layout in HBM: every activation is NHWC fp32 ([B][H][W][C], channels padded to a
// multiple of 8), so a pixel's channel vector is one contiguous run and the 3x3 taps are
// plain row offsets.  Conv weights [Cout][Cin][kh][kw] are repacked once into
// MFMA-fragment order Wp[cout_tile][tap][cin_block][lane][4] (A operand of
// v_mfma_f32_32x32x2_f32: lane = (cout%32) + 32*((cin%8)/4)).
//
// conv3x3 / conv1x1 = implicit GEMM, D[cout][pixel] += W[cout][tap,cin] * X[pixel+tap][cin]:
// a workgroup owns an 8x8 pixel tile and up to 4 cout tiles; the input patch (tile + halo)
// for 32 input channels at a time is staged through LDS (coalesced rows in, 16-byte
// fragment reads out), each wave owns one cout tile and both 32-pixel halves.
// Nearest x2 upsampling (Upsample.forward) and the asymmetric-pad stride-2 downsample
// (Downsample.forward) are folded into the patch gather; bias and the residual add are
// folded into the epilogue.  GroupNorm(32, eps 1e-6)+swish is a statistics pass
// (fp64 partial sums, fixed order) plus an elementwise pass.
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "bx_split.h"

namespace wmar {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ------------------------------------------------------------------------ packing
// W [Cout][Cin][ks][ks] -> Wp[ct][tap][kb][lane] float4;  cout/cin padded with zeros.
__global__ void k_pack_conv(const float* __restrict__ W, float4* __restrict__ Wp, int Cout, int Cin, int ks,
                            int CT, int KBc) {
    const int T = ks * ks;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)CT * T * KBc * 64;
    if (idx >= total) return;
    int lane = (int)(idx & 63);
    long long r = idx >> 6;
    int kb = (int)(r % KBc); r /= KBc;
    int tap = (int)(r % T);
    int ct = (int)(r / T);
    int co = ct * 32 + (lane & 31);
    int ci0 = kb * 8 + 4 * (lane >> 5);
    float v[4];
    for (int i = 0; i < 4; ++i) {
        int ci = ci0 + i;
        v[i] = (co < Cout && ci < Cin) ? W[((long long)co * Cin + ci) * T + tap] : 0.f;
    }
    Wp[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

// W [Cout][Cin][ks][ks] -> Wq[ct][tap][cin/16][piece][lane] (16 bytes = 8 bf16): lane holds cout = 32 ct + lane % 32, input channels
// 16 k16 + 8 (lane / 32) + 0..7 -- the A operand of v_mfma_f32_32x32x16_bf16, one plane per bf16 piece of the weight (bx_split.h).
__global__ void k_pack_conv_bx(const float* __restrict__ W, u32x4* __restrict__ Wq, int Cout, int Cin, int ks, int CT, int KU,
                               long long w_bstride = 0, long long wq_bstride = 0, int transposed = 0) {
    // blockIdx.y = image for per-image "weights" (the attention's K and V^T, ks = 1): W + y * w_bstride floats, Wq + y * wq_bstride;
    // transposed: W is stored [Cin][Cout] (W[co][ci] = src[ci * Cout + co])
    const int T = ks * ks;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)CT * T * KU * 64) return;
    W += (long long)blockIdx.y * w_bstride;
    Wq += (long long)blockIdx.y * wq_bstride;
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int ku = (int)(r % KU); r /= KU;
    const int tap = (int)(r % T);
    const int ct = (int)(r / T);
    const int co = ct * 32 + (lane & 31), ci0 = ku * 16 + 8 * (lane >> 5);
    float v[8];
    for (int i = 0; i < 8; ++i)
        v[i] = (co < Cout && ci0 + i < Cin) ? (transposed ? W[(long long)(ci0 + i) * Cout + co] : W[((long long)co * Cin + ci0 + i) * T + tap]) : 0.f;
    unsigned h[4], m[4], l[4];
    for (int i = 0; i < 4; ++i) bx_split2(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
    u32x4* o = Wq + (idx >> 6) * 192 + lane;
    o[0] = u32x4{h[0], h[1], h[2], h[3]}; o[64] = u32x4{m[0], m[1], m[2], m[3]}; o[128] = u32x4{l[0], l[1], l[2], l[3]};
}

__global__ void k_pad_vec(const float* __restrict__ src, float* __restrict__ dst, int n, int npad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) dst[i] = i < n ? src[i] : 0.f;
}

// ------------------------------------------------------------------------ conv
struct ConvArgs {
    const float* in;     // NHWC [B][Hs][Ws][Cin]  (Hs,Ws = stored size; the conv sees 2x that when up=1)
    const float4* wp;
    const u32x4* wq;     // k_conv_bx: the weights as bf16 pieces (k_pack_conv_bx)
    long long wq_bstride;  // > 0: per-image weights (attention), image b uses wq + b * wq_bstride
    const float* bias;   // [CT*32]
    const float* res;    // nullable NHWC [B][Ho][Wo][Cout_s]
    float* out;          // NHWC [B][Ho][Wo][Cout_s]
    int Hs, Ws, Cin;     // Cin: stored (padded) input channels, multiple of 8
    int Ho, Wo, Cout_s;  // Cout_s: stored output channels (multiple of 8, >= real cout)
    int CT, KBc;         // cout tiles of 32; cin blocks of 8
    int ks, stride, up, pad;  // pad = leading pad (1 for 3x3 stride 1, 0 otherwise)
    int tiles_x, tiles_y;     // 8x8 output tiles per image
    int dbuf;                 // double-buffered patch staging (host: full 32-channel rounds, patch <= 4 float4 per thread)
    // GroupNorm(32, eps 1e-6) (+ swish) of the INPUT applied while the patch is staged (Normalize + nonlinearity in front of
    // every conv of a ResnetBlock / AttnBlock / the output conv, model.py:37-39, 104-120): the normalised tensor never exists in HBM
    const float2* gn_mr;      // nullable: (mean, rstd) per [image][group]
    const float* gn_g; const float* gn_b;   // per channel
    int gn_cpg, gn_swish;     // channels per group
    // GroupNorm statistics of the OUTPUT (the next Normalize's input): per (image, 8x8 tile, group) partial (sum, sum of squares) in
    // fp64, written by the epilogue so that no separate pass over the tensor is needed; k_gn_finalize_tiles adds the tiles in order
    double* st_part;          // nullable: [B][tiles][32][2]
    int st_cpg;               // output channels per group: 4, 8 or 16 (a group never straddles a 32-channel tile)
};

__device__ __forceinline__ float4 conv_gn(const ConvArgs& a, float4 v, int b, int c) {
    const float4 g = *(const float4*)(a.gn_g + c), bt = *(const float4*)(a.gn_b + c);
    float r[4] = {v.x, v.y, v.z, v.w};
    const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
    const float2* mr = a.gn_mr + b * 32;
    if (a.gn_cpg >= 4 && (a.gn_cpg & 3) == 0) {           // the four channels share a group
        const float2 m = mr[c / a.gn_cpg];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (r[i] - m.x) * m.y * gg[i] + bb[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float2 m = mr[(c + i) / a.gn_cpg]; r[i] = (r[i] - m.x) * m.y * gg[i] + bb[i]; }
    }
    if (a.gn_swish)
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = r[i] / (1.0f + __expf(-r[i]));
    return make_float4(r[0], r[1], r[2], r[3]);
}

// the same normalisation with the parameters already in registers (k_conv_bx requests them a whole round ahead)
__device__ __forceinline__ float4 conv_gn_apply(float4 v, const float4 g, const float4 bt, const float2 m, int swish) {
    float r[4] = {(v.x - m.x) * m.y * g.x + bt.x, (v.y - m.x) * m.y * g.y + bt.y, (v.z - m.x) * m.y * g.z + bt.z, (v.w - m.x) * m.y * g.w + bt.w};
    if (swish)
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = r[i] / (1.0f + __expf(-r[i]));
    return make_float4(r[0], r[1], r[2], r[3]);
}

// Epilogue of both conv kernels: bias (+ residual), NHWC store, per-tile GroupNorm partial sums of the output.
__device__ __forceinline__ void conv_store(const ConvArgs& a, const f32x16 (&acc)[2], int b, int ct, int ty, int tx, int lane) {
    const int j = lane & 31, half = lane >> 5;
    const int prow = j >> 3, pcol = j & 7;
    const int oy0 = ty * 8, ox0 = tx * 8;
    // epilogue: lane holds pixel j of each half-tile and couts ct*32 + 8g + 4*half + {0..3}
    double gs[4] = {0.0, 0.0, 0.0, 0.0}, gss[4] = {0.0, 0.0, 0.0, 0.0};   // this lane's sums per g over its 2 pixels x 4 channels
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int oy = oy0 + p * 4 + prow, ox = ox0 + pcol;
        const long long pix = ((long long)b * a.Ho + oy) * a.Wo + ox;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = ct * 32 + g * 8 + half * 4;
            if (co >= a.Cout_s) continue;
            const float4 bb = *(const float4*)(a.bias + co);
            float4 o = make_float4(acc[p][g * 4 + 0] + bb.x, acc[p][g * 4 + 1] + bb.y, acc[p][g * 4 + 2] + bb.z,
                                   acc[p][g * 4 + 3] + bb.w);
            if (a.res) {
                const float4 rr = *(const float4*)(a.res + pix * a.Cout_s + co);
                o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
            }
            *(float4*)(a.out + pix * a.Cout_s + co) = o;
            if (a.st_part) {
                gs[g] += (double)o.x + (double)o.y + (double)o.z + (double)o.w;
                gss[g] += sq4_f64(o);      // never a v_fmac_f64 chain: common.h
            }
        }
    }
    if (a.st_part) {
        // fold the 32 pixels of a lane half (always), the two halves (groups of >= 8 channels) and pairs of g (16 channels)
        const int top = a.st_cpg >= 8 ? 32 : 16;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            for (int o = 1; o <= top; o <<= 1) { gs[g] += __shfl_xor(gs[g], o); gss[g] += __shfl_xor(gss[g], o); }
        if (a.st_cpg == 16) { gs[0] += gs[1]; gss[0] += gss[1]; gs[2] += gs[3]; gss[2] += gss[3]; }
        const long long tile = ((long long)b * a.tiles_y + ty) * a.tiles_x + tx;
        double* dst = a.st_part + tile * 64;
        if (a.st_cpg == 4) {              // group = ct*8 + 2g + half: lanes 0 and 32 each write four groups
            if ((lane & 31) == 0)
#pragma unroll
                for (int g = 0; g < 4; ++g) { const int grp = ct * 8 + 2 * g + half; dst[grp * 2] = gs[g]; dst[grp * 2 + 1] = gss[g]; }
        } else if (lane == 0) {
            if (a.st_cpg == 8) {
#pragma unroll
                for (int g = 0; g < 4; ++g) { const int grp = ct * 4 + g; dst[grp * 2] = gs[g]; dst[grp * 2 + 1] = gss[g]; }
            } else {
                dst[(ct * 2) * 2] = gs[0]; dst[(ct * 2) * 2 + 1] = gss[0];
                dst[(ct * 2 + 1) * 2] = gs[2]; dst[(ct * 2 + 1) * 2 + 1] = gss[2];
            }
        }
    }
}

constexpr int CONV_CCH = 32;          // input channels staged per LDS round
constexpr int CONV_PSTRIDE = 36;      // floats per staged pixel (32 + 4 pad: spreads LDS banks)

// COT = cout tiles (= waves) per workgroup.
template <int COT>
__global__ __launch_bounds__(COT * 64) void k_conv(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float patch[];  // [PH*PW][CONV_PSTRIDE]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int cgroups = (a.CT + COT - 1) / COT;
    const int cg = bid % cgroups;
    const int b = bid / cgroups;
    const int ct = cg * COT + w;
    const bool active = ct < a.CT;
    const int T = a.ks * a.ks;
    const int PW = 7 * a.stride + a.ks, PH = PW;
    const int oy0 = ty * 8, ox0 = tx * 8;
    const int iy0 = oy0 * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;
    const int Hc = a.up ? a.Hs * 2 : a.Hs, Wc = a.up ? a.Ws * 2 : a.Ws;  // size the conv sees

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int j = lane & 31, half = lane >> 5;
    // pixel of lane j in pixel-tile p: row = p*4 + j/8, col = j%8
    const int prow = j >> 3, pcol = j & 7;
    const float* inb = a.in + (long long)b * a.Hs * a.Ws * a.Cin;

    // one 32-channel round: 9 taps x 4 k-blocks x 8 MFMAs per wave out of the staged patch
#define WMAR_CONV_ROUND(PATCH, C0, CCH)                                                                         \
    if (active) {                                                                                               \
        const int nkb = (CCH) >> 3;                                                                             \
        const float4* wbase = a.wp + ((long long)ct * T * a.KBc + ((C0) >> 3)) * 64 + lane;                     \
        for (int tap = 0; tap < T; ++tap) {                                                                     \
            const int dy = tap / a.ks, dx = tap - dy * a.ks;                                                    \
            const float4* wt = wbase + (long long)tap * a.KBc * 64;                                             \
            const float* p0 = (PATCH) + (((prow)*a.stride + dy) * PW + pcol * a.stride + dx) * CONV_PSTRIDE + half * 4; \
            const float* p1 = p0 + 4 * a.stride * PW * CONV_PSTRIDE;                                            \
            _Pragma("unroll 4") for (int kb = 0; kb < nkb; ++kb) {                                              \
                const float4 wv = wt[kb * 64];                                                                  \
                const float4 x0 = *(const float4*)(p0 + kb * 8);                                                \
                const float4 x1 = *(const float4*)(p1 + kb * 8);                                                \
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, x0.x, acc[0], 0, 0, 0);                     \
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, x1.x, acc[1], 0, 0, 0);                     \
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, x0.y, acc[0], 0, 0, 0);                     \
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, x1.y, acc[1], 0, 0, 0);                     \
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, x0.z, acc[0], 0, 0, 0);                     \
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, x1.z, acc[1], 0, 0, 0);                     \
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, x0.w, acc[0], 0, 0, 0);                     \
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, x1.w, acc[1], 0, 0, 0);                     \
            }                                                                                                   \
        }                                                                                                       \
    }
    // Double-buffered staging (the common case: full 32-channel rounds whose patch is at most 4 float4 per thread): the next
    // round's patch is fetched into registers while this round's MFMAs run and lands in the other LDS buffer afterwards --
    // one barrier per round and no exposed global-load latency.  a.dbuf is set by the host together with the doubled LDS size.
    if (a.dbuf) {
        constexpr int NPT = 4;
        const int nelem = PH * PW * 8;
        long long goff[NPT];
        int loff[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int e = threadIdx.x + i * COT * 64;
            goff[i] = -1; loff[i] = -1;
            if (e < nelem) {
                const int pix = e >> 3, qq = e & 7;
                const int py = pix / PW, px = pix - py * PW;
                int y = iy0 + py, x = ix0 + px;
                loff[i] = pix * CONV_PSTRIDE + qq * 4;
                if (y >= 0 && y < Hc && x >= 0 && x < Wc) {
                    if (a.up) { y >>= 1; x >>= 1; }
                    goff[i] = ((long long)y * a.Ws + x) * a.Cin + qq * 4;
                }
            }
        }
        const int psz = PH * PW * CONV_PSTRIDE;
        float4 pr[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) pr[i] = goff[i] >= 0 ? *(const float4*)(inb + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NPT; ++i)
            if (loff[i] >= 0) {
                if (a.gn_mr && goff[i] >= 0) pr[i] = conv_gn(a, pr[i], b, (threadIdx.x + i * COT * 64) % 8 * 4);
                *(float4*)(patch + loff[i]) = pr[i];
            }
        __syncthreads();
        int buf = 0;
        for (int c0 = 0; c0 < a.Cin; c0 += CONV_CCH) {
            const bool more = c0 + CONV_CCH < a.Cin;
            if (more) {
#pragma unroll
                for (int i = 0; i < NPT; ++i)
                    pr[i] = goff[i] >= 0 ? *(const float4*)(inb + goff[i] + c0 + CONV_CCH) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const float* cur = patch + buf * psz;
            WMAR_CONV_ROUND(cur, c0, CONV_CCH)
            if (more) {
                float* nxt = patch + (buf ^ 1) * psz;
#pragma unroll
                for (int i = 0; i < NPT; ++i)
                    if (loff[i] >= 0) {
                        if (a.gn_mr && goff[i] >= 0) pr[i] = conv_gn(a, pr[i], b, c0 + CONV_CCH + (threadIdx.x + i * COT * 64) % 8 * 4);
                        *(float4*)(nxt + loff[i]) = pr[i];
                    }
            }
            __syncthreads();
            buf ^= 1;
        }
    } else {
    for (int c0 = 0; c0 < a.Cin; c0 += CONV_CCH) {
        const int cch = min(CONV_CCH, a.Cin - c0);   // multiple of 8
        const int q4 = cch >> 2;                     // float4s per pixel this round
        __syncthreads();
        for (int e = threadIdx.x; e < PH * PW * q4; e += COT * 64) {
            const int pix = e / q4, qq = e - pix * q4;
            const int py = pix / PW, px = pix - py * PW;
            int y = iy0 + py, x = ix0 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < Hc && x >= 0 && x < Wc) {
                if (a.up) { y >>= 1; x >>= 1; }
                v = *(const float4*)(inb + ((long long)y * a.Ws + x) * a.Cin + c0 + qq * 4);
                if (a.gn_mr) v = conv_gn(a, v, b, c0 + qq * 4);
            }
            *(float4*)(patch + pix * CONV_PSTRIDE + qq * 4) = v;
        }
        __syncthreads();
        WMAR_CONV_ROUND(patch, c0, cch)
    }
    }
#undef WMAR_CONV_ROUND
    if (!active) return;
    conv_store(a, acc, b, ct, ty, tx, lane);
}

// ---- the same conv on the bf16 matrix pipe (bx_split.h): full 32-channel rounds, double-buffered staging.
// The patch is staged as the three bf16 pieces of every (normalised, swished) input value -- [pixel][piece][32 channels], 208 bytes
// per pixel (192 + 16 of padding: eight consecutive pixels' 16-byte reads cover all 32 LDS banks) -- and a wave reads one
// 16-byte B operand per (pixel tile, piece, 16 channels).  The weights arrive pre-split (k_pack_conv_bx).  Per (tap, 16 channels):
// 3 + 6 operand loads and 12 MFMAs of 32 cycles, against 16 MFMAs of 64 cycles in k_conv.
constexpr int CONV_PSTRIDE_BX = 208;   // bytes per staged pixel

// PT = pixel tiles of 32 per wave: 2 (an 8 x 8 output block) or 4 (round 5: two blocks side by side, 8 x 16) -- every weight operand
// then feeds 24 MFMAs instead of 12 (the weights stream from L2 at half the rate per MFMA) and a wave has four independent accumulators.
// STRIDE = 2 (round 5): the downsampling convolutions (3 x 3, stride 2, asymmetric zero pad (0, 1, 0, 1): model.py:69-73) on the same
// pipe -- a 17 x 17-pixel patch per 8 x 8 outputs (two buffers = 120 KB of LDS, one workgroup per CU), operands read two pixels apart.
template <int COT, int KS, int PT = 2, int STRIDE = 1>
__global__ __launch_bounds__(COT * 64) void k_conv_bx(ConvArgs a) {
    static_assert(PT == 2 || (PT == 4 && STRIDE == 1), "two pixel tiles per wave, or four at stride 1");
    extern __shared__ __attribute__((aligned(16))) unsigned char patchb[];  // [2][PH*PW][CONV_PSTRIDE_BX]
    constexpr int T = KS * KS, NIT = 2 * T;      // (tap, 16-channel half) steps per 32-channel round
    constexpr int PW = (PT == 4 ? 15 : 7) * STRIDE + KS, PH = 7 * STRIDE + KS;
    // register rings: weights WR - 1 steps ahead (L2), patch operands one (LDS); a step's slot is its index in the round modulo the
    // ring, so the ring length must divide the steps of a round for the prefetch across the round boundary to land in the right slot
    constexpr int WR = NIT % 3 == 0 ? 3 : 2, XR = 2;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = blockIdx.x;
    const int tiles_xw = PT == 4 ? a.tiles_x >> 1 : a.tiles_x;      // workgroup tiles per row (host: tiles_x even for PT = 4)
    const int tx = bid % tiles_xw; bid /= tiles_xw;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int cgroups = (a.CT + COT - 1) / COT;
    const int cg = bid % cgroups;
    const int b = bid / cgroups;
    const int ct = cg * COT + w;
    const bool active = ct < a.CT;
    const int iy0 = ty * 8 * STRIDE - a.pad, ix0 = tx * (PT == 4 ? 16 : 8) * STRIDE - a.pad;
    const int Hc = a.up ? a.Hs * 2 : a.Hs, Wc = a.up ? a.Ws * 2 : a.Ws;  // size the conv sees
    const int KU = a.Cin >> 4;

    f32x16 acc[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int j = lane & 31, half = lane >> 5;
    const int prow = j >> 3, pcol = j & 7;
    const float* inb = a.in + (long long)b * a.Hs * a.Ws * a.Cin;

    constexpr int nelem = PH * PW * 8;
    constexpr int NPT = (nelem + COT * 64 - 1) / (COT * 64);     // float4 per thread of a 32-channel patch (4 with 4 waves, 13 with 1)
    long long goff[NPT];
    int loff[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int e = threadIdx.x + i * COT * 64;
        goff[i] = -1; loff[i] = -1;
        if (e < nelem) {
            const int pix = e >> 3, qq = e & 7;
            const int py = pix / PW, px = pix - py * PW;
            int y = iy0 + py, x = ix0 + px;
            loff[i] = pix * CONV_PSTRIDE_BX + qq * 8;
            if (y >= 0 && y < Hc && x >= 0 && x < Wc) {
                if (a.up) { y >>= 1; x >>= 1; }
                goff[i] = ((long long)y * a.Ws + x) * a.Cin + qq * 4;
            }
        }
    }
    constexpr int psz = PH * PW * CONV_PSTRIDE_BX;
    float4 pr[NPT];
    // normalise (+ swish), split, store the three pieces of this thread's four channels (8 bytes each)
#define WMAR_CONVBX_STAGE(DST, C0)                                                                              \
    _Pragma("unroll") for (int i = 0; i < NPT; ++i)                                                             \
        if (loff[i] >= 0) {                                                                                     \
            if (goff[i] < 0) pr[i] = make_float4(0.f, 0.f, 0.f, 0.f);          /* zero padding (the load itself was unconditional) */ \
            else if (gn_fast) pr[i] = conv_gn_apply(pr[i], gnG, gnB, gnM, a.gn_swish);                          \
            else if (a.gn_mr) pr[i] = conv_gn(a, pr[i], b, (C0) + cthr);                                        \
            unsigned h0, m0, l0, h1, m1, l1;                                                                    \
            bx_split2(pr[i].x, pr[i].y, h0, m0, l0);                                                            \
            bx_split2(pr[i].z, pr[i].w, h1, m1, l1);                                                            \
            unsigned char* d = (DST) + loff[i];                                                                 \
            *(u32x2*)d = u32x2{h0, h1}; *(u32x2*)(d + 64) = u32x2{m0, m1}; *(u32x2*)(d + 128) = u32x2{l0, l1};  \
        }
#define WMAR_CONVBX_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)
    // step it of round r: tap = it / 2, channels 32 r + 16 (it % 2) .. + 15
    const u32x4* wtile = a.wq + (long long)b * a.wq_bstride + (long long)(active ? ct : 0) * T * KU * 192 + lane;
    u32x4 wr[WR][3], xr[XR][6];
#define WMAR_CONVBX_LOADW(SLOT, R, IT)                                                                          \
    { const u32x4* wp_ = wtile + ((long long)((IT) >> 1) * KU + 2 * (R) + ((IT) & 1)) * 192;                       \
      wr[SLOT][0] = wp_[0]; wr[SLOT][1] = wp_[64]; wr[SLOT][2] = wp_[128]; }
    const int lbase = (prow * STRIDE * PW + pcol * STRIDE) * CONV_PSTRIDE_BX + half * 16;
#define WMAR_CONVBX_LOADX(SLOT, CUR, IT)                                                                        \
    { const int dy_ = ((IT) >> 1) / KS, dx_ = ((IT) >> 1) % KS;                                                  \
      const unsigned char* p0_ = (CUR) + lbase + (dy_ * PW + dx_) * CONV_PSTRIDE_BX + ((IT) & 1) * 32;            \
      const unsigned char* p1_ = p0_ + 4 * STRIDE * PW * CONV_PSTRIDE_BX;                                                 \
      xr[SLOT][0] = *(const u32x4*)p0_; xr[SLOT][1] = *(const u32x4*)(p0_ + 64); xr[SLOT][2] = *(const u32x4*)(p0_ + 128); \
      xr[SLOT][3] = *(const u32x4*)p1_; xr[SLOT][4] = *(const u32x4*)(p1_ + 64); xr[SLOT][5] = *(const u32x4*)(p1_ + 128); \
 }
// PT = 4: the operands of ONE 8 x 8 block (HB = 0 left, 1 right) of step IT -- xr[0] / xr[1] hold the two blocks of the current step and
// are refilled for the next step right behind their twelve MFMAs (half-step ring: 48 operand registers instead of 96)
#define WMAR_CONVBX_LOADXH(SLOT, CUR, IT, HB)                                                                   \
    { const int dy_ = ((IT) >> 1) / KS, dx_ = ((IT) >> 1) % KS;                                                  \
      const unsigned char* p0_ = (CUR) + lbase + (dy_ * PW + dx_ + 8 * (HB)) * CONV_PSTRIDE_BX + ((IT) & 1) * 32; \
      const unsigned char* p1_ = p0_ + 4 * STRIDE * PW * CONV_PSTRIDE_BX;                                                 \
      xr[SLOT][0] = *(const u32x4*)p0_; xr[SLOT][1] = *(const u32x4*)(p0_ + 64); xr[SLOT][2] = *(const u32x4*)(p0_ + 128); \
      xr[SLOT][3] = *(const u32x4*)p1_; xr[SLOT][4] = *(const u32x4*)(p1_ + 64); xr[SLOT][5] = *(const u32x4*)(p1_ + 128); }
    // GroupNorm parameters of this thread's four channels (gamma, beta, the group's mean / rstd): they depend on the round only, so
    // they are requested a whole round ahead, in front of the patch -- inside the staging they were a dependent L2 round trip per
    // round with every wave of the workgroup waiting (and the wait drained the weight ring as well)
    const int cthr = (threadIdx.x % 8) * 4;
    const bool gn_fast = a.gn_mr && a.gn_cpg >= 4 && (a.gn_cpg & 3) == 0;
    float4 gnG = make_float4(1.f, 1.f, 1.f, 1.f), gnB = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 gnM = make_float2(0.f, 1.f);
#define WMAR_CONVBX_GN(C0)                                                                                      \
    if (gn_fast) { const int c_ = (C0) + cthr; gnG = *(const float4*)(a.gn_g + c_); gnB = *(const float4*)(a.gn_b + c_); gnM = a.gn_mr[b * 32 + c_ / a.gn_cpg]; }
    WMAR_CONVBX_GN(0)
#pragma unroll
    for (int i = 0; i < NPT; ++i) pr[i] = *(const float4*)(inb + (goff[i] >= 0 ? goff[i] : 0));
    WMAR_CONVBX_LOADW(0, 0, 0)
    if (WR > 2) { WMAR_CONVBX_LOADW(1, 0, 1) }
    WMAR_CONVBX_STAGE(patchb, 0)
    __syncthreads();
    int buf = 0;
    const int rounds = a.Cin / CONV_CCH;
    for (int r = 0; r < rounds; ++r) {
        const bool more = r + 1 < rounds;
        // the next round's patch and (at the end of this round) its first weights are requested UNCONDITIONALLY -- in the last round a
        // harmless re-read of the same round: a prefetch under a run-time branch makes hipcc's wait-count pass assume the smaller
        // outstanding count at the merge, and the last steps of every round then wait for the next round's loads as well
        const int rn = more ? r + 1 : r;
        WMAR_CONVBX_GN(rn * CONV_CCH)
#pragma unroll
        for (int i = 0; i < NPT; ++i) pr[i] = *(const float4*)(inb + (goff[i] >= 0 ? goff[i] : 0) + rn * CONV_CCH);   // unconditional; padding is zeroed when staged
        const unsigned char* cur = patchb + buf * psz;
#define WMAR_CONVBX_ROUND6(XS, A0, A1)                                                                          \
            WMAR_CONVBX_MFMA(wl, xr[XS][0], A0); WMAR_CONVBX_MFMA(wl, xr[XS][3], A1);                            \
            WMAR_CONVBX_MFMA(wh, xr[XS][2], A0); WMAR_CONVBX_MFMA(wh, xr[XS][5], A1);                            \
            WMAR_CONVBX_MFMA(wm, xr[XS][1], A0); WMAR_CONVBX_MFMA(wm, xr[XS][4], A1);                            \
            WMAR_CONVBX_MFMA(wm, xr[XS][0], A0); WMAR_CONVBX_MFMA(wm, xr[XS][3], A1);                            \
            WMAR_CONVBX_MFMA(wh, xr[XS][1], A0); WMAR_CONVBX_MFMA(wh, xr[XS][4], A1);                            \
            WMAR_CONVBX_MFMA(wh, xr[XS][0], A0); WMAR_CONVBX_MFMA(wh, xr[XS][3], A1);
        if (PT == 4) {
            WMAR_CONVBX_LOADXH(0, cur, 0, 0)
            WMAR_CONVBX_LOADXH(1, cur, 0, 1)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it + WR - 1 < NIT) { WMAR_CONVBX_LOADW((it + WR - 1) % WR, r, it + WR - 1) }
                else { WMAR_CONVBX_LOADW((it + WR - 1) % WR, rn, it + WR - 1 - NIT) }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 wh = wr[it % WR][0], wm = wr[it % WR][1], wl = wr[it % WR][2];
                WMAR_CONVBX_ROUND6(0, acc[0], acc[1])                       // the small products first
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < NIT) { WMAR_CONVBX_LOADXH(0, cur, it + 1, 0) }
                __builtin_amdgcn_sched_barrier(0);
                WMAR_CONVBX_ROUND6(1, acc[PT - 2], acc[PT - 1])
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < NIT) { WMAR_CONVBX_LOADXH(1, cur, it + 1, 1) }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        WMAR_CONVBX_LOADX(0, cur, 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            // weights of step it + WR - 1 (the next round's first steps at the end of this one), patch operands of step it + 1
            if (it + WR - 1 < NIT) { WMAR_CONVBX_LOADW((it + WR - 1) % WR, r, it + WR - 1) }
            else { WMAR_CONVBX_LOADW((it + WR - 1) % WR, rn, it + WR - 1 - NIT) }
            if (it + 1 < NIT) { WMAR_CONVBX_LOADX((it + 1) % XR, cur, it + 1) }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 wh = wr[it % WR][0], wm = wr[it % WR][1], wl = wr[it % WR][2];
            WMAR_CONVBX_ROUND6(it % XR, acc[0], acc[1])                     // the small products first
            __builtin_amdgcn_sched_barrier(0);
        }
        }
#undef WMAR_CONVBX_ROUND6
        if (more) {
            unsigned char* nxt = patchb + (buf ^ 1) * psz;
            WMAR_CONVBX_STAGE(nxt, (r + 1) * CONV_CCH)
        }
        __syncthreads();
        buf ^= 1;
    }
#undef WMAR_CONVBX_GN
#undef WMAR_CONVBX_STAGE
#undef WMAR_CONVBX_MFMA
#undef WMAR_CONVBX_LOADW
#undef WMAR_CONVBX_LOADX
#undef WMAR_CONVBX_LOADXH
    if (!active) return;
    if (PT == 4) {
        conv_store(a, *reinterpret_cast<const f32x16(*)[2]>(&acc[0]), b, ct, ty, 2 * tx, lane);
        conv_store(a, *reinterpret_cast<const f32x16(*)[2]>(&acc[PT - 2]), b, ct, ty, 2 * tx + 1, lane);
    } else {
        conv_store(a, *reinterpret_cast<const f32x16(*)[2]>(&acc[0]), b, ct, ty, tx, lane);
    }
}

// ---- 3 x 3, stride 1, at most 4 real output channels (the decoders' 128 -> 3 output conv at 256 x 256: 2.9 ms per 64 images on a
// 32-wide MFMA tile, where 29 of 32 output columns are padding -- 10 TF/s).  Round 5: fp32 FMAs on the vector ALU, one output
// pixel per lane, the weights of a round (wf[round][tap][32 channels][4 couts], k_pack_conv_few) through the scalar cache as
// SGPR operands; the patch staging (GroupNorm + swish applied on the way into LDS) is k_conv's.  16 x 16 output pixels per
// workgroup, 46 KB of LDS: three workgroups per CU cover each other's staging.
__global__ void k_pack_conv_few(const float* __restrict__ W, float* __restrict__ wf, int Cout, int Cin, int Cin_s) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // over (Cin_s / 32) * 9 * 32 * 4
    if (idx >= Cin_s * 9 * 4) return;
    const int co = idx & 3, ch = (idx >> 2) & 31, r = idx >> 7;
    const int tap = r % 9, c = (r / 9) * 32 + ch;
    wf[idx] = (co < Cout && c < Cin) ? W[((long long)co * Cin + c) * 9 + tap] : 0.f;
}

constexpr int CF_T = 16, CF_PW = CF_T + 2;
template <int NCO>
__global__ __launch_bounds__(256) void k_conv_few(ConvArgs a, const float* __restrict__ wf) {
    extern __shared__ __attribute__((aligned(16))) float patch[];  // [CF_PW * CF_PW][CONV_PSTRIDE]
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int px = threadIdx.x & (CF_T - 1), py = threadIdx.x >> 4;
    const int iy0 = ty * CF_T - 1, ix0 = tx * CF_T - 1;
    const float* inb = a.in + (long long)b * a.Hs * a.Ws * a.Cin;
    float acc[NCO];
#pragma unroll
    for (int i = 0; i < NCO; ++i) acc[i] = 0.f;
    for (int c0 = 0; c0 < a.Cin; c0 += CONV_CCH) {
        __syncthreads();                                   // the previous round's reads are done
        for (int e = threadIdx.x; e < CF_PW * CF_PW * 8; e += 256) {
            const int pix = e >> 3, qq = e & 7;
            const int yy = pix / CF_PW, xx = pix - yy * CF_PW;
            const int y = iy0 + yy, x = ix0 + xx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < a.Hs && x >= 0 && x < a.Ws) {
                v = *(const float4*)(inb + ((long long)y * a.Ws + x) * a.Cin + c0 + qq * 4);
                if (a.gn_mr) v = conv_gn(a, v, b, c0 + qq * 4);
            }
            *(float4*)(patch + pix * CONV_PSTRIDE + qq * 4) = v;
        }
        __syncthreads();
        const float* w = wf + (long long)(c0 / CONV_CCH) * (9 * CONV_CCH * 4);      // uniform: scalar loads
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const float* p = patch + ((py + dy) * CF_PW + px + dx) * CONV_PSTRIDE;
#pragma unroll
            for (int q = 0; q < CONV_CCH / 4; ++q) {
                const float4 xv = *(const float4*)(p + q * 4);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* wj = w + (tap * CONV_CCH + q * 4 + j) * 4;
#pragma unroll
                    for (int i = 0; i < NCO; ++i) acc[i] += xs[j] * wj[i];
                }
            }
        }
    }
    const long long pix = ((long long)b * a.Ho + ty * CF_T + py) * a.Wo + tx * CF_T + px;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NCO; ++i) o[i] = acc[i] + a.bias[i];
    float* dst = a.out + pix * a.Cout_s;
    *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
    for (int c = 4; c < a.Cout_s; c += 4) *(float4*)(dst + c) = make_float4(0.f, 0.f, 0.f, 0.f);     // padding channels (zero weights, zero bias)
}

// ------------------------------------------------------------------------ GroupNorm
// pass 1: per (image, pixel chunk) partial (sum, sumsq) of every group, fp64.
struct GnArgs {
    const float* x;   // NHWC [B][HW][C]
    float* y;
    double* partial;  // [B][nchunk][32][2]
    const float* gamma; const float* beta;
    int HW, C, nchunk, swish;
};

__global__ __launch_bounds__(256) void k_gn_partial(GnArgs a) {
    __shared__ double red[256][2];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int q4 = a.C >> 2;                 // float4 per pixel
    const int cpg = a.C / 32;                // channels per group (4, 8, 16 or 1/2 in tiny configs)
    const int p0 = (int)((long long)chunk * a.HW / a.nchunk), p1 = (int)((long long)(chunk + 1) * a.HW / a.nchunk);
    // thread t owns float4 column (t % q4) and walks pixels p0 + t/q4, + 256/q4 ...  (q4 divides 256 for C in {32..1024})
    const int col = threadIdx.x % q4, prow = threadIdx.x / q4, pstep = 256 / q4;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const float* xb = a.x + (long long)b * a.HW * a.C;
    if (prow < pstep)
        for (int p = p0 + prow; p < p1; p += pstep) {
            float4 v = *(const float4*)(xb + (long long)p * a.C + col * 4);
            s[0] += v.x; ss[0] += prod_f64(v.x, v.x);       // never a v_fmac_f64 chain: common.h
            s[1] += v.y; ss[1] += prod_f64(v.y, v.y);
            s[2] += v.z; ss[2] += prod_f64(v.z, v.z);
            s[3] += v.w; ss[3] += prod_f64(v.w, v.w);
        }
    // fold the 4 channels of this thread into their groups, then reduce threads in fixed order
    __shared__ double acc[32][2];
    if (threadIdx.x < 32) { acc[threadIdx.x][0] = 0; acc[threadIdx.x][1] = 0; }
    __syncthreads();
    // deterministic: one pass per pixel-row slot, serialised by barrier-free ownership:
    // write per-thread values to LDS, then 32 threads each sum their group's contributors in order.
    __shared__ double tmp[256][4][2];
    for (int i = 0; i < 4; ++i) { tmp[threadIdx.x][i][0] = s[i]; tmp[threadIdx.x][i][1] = ss[i]; }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int g = threadIdx.x;
        double ts = 0, tss = 0;
        for (int t = 0; t < 256; ++t) {
            if (t / q4 >= pstep) break;
            const int c = (t % q4) * 4;
            for (int i = 0; i < 4; ++i)
                if ((c + i) / cpg == g) { ts += tmp[t][i][0]; tss += tmp[t][i][1]; }
        }
        double* o = a.partial + (((long long)b * a.nchunk + chunk) * 32 + g) * 2;
        o[0] = ts; o[1] = tss;
    }
}

// pass 2: (mean, rstd) per (image, group) from the partial sums in fixed order; the normalisation itself happens in the
// consuming conv's patch loader (conv_gn)
__global__ void k_gn_finalize(GnArgs a, float2* mr) {
    const int b = blockIdx.x, g = threadIdx.x;
    double ts = 0, tss = 0;
    for (int c = 0; c < a.nchunk; ++c) {
        const double* p = a.partial + (((long long)b * a.nchunk + c) * 32 + g) * 2;
        ts += p[0]; tss += p[1];
    }
    const double n = (double)a.HW * (a.C / 32);
    const double rn = inv_count_f64(n);          // no fp64 division / sqrt expansions (chains of fused multiply-adds): common.h
    const double mean = ts * rn;
    const double var = var_f64(tss * rn, mean);
    mr[b * 32 + g] = make_float2((float)mean, (float)rsqrt_f64(var + 1e-6));
}

// (mean, rstd) from the per-tile partial sums a conv epilogue left behind: 8 thread groups add every 8th tile in order, then
// thread g adds the 8 partial results in order (fixed summation order, no atomics)
__global__ __launch_bounds__(256) void k_gn_finalize_tiles(const double* part, int tiles, double count, float2* mr) {
    __shared__ double red[8][32][2];
    const int b = blockIdx.x, g = threadIdx.x & 31, q = threadIdx.x >> 5;
    const double* p = part + (long long)b * tiles * 64 + g * 2;
    double ts = 0, tss = 0;
    for (int t = q; t < tiles; t += 8) { ts += p[(long long)t * 64]; tss += p[(long long)t * 64 + 1]; }
    red[q][g][0] = ts; red[q][g][1] = tss;
    __syncthreads();
    if (threadIdx.x < 32) {
        ts = 0; tss = 0;
        for (int i = 0; i < 8; ++i) { ts += red[i][g][0]; tss += red[i][g][1]; }
        const double rn = inv_count_f64(count);  // no fp64 division / sqrt expansions (chains of fused multiply-adds): common.h
        const double mean = ts * rn;
        const double var = var_f64(tss * rn, mean);
        mr[b * 32 + g] = make_float2((float)mean, (float)rsqrt_f64(var + 1e-6));
    }
}

// ------------------------------------------------------------------------ attention (single head)
// q,k,v: NHWC [B][N][C] (N = H*W tokens).  scores[b][i][j] = softmax_j( q_i . k_j * C^-0.5 );
// o[b][i][:] = sum_j scores[i][j] v[j][:]     (AttnBlock.forward, model.py:176-189)
__global__ __launch_bounds__(256) void k_attn_scores(const float* __restrict__ q, const float* __restrict__ k,
                                                     float* __restrict__ sc, int N, int C, float scale) {
    // one workgroup per (b, query i): 256 threads cover keys j
    extern __shared__ float qs[];  // [C]
    __shared__ float red[4];
    const int b = blockIdx.y, i = blockIdx.x;
    const float* qi = q + ((long long)b * N + i) * C;
    for (int c = threadIdx.x; c < C; c += 256) qs[c] = qi[c];
    __syncthreads();
    float* row = sc + ((long long)b * N + i) * N;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float* kj = k + ((long long)b * N + j) * C;
        float s = 0.f;
        for (int c = 0; c < C; c += 4) {
            float4 kv = *(const float4*)(kj + c);
            s += qs[c] * kv.x + qs[c + 1] * kv.y + qs[c + 2] * kv.z + qs[c + 3] * kv.w;
        }
        s *= scale;
        row[j] = s;
        mx = fmaxf(mx, s);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) {
        float e = __expf(row[j] - mx);
        row[j] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / sum;
    for (int j = threadIdx.x; j < N; j += 256) row[j] *= inv;
}

__global__ __launch_bounds__(256) void k_attn_pv(const float* __restrict__ sc, const float* __restrict__ v,
                                                 float* __restrict__ o, int N, int C) {
    // one workgroup per (b, query i); threads cover channels
    extern __shared__ float ps[];  // [N]
    const int b = blockIdx.y, i = blockIdx.x;
    const float* row = sc + ((long long)b * N + i) * N;
    for (int j = threadIdx.x; j < N; j += 256) ps[j] = row[j];
    __syncthreads();
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < N; ++j) {
            float4 vv = *(const float4*)(v + ((long long)b * N + j) * C + c);
            const float p = ps[j];
            acc.x += p * vv.x; acc.y += p * vv.y; acc.z += p * vv.z; acc.w += p * vv.w;
        }
        *(float4*)(o + ((long long)b * N + i) * C + c) = acc;
    }
}

// softmax over the rows of sc [rows][N] after scaling, in place: one wave per row (the MFMA attention path below)
__global__ __launch_bounds__(256) void k_attn_softmax(float* __restrict__ sc, long long rows, int N, float scale) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = sc + row * N;
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) { const float s = p[j] * scale; p[j] = s; mx = fmaxf(mx, s); }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) { const float e = __expf(p[j] - mx); p[j] = e; sum += e; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int j = lane; j < N; j += 64) p[j] *= inv;
}

// ------------------------------------------------------------------------ quantizer
__global__ void k_codebook_gather(const long long* __restrict__ codes, const float* __restrict__ emb,
                                  float* __restrict__ z, long long npix, int E, int n_embed) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over npix * E/4
    const int q4 = E >> 2;
    if (idx >= npix * q4) return;
    long long p = idx / q4;
    int c = (int)(idx % q4) * 4;
    long long code = codes[p];
    if (code < 0) code = 0;
    if (code >= n_embed) code = n_embed - 1;
    *(float4*)(z + p * E + c) = *(const float4*)(emb + code * E + c);
}

__global__ __launch_bounds__(64) void k_row_sqnorm(const float* __restrict__ x, float* __restrict__ out, int E) {
    // out[r] = sum_c x[r][c]^2, sequential-pairwise like a plain fp32 reduction
    const long long r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < E; c += 64) { float v = x[r * E + c]; s += v * v; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[r] = s;
}

// argmin_n ( |z|^2 + |e_n|^2 - 2 z.e_n ), first minimum (VectorQuantizer2.forward, quantize.py:277-285).
// One workgroup = 64 pixels (2 MFMA column tiles) x all codes; z rows live in LDS for the whole
// sweep; each of the 4 waves takes every 4th 32-code tile and keeps a running (min, index).
struct VqArgs {
    const float* z;       // [P][E]
    const float4* ep;     // packed codebook [NT][KB][64]
    const float* enorm;   // [n_embed]
    const float* znorm;   // [P]
    long long* codes;     // [P]
    long long P;
    int E, n_embed;
};

__global__ __launch_bounds__(256) void k_vq_argmin(VqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float zs[];  // [64][E+4]
    __shared__ float bestv[4][64];
    __shared__ int besti[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * 64;
    const int ZS = a.E + 4;
    for (int e = threadIdx.x; e < 64 * (a.E >> 2); e += 256) {
        const int r = e / (a.E >> 2), c = (e % (a.E >> 2)) * 4;
        float4 v = (p0 + r < a.P) ? *(const float4*)(a.z + (p0 + r) * a.E + c) : make_float4(0, 0, 0, 0);
        *(float4*)(zs + r * ZS + c) = v;
    }
    __syncthreads();
    const int j = lane & 31, half = lane >> 5;
    const int KB = a.E >> 3, NT = a.n_embed >> 5;
    float zn[2];
    zn[0] = (p0 + j < a.P) ? a.znorm[p0 + j] : 0.f;
    zn[1] = (p0 + 32 + j < a.P) ? a.znorm[p0 + 32 + j] : 0.f;
    float bv[2] = {INFINITY, INFINITY};
    int bi[2] = {0, 0};
    for (int nt = w; nt < NT; nt += 4) {
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float4* wt = a.ep + (long long)nt * KB * 64 + lane;
#pragma unroll 4
        for (int kb = 0; kb < KB; ++kb) {
            const float4 wv = wt[kb * 64];
            const float4 x0 = *(const float4*)(zs + j * ZS + kb * 8 + half * 4);
            const float4 x1 = *(const float4*)(zs + (32 + j) * ZS + kb * 8 + half * 4);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, x0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, x1.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, x0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, x1.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, x0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, x1.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, x0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, x1.w, acc[1], 0, 0, 0);
        }
        // lane holds pixel j (of each half) and codes nt*32 + (r&3) + 8*(r>>2) + 4*half, ascending in r
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float d = (zn[p] + a.enorm[n]) - 2.0f * acc[p][r];
                if (d < bv[p] || (d == bv[p] && n < bi[p])) { bv[p] = d; bi[p] = n; }
            }
    }
    // combine the two lane halves (same pixel, interleaved code sets), then the 4 waves
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float ov = __shfl_xor(bv[p], 32);
        int oi = __shfl_xor(bi[p], 32);
        if (ov < bv[p] || (ov == bv[p] && oi < bi[p])) { bv[p] = ov; bi[p] = oi; }
        if (half == 0) { bestv[w][p * 32 + j] = bv[p]; besti[w][p * 32 + j] = bi[p]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = bestv[0][threadIdx.x];
        int i = besti[0][threadIdx.x];
        for (int ww = 1; ww < 4; ++ww) {
            float ov = bestv[ww][threadIdx.x];
            int oi = besti[ww][threadIdx.x];
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if (p0 + threadIdx.x < a.P) a.codes[p0 + threadIdx.x] = i;
    }
}

// Round 5: the same search with 32 * PT pixels per workgroup (PT = 4: every 1-KiB codebook operand feeds 16 MFMAs instead of 8 and
// the 16 MB codebook is streamed by half as many workgroups) and the CODES split over CS workgroups per pixel group, so that
// (pixels / 128) * CS fills the chip; the partial winners meet in a packed (ordered distance, index) word with atomicMin --
// order-independent, and the smaller index wins a tie exactly as in the sequential rule above.  The codebook operands run through a
// two-stage register ring across tile boundaries (in k_vq_argmin every k-block waited for its own L2 / MALL round trip: 34 % of
// the fp32-MFMA rate).  Same MFMA sequence per (pixel, code), so the distances -- and the codes -- are bit-identical.
__device__ __forceinline__ unsigned long long vq_pack(float d, int n) {
    d = d + 0.0f;                                          // -0 -> +0: equal distances must compare equal
    uint32_t u = __float_as_uint(d);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;            // monotone in d
    return ((unsigned long long)u << 32) | (uint32_t)n;
}

template <int PT>
__global__ __launch_bounds__(256) void k_vq_argmin_split(VqArgs a, unsigned long long* __restrict__ best, int CS) {
    extern __shared__ __attribute__((aligned(16))) float zs[];  // [32 * PT][E + 4]
    __shared__ unsigned long long wbest[4][32 * PT];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = blockIdx.x / CS, cs = blockIdx.x - grp * CS;
    const long long p0 = (long long)grp * 32 * PT;
    const int ZS = a.E + 4;
    for (int e = threadIdx.x; e < 32 * PT * (a.E >> 2); e += 256) {
        const int r = e / (a.E >> 2), c = (e % (a.E >> 2)) * 4;
        const float4 v = (p0 + r < a.P) ? *(const float4*)(a.z + (p0 + r) * a.E + c) : make_float4(0, 0, 0, 0);
        *(float4*)(zs + r * ZS + c) = v;
    }
    __syncthreads();
    const int j = lane & 31, half = lane >> 5;
    const int KB = a.E >> 3, NT = a.n_embed >> 5;
    const int tps = NT / CS;                              // host: NT % CS == 0, tps >= 4
    float zn[PT];
    unsigned long long bk[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) { zn[i] = (p0 + 32 * i + j < a.P) ? a.znorm[p0 + 32 * i + j] : 0.f; bk[i] = ~0ull; }
    const float* zrow = zs + j * ZS + half * 4;
    constexpr int U = 4;
    const int nst = KB / U;                               // host: KB % (2 * U) == 0
    float4 wA[U], wB[U];
#define WMAR_VQ_LOAD(WB, NTI, SI)                                                                  \
    {                                                                                               \
        const float4* wt_ = a.ep + ((long long)(NTI) * KB + (SI) * U) * 64 + lane;                  \
        _Pragma("unroll") for (int u = 0; u < U; ++u) WB[u] = wt_[u * 64];                          \
    }
#define WMAR_VQ_MMA(WB, SI)                                                                        \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
        float4 x[PT];                                                                               \
        _Pragma("unroll") for (int i = 0; i < PT; ++i) x[i] = *(const float4*)(zrow + 32 * i * ZS + ((SI) * U + u) * 8); \
        _Pragma("unroll") for (int i = 0; i < PT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[u].x, x[i].x, acc[i], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < PT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[u].y, x[i].y, acc[i], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < PT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[u].z, x[i].z, acc[i], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < PT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[u].w, x[i].w, acc[i], 0, 0, 0); \
    }
    const int nt_first = cs * tps + w, nt_end = (cs + 1) * tps;
    WMAR_VQ_LOAD(wA, nt_first, 0)
    for (int nt = nt_first; nt < nt_end; nt += 4) {
        const bool last_tile = nt + 4 >= nt_end;
        float4 en[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) en[g] = *(const float4*)(a.enorm + nt * 32 + 8 * g + 4 * half);
        f32x16 acc[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int st = 0; st < nst; st += 2) {
            WMAR_VQ_LOAD(wB, nt, st + 1)
            __builtin_amdgcn_sched_barrier(0);
            WMAR_VQ_MMA(wA, st)
            __builtin_amdgcn_sched_barrier(0);
            // the stage after next: the same tile, or the first stage of this wave's next tile (the very last request re-reads a
            // valid block: no load sits under a branch)
            const bool wrap = st + 2 >= nst;
            WMAR_VQ_LOAD(wA, wrap ? (last_tile ? nt : nt + 4) : nt, wrap ? 0 : st + 2)
            __builtin_amdgcn_sched_barrier(0);
            WMAR_VQ_MMA(wB, st + 1)
            __builtin_amdgcn_sched_barrier(0);
        }
        // lane holds pixel j of each tile and codes nt*32 + (r&3) + 8*(r>>2) + 4*half
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float e_ = (r & 3) == 0 ? en[r >> 2].x : ((r & 3) == 1 ? en[r >> 2].y : ((r & 3) == 2 ? en[r >> 2].z : en[r >> 2].w));
                const float d = (zn[i] + e_) - 2.0f * acc[i][r];
                const unsigned long long k = vq_pack(d, n);
                bk[i] = k < bk[i] ? k : bk[i];
            }
    }
#undef WMAR_VQ_LOAD
#undef WMAR_VQ_MMA
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const unsigned long long o = __shfl_xor(bk[i], 32);
        bk[i] = o < bk[i] ? o : bk[i];
        if (half == 0) wbest[w][i * 32 + j] = bk[i];
    }
    __syncthreads();
    if (threadIdx.x < 32 * PT) {
        unsigned long long v = wbest[0][threadIdx.x];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) { const unsigned long long o = wbest[ww][threadIdx.x]; v = o < v ? o : v; }
        if (p0 + threadIdx.x < a.P) atomicMin(best + p0 + threadIdx.x, v);
    }
}

__global__ void k_vq_finish(const unsigned long long* __restrict__ best, long long* __restrict__ codes, long long P) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) codes[p] = (long long)(best[p] & 0xffffffffull);
}

// ------------------------------------------------------------------------ layout conversions
// NCHW image -> NHWC with channels padded to Cs
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, int Cs) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW (grid.y = unused)
    const long long b = blockIdx.y;
    if (idx >= HW) return;
    for (int c = 0; c < Cs; ++c)
        dst[((long long)b * HW + idx) * Cs + c] = c < C ? src[((long long)b * C + c) * HW + idx] : 0.f;
}

// NHWC (stored Cs channels) -> NCHW first C channels, clamped to [-1, 1] (taming_wrapper.py:82)
__global__ void k_nhwc_to_nchw_clamp(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, int Cs) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = blockIdx.y;
    if (idx >= HW) return;
    for (int c = 0; c < C; ++c) {
        float v = src[((long long)b * HW + idx) * Cs + c];
        dst[((long long)b * C + c) * HW + idx] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}

}  // namespace wmar

using namespace wmar;

// ------------------------------------------------------------------------------ engine
struct ConvW {
    float4* wp = nullptr;
    u32x4* wq = nullptr;     // bf16 pieces for k_conv_bx (input channels a multiple of 32)
    float* wf = nullptr;     // k_conv_few: 3 x 3 convs with at most 4 output channels
    float* bias = nullptr;
    int cin = 0, cout = 0, cin_s = 0, cout_s = 0, ks = 1, CT = 0, KBc = 0;
};
struct NormW { float* g = nullptr; float* b = nullptr; int C = 0; };
struct ResW { NormW n1, n2; ConvW c1, c2, nin; bool has_nin = false; };
struct AttnW { NormW n; ConvW q, k, v, proj; };

struct wmar_vq {
    wmar_vq_config cfg{};
    std::vector<void*> allocs;
    int64_t bytes = 0;
    int Bmax = 0, S = 0;
    // decoder
    ConvW post_quant, d_conv_in, d_conv_out;
    ResW d_mid1, d_mid2; AttnW d_midattn;
    std::vector<std::vector<ResW>> d_up;        // [level][block]
    std::vector<std::vector<AttnW>> d_upattn;   // [level][block]
    std::vector<ConvW> d_upsample;              // [level]
    NormW d_norm_out;
    // encoder
    ConvW quant, e_conv_in, e_conv_out;
    ResW e_mid1, e_mid2; AttnW e_midattn;
    std::vector<std::vector<ResW>> e_down;
    std::vector<std::vector<AttnW>> e_downattn;
    std::vector<ConvW> e_downsample;
    NormW e_norm_out;
    // quantizer
    float* emb = nullptr; float4* emb_p = nullptr; float* enorm = nullptr;
    // workspaces
    float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t buf_elems = 0;
    float *aq = nullptr, *ak = nullptr, *av = nullptr, *ao = nullptr, *asc = nullptr;
    u32x4 *attk = nullptr, *attv = nullptr;   // per-image K and V^T as bf16-piece conv weights (MFMA attention)
    float* zbias = nullptr;                    // zeros, max(tokens, channels) long
    double* gn_partial = nullptr;
    double* gn_tiles = nullptr; long long gn_tiles_cap = 0;   // per-tile GroupNorm partial sums written by conv epilogues
    float* znorm = nullptr;
    unsigned long long* vqbest = nullptr;      // packed (distance, code) winners of k_vq_argmin_split

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
    ~wmar_vq() { for (void* p : allocs) (void)hipFree(p); }
};

namespace {

constexpr int GN_CHUNKS_MAX = 64;
constexpr int GN_MR_DOUBLES = 32768;   // head of the GroupNorm scratch: (mean, rstd) float2 per [image][32 groups], up to 1024 images
inline int pad8(int c) { return (c + 7) & ~7; }

struct ArenaRef {   // the engine that owns the allocations
    std::vector<void*>* allocs;
    int64_t* bytes;
    template <typename T>
    int alloc(T** p, size_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
        if (e != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
            return WMAR_ENOMEM;
        }
        allocs->push_back(q);
        *bytes += (int64_t)(n * sizeof(T));
        *p = (T*)q;
        return WMAR_OK;
    }
};

struct Loader {
    std::map<std::string, const void*> m;
    ArenaRef* v;
    hipStream_t st;
    int rc = WMAR_OK;
    const float* need(const std::string& k) {
        auto it = m.find(k);
        if (it == m.end()) {
            if (rc == WMAR_OK) { set_error("checkpoint tensor '%s' is missing", k.c_str()); rc = WMAR_EMISSING; }
            return nullptr;
        }
        return (const float*)it->second;
    }
    // has_bias = false: bias-free conv (MaskGIT-VQGAN ResnetBlock / encoder conv_in) -> zero bias
    void conv(const std::string& p, int cin, int cout, int ks, ConvW& c, bool has_bias = true) {
        const float* W = need(p + ".weight");
        const float* bsrc = has_bias ? need(p + ".bias") : nullptr;
        if (rc) return;
        c.cin = cin; c.cout = cout; c.ks = ks; c.cin_s = pad8(cin); c.cout_s = pad8(cout);
        c.CT = (cout + 31) / 32; c.KBc = c.cin_s / 8;
        size_t n = (size_t)c.CT * ks * ks * c.KBc * 64;
        if ((rc = v->alloc(&c.wp, n))) return;
        hipLaunchKernelGGL(k_pack_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, c.wp, cout, cin, ks, c.CT, c.KBc);
        if ((rc = v->alloc(&c.bias, (size_t)c.CT * 32))) return;
        if (bsrc) hipLaunchKernelGGL(k_pad_vec, dim3((c.CT * 32 + 255) / 256), dim3(256), 0, st, bsrc, c.bias, cout, c.CT * 32);
        else if (hipMemsetAsync(c.bias, 0, (size_t)c.CT * 32 * 4, st) != hipSuccess) { set_error("bias memset failed"); rc = WMAR_EHIP; return; }
        rc = launch_status("k_pack_conv");
        if (rc == WMAR_OK && c.cin_s % CONV_CCH == 0) {
            const size_t nq = (size_t)c.CT * ks * ks * (c.cin_s / 16) * 64;
            if ((rc = v->alloc(&c.wq, nq * 3))) return;
            hipLaunchKernelGGL(k_pack_conv_bx, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, W, c.wq, cout, cin, ks, c.CT, c.cin_s / 16);
            rc = launch_status("k_pack_conv_bx");
        }
        if (rc == WMAR_OK && c.cin_s % CONV_CCH == 0 && ks == 3 && cout <= 4) {
            const size_t nf = (size_t)c.cin_s * 9 * 4;
            if ((rc = v->alloc(&c.wf, nf))) return;
            hipLaunchKernelGGL(k_pack_conv_few, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, W, c.wf, cout, cin, c.cin_s);
            rc = launch_status("k_pack_conv_few");
        }
    }
    void norm(const std::string& p, int C, NormW& n) {
        const float* g = need(p + ".weight");
        const float* b = need(p + ".bias");
        if (rc) return;
        n.C = C;
        if ((rc = v->alloc(&n.g, (size_t)C))) return;
        if ((rc = v->alloc(&n.b, (size_t)C))) return;
        if (hipMemcpyAsync(n.g, g, C * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(n.b, b, C * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("norm copy failed"); rc = WMAR_EHIP;
        }
    }
    void res(const std::string& p, int cin, int cout, ResW& r) {
        norm(p + "norm1", cin, r.n1);
        conv(p + "conv1", cin, cout, 3, r.c1);
        norm(p + "norm2", cout, r.n2);
        conv(p + "conv2", cout, cout, 3, r.c2);
        r.has_nin = cin != cout;
        if (r.has_nin) conv(p + "nin_shortcut", cin, cout, 1, r.nin);
    }
    void attn(const std::string& p, int c, AttnW& a) {
        norm(p + "norm", c, a.n);
        conv(p + "q", c, c, 1, a.q);
        conv(p + "k", c, c, 1, a.k);
        conv(p + "v", c, c, 1, a.v);
        conv(p + "proj_out", c, c, 1, a.proj);
    }
};

bool in_attn_res(const wmar_vq_config& c, int res) {
    for (int i = 0; i < c.n_attn_res; ++i)
        if (c.attn_resolutions[i] == res) return true;
    return false;
}

// WMAR_CONV_NO_BX=1 (read once, any build): keep every convolution on the fp32-input MFMA.  The bf16-piece split turns an infinite
// operand into NaN (inf - inf) where fp32 arithmetic gives +-inf (bx_split.h): a model with non-finite activations can opt out.
static bool conv_s2_bx() { static int v = -1; if (v < 0) { const char* e = getenv("WMAR_CONV_S2_FP32"); v = e ? 0 : 1; } return v != 0; }      // A/B: WMAR_CONV_S2_FP32=1 keeps k_conv
static bool conv_pt4() { static int v = -1; if (v < 0) { const char* e = getenv("WMAR_CONV_PT"); v = (e && atoi(e) == 2) ? 0 : 1; } return v != 0; }
static bool conv_no_bx() { static int v = -1; if (v < 0) v = getenv("WMAR_CONV_NO_BX") ? 1 : 0; return v != 0; }
#ifdef WMAR_DEV_KNOBS
static bool vq_trace() { static int v = -1; if (v < 0) { const char* e = getenv("WMAR_VQ_TRACE"); v = e ? atoi(e) : 0; } return v != 0; }
#else
static constexpr bool vq_trace() { return false; }
#endif

// Which tensor the per-tile statistics buffer currently describes.  Set by the conv that produced the tensor, consumed by the
// next run_gn on it; one call (decode / encode) at a time per thread.
struct GnTrack {
    double* part = nullptr; long long cap = 0;      // [B][tiles][32][2] doubles, capacity in doubles
    const float* src = nullptr; int tiles = 0, C = 0;
};
static thread_local GnTrack g_trk;

// a GroupNorm whose statistics are ready and whose normalisation is applied by the convs that consume it
struct GnRef {
    const float2* mr; const float* g; const float* b; int C, swish;
};

int run_conv(const ConvW& c, const float* in, float* out, const float* res, int B, int Hs, int Ws, int stride, int up,
             hipStream_t st, const GnRef* gn = nullptr, long long wq_bstride = 0) {
    ConvArgs a{};
    if (gn) {
        WMAR_REQUIRE(gn->C == c.cin_s && !up, "fused GroupNorm: channel count %d != conv input %d", gn->C, c.cin_s);
        a.gn_mr = gn->mr; a.gn_g = gn->g; a.gn_b = gn->b; a.gn_cpg = gn->C / 32; a.gn_swish = gn->swish;
    }
    a.in = in; a.wp = c.wp; a.wq = c.wq; a.wq_bstride = wq_bstride; a.bias = c.bias; a.res = res; a.out = out;
    a.Hs = Hs; a.Ws = Ws; a.Cin = c.cin_s;
    const int Hc = up ? 2 * Hs : Hs, Wc = up ? 2 * Ws : Ws;
    a.Ho = stride == 2 ? Hc / 2 : Hc; a.Wo = stride == 2 ? Wc / 2 : Wc;
    a.Cout_s = c.cout_s; a.CT = c.CT; a.KBc = c.KBc; a.ks = c.ks; a.stride = stride; a.up = up;
    a.pad = (c.ks == 3 && stride == 1) ? 1 : 0;
    WMAR_REQUIRE(a.Ho % 8 == 0 && a.Wo % 8 == 0, "conv output %dx%d is not a multiple of the 8x8 tile", a.Ho, a.Wo);
    a.tiles_x = a.Wo / 8; a.tiles_y = a.Ho / 8;
    const int PW = 7 * stride + c.ks;
    const int COT = c.CT >= 4 ? 4 : (c.CT >= 2 ? 2 : 1);
    a.dbuf = (c.cin_s % CONV_CCH == 0 && PW * PW * 8 <= 4 * COT * 64) ? 1 : 0;
    // statistics of the output for the GroupNorm that (usually) follows: groups of 4 / 8 / 16 channels inside one 32-channel tile
    const int cpg_out = c.cout / 32;
    const bool stats = wq_bstride == 0 && g_trk.part && c.cout == c.cout_s && c.cout % 32 == 0 && (cpg_out == 4 || cpg_out == 8 || cpg_out == 16) &&
                       (long long)B * a.tiles_x * a.tiles_y * 64 <= g_trk.cap;
    if (stats) { a.st_part = g_trk.part; a.st_cpg = cpg_out; g_trk.src = out; g_trk.tiles = a.tiles_x * a.tiles_y; g_trk.C = c.cout; }
    else if (g_trk.src == out) g_trk.src = nullptr;      // the tensor the buffer described is being overwritten
    // k_conv_bx stages (PW * PW * 8) / threads float4 per thread: up to 13 (one wave per workgroup: the 128 -> 3 output conv)
    const bool bx2 = c.wq && c.cin_s % CONV_CCH == 0 && stride == 2 && c.ks == 3 && COT == 4 && !up && !conv_no_bx() && conv_s2_bx();   // downsampling convs
    const bool bx = (c.wq && c.cin_s % CONV_CCH == 0 && stride == 1 && (c.ks == 3 || c.ks == 1) && !conv_no_bx()) || bx2;
    if (bx) a.dbuf = 1;
    // 8 x 16 output pixels per workgroup (four pixel tiles per wave) for the wide 3 x 3 convolutions on the bf16 pipe: WMAR_CONV_PT=2 keeps 8 x 8 (A/B)
    const bool wide = bx && !bx2 && COT == 4 && c.ks == 3 && a.tiles_x % 2 == 0 && wq_bstride == 0 && conv_pt4();
    const size_t lds = bx ? (size_t)2 * PW * (wide ? PW + 8 : PW) * CONV_PSTRIDE_BX : (size_t)(a.dbuf ? 2 : 1) * PW * PW * CONV_PSTRIDE * sizeof(float);
    const int cgroups = (c.CT + COT - 1) / COT;
    const unsigned grid = (unsigned)((long long)B * cgroups * (wide ? a.tiles_x / 2 : a.tiles_x) * a.tiles_y);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (vq_trace()) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
    static const bool few_off = getenv("WMAR_CONV_FEW_OFF") != nullptr;      // A/B: keep the MFMA tile for the output conv
    const bool few = c.wf && c.ks == 3 && stride == 1 && !up && !res && !stats && wq_bstride == 0 && c.cout <= 4 && a.Ho % CF_T == 0 &&
                     a.Wo % CF_T == 0 && !few_off;
    if (few) {
        a.tiles_x = a.Wo / CF_T; a.tiles_y = a.Ho / CF_T;
        const unsigned gridf = (unsigned)((long long)B * a.tiles_x * a.tiles_y);
        const size_t ldsf = (size_t)CF_PW * CF_PW * CONV_PSTRIDE * sizeof(float);
        if (c.cout == 3) hipLaunchKernelGGL(k_conv_few<3>, dim3(gridf), dim3(256), ldsf, st, a, (const float*)c.wf);
        else hipLaunchKernelGGL(k_conv_few<4>, dim3(gridf), dim3(256), ldsf, st, a, (const float*)c.wf);
    } else if (bx2) {
        static const hipError_t lds2_ok = hipFuncSetAttribute((const void*)k_conv_bx<4, 3, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        WMAR_REQUIRE(lds2_ok == hipSuccess, "k_conv_bx (stride 2): raising the dynamic LDS limit failed: %s", hipGetErrorString(lds2_ok));
        hipLaunchKernelGGL((k_conv_bx<4, 3, 2, 2>), dim3(grid), dim3(256), lds, st, a);
    } else if (wide) {
        // 75 KB of dynamic LDS (two 10 x 18-pixel patches): above the 64 KB a kernel gets without asking
        static const hipError_t lds_ok = hipFuncSetAttribute((const void*)k_conv_bx<4, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        WMAR_REQUIRE(lds_ok == hipSuccess, "k_conv_bx: raising the dynamic LDS limit failed: %s", hipGetErrorString(lds_ok));
        hipLaunchKernelGGL((k_conv_bx<4, 3, 4>), dim3(grid), dim3(256), lds, st, a);
    }
    else if (bx && COT == 4 && c.ks == 3) hipLaunchKernelGGL((k_conv_bx<4, 3>), dim3(grid), dim3(256), lds, st, a);
    else if (bx && COT == 4) hipLaunchKernelGGL((k_conv_bx<4, 1>), dim3(grid), dim3(256), lds, st, a);
    else if (bx && COT == 2 && c.ks == 3) hipLaunchKernelGGL((k_conv_bx<2, 3>), dim3(grid), dim3(128), lds, st, a);
    else if (bx && COT == 2) hipLaunchKernelGGL((k_conv_bx<2, 1>), dim3(grid), dim3(128), lds, st, a);
    else if (bx && c.ks == 3) hipLaunchKernelGGL((k_conv_bx<1, 3>), dim3(grid), dim3(64), lds, st, a);
    else if (bx) hipLaunchKernelGGL((k_conv_bx<1, 1>), dim3(grid), dim3(64), lds, st, a);
    else if (COT == 4) hipLaunchKernelGGL(k_conv<4>, dim3(grid), dim3(256), lds, st, a);
    else if (COT == 2) hipLaunchKernelGGL(k_conv<2>, dim3(grid), dim3(128), lds, st, a);
    else hipLaunchKernelGGL(k_conv<1>, dim3(grid), dim3(64), lds, st, a);
    if (vq_trace()) {
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        double fl = 2.0 * B * a.Ho * a.Wo * (double)c.cout * c.cin * c.ks * c.ks;
        fprintf(stderr, "conv %4d->%4d k%d s%d up%d %3dx%-3d  %8.1f us  %6.1f TF/s  grid %u\n", c.cin, c.cout, c.ks, stride, up, a.Ho, a.Wo,
                ms * 1e3, fl / (ms * 1e-3) * 1e-12, grid);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    return launch_status("k_conv");
}

// nearest-code search: k_vq_argmin_split when the shape allows it (>= 128 pixels, 8 | E / 8, z tile within the LDS), else k_vq_argmin
int run_vq_argmin(VqArgs a, unsigned long long* best, hipStream_t st) {
    static const bool split_off = getenv("WMAR_VQ_NO_SPLIT") != nullptr;      // A/B
    const int KB = a.E >> 3, NT = a.n_embed >> 5;
    const size_t lds4 = (size_t)128 * (a.E + 4) * sizeof(float);
    if (!split_off && best && a.P >= 128 && KB % 8 == 0 && NT >= 4 && lds4 <= 150 * 1024) {
        const long long groups = (a.P + 127) / 128;
        int CS = 1;                                        // code splits: fill 256 CUs, every wave keeps at least one 32-code tile
        while (groups * CS * 2 <= 256 && NT % (CS * 2) == 0 && NT / (CS * 2) >= 4) CS *= 2;
        WMAR_HIP_CHECK(hipMemsetAsync(best, 0xff, (size_t)a.P * 8, st));
        static const hipError_t lds_ok = hipFuncSetAttribute((const void*)k_vq_argmin_split<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);      // + 4 KB static
        WMAR_REQUIRE(lds_ok == hipSuccess, "k_vq_argmin_split: raising the dynamic LDS limit failed: %s", hipGetErrorString(lds_ok));
        hipLaunchKernelGGL(k_vq_argmin_split<4>, dim3((unsigned)(groups * CS)), dim3(256), lds4, st, a, best, CS);
        hipLaunchKernelGGL(k_vq_finish, dim3((unsigned)((a.P + 255) / 256)), dim3(256), 0, st, (const unsigned long long*)best, a.codes, a.P);
        return launch_status("k_vq_argmin_split");
    }
    const size_t lds = (size_t)64 * (a.E + 4) * sizeof(float);
    WMAR_HIP_CHECK(hipFuncSetAttribute((const void*)k_vq_argmin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_vq_argmin, dim3((unsigned)((a.P + 63) / 64)), dim3(256), lds, st, a);
    return launch_status("k_vq_argmin");
}

// statistics of GroupNorm(x); the (mean, rstd) table lives behind the partial sums in the same allocation
int run_gn(double* gn_partial, const NormW& n, const float* x, int B, int HW, int swish, hipStream_t st, GnRef* out) {
    GnArgs a{};
    a.x = x; a.y = nullptr; a.partial = gn_partial + GN_MR_DOUBLES; a.gamma = n.g; a.beta = n.b; a.HW = HW; a.C = n.C; a.swish = swish;
    WMAR_REQUIRE(n.C % 32 == 0 && n.C / 4 <= 256, "GroupNorm channel count %d unsupported (multiple of 32, <= 1024)", n.C);
    int nchunk = HW / 256;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > GN_CHUNKS_MAX) nchunk = GN_CHUNKS_MAX;
    a.nchunk = nchunk;
    WMAR_REQUIRE(B <= GN_MR_DOUBLES / 32, "GroupNorm: batch %d too large", B);
    float2* mr = (float2*)gn_partial;
    if (g_trk.part && g_trk.src == x && g_trk.C == n.C && g_trk.tiles * 64 == HW) {
        // the conv that produced x already left per-tile sums behind: no pass over the tensor
        hipLaunchKernelGGL(k_gn_finalize_tiles, dim3(B), dim3(256), 0, st, (const double*)g_trk.part, g_trk.tiles,
                           (double)HW * (n.C / 32), mr);
    } else {
        hipLaunchKernelGGL(k_gn_partial, dim3(nchunk, B), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_gn_finalize, dim3(B), dim3(32), 0, st, a, mr);
    }
    out->mr = mr; out->g = n.g; out->b = n.b; out->C = n.C; out->swish = swish;
    return launch_status("k_gn");
}

// x (buf X) -> result left in returned buffer index; uses the 4 rotating buffers
struct Bufs {
    float** buf;
    int x = 0;          // index of the current activation
    float* X() { return buf[x]; }
    float* other(int k) { return buf[(x + k) & 3]; }
    void advance(int k) { x = (x + k) & 3; }
};

int run_res(wmar_vq* v, const ResW& r, Bufs& bf, int B, int H, int W, hipStream_t st) {
    int rc;
    float *X = bf.X(), *A = bf.other(1), *T = bf.other(2), *C = bf.other(3);
    GnRef gn{};
    if ((rc = run_gn(v->gn_partial, r.n1, X, B, H * W, 1, st, &gn))) return rc;
    if ((rc = run_conv(r.c1, X, T, nullptr, B, H, W, 1, 0, st, &gn))) return rc;
    if ((rc = run_gn(v->gn_partial, r.n2, T, B, H * W, 1, st, &gn))) return rc;
    const float* shortcut = X;
    if (r.has_nin) {
        if ((rc = run_conv(r.nin, X, C, nullptr, B, H, W, 1, 0, st))) return rc;
        shortcut = C;
    }
    if ((rc = run_conv(r.c2, T, A, shortcut, B, H, W, 1, 0, st, &gn))) return rc;
    bf.advance(1);
    return WMAR_OK;
}

int run_attn(wmar_vq* v, const AttnW& w, Bufs& bf, int B, int H, int W, hipStream_t st) {
    int rc;
    const int N = H * W, C = w.n.C;
    float *X = bf.X(), *A = bf.other(1), *T = bf.other(2);
    GnRef gn{};
    if ((rc = run_gn(v->gn_partial, w.n, X, B, N, 0, st, &gn))) return rc;
    if ((rc = run_conv(w.q, X, v->aq, nullptr, B, H, W, 1, 0, st, &gn))) return rc;
    if ((rc = run_conv(w.k, X, v->ak, nullptr, B, H, W, 1, 0, st, &gn))) return rc;
    if ((rc = run_conv(w.v, X, v->av, nullptr, B, H, W, 1, 0, st, &gn))) return rc;
    const float scale = 1.0f / sqrtf((float)C);   // int(c)**(-0.5)
    if (v->attk && N % 32 == 0 && C % 32 == 0 && H % 8 == 0 && W % 8 == 0 && N >= 64 && C >= 64 && !conv_no_bx()) {
        // Both products as 1x1 convolutions with PER-IMAGE weights on the bf16 matrix pipe (k_conv_bx): scores = conv(q; weights K_b),
        // out = conv(softmax(scores); weights V_b^T).  K_b and V_b^T are split into bf16 pieces by the weight packer.
        ConvW ck{}, cv{};
        ck.wq = v->attk; ck.bias = v->zbias; ck.cin = ck.cin_s = C; ck.cout = ck.cout_s = N; ck.ks = 1; ck.CT = N / 32; ck.KBc = C / 8;
        cv.wq = v->attv; cv.bias = v->zbias; cv.cin = cv.cin_s = N; cv.cout = cv.cout_s = C; cv.ks = 1; cv.CT = C / 32; cv.KBc = N / 8;
        const long long sk = (long long)ck.CT * (C / 16) * 192, sv = (long long)cv.CT * (N / 16) * 192;
        const long long nk = (long long)ck.CT * (C / 16) * 64, nv = (long long)cv.CT * (N / 16) * 64;
        hipLaunchKernelGGL(k_pack_conv_bx, dim3((unsigned)((nk + 255) / 256), B), dim3(256), 0, st, (const float*)v->ak, v->attk, N, C, 1, ck.CT,
                           C / 16, (long long)N * C, sk, 0);
        hipLaunchKernelGGL(k_pack_conv_bx, dim3((unsigned)((nv + 255) / 256), B), dim3(256), 0, st, (const float*)v->av, v->attv, C, N, 1, cv.CT,
                           N / 16, (long long)N * C, sv, 1);
        if ((rc = launch_status("k_pack_conv_bx"))) return rc;
        if ((rc = run_conv(ck, v->aq, v->asc, nullptr, B, H, W, 1, 0, st, nullptr, sk))) return rc;
        hipLaunchKernelGGL(k_attn_softmax, dim3((unsigned)(((long long)B * N + 3) / 4)), dim3(256), 0, st, v->asc, (long long)B * N, N, scale);
        if ((rc = launch_status("k_attn_softmax"))) return rc;
        if ((rc = run_conv(cv, v->asc, v->ao, nullptr, B, H, W, 1, 0, st, nullptr, sv))) return rc;
    } else {
        hipLaunchKernelGGL(k_attn_scores, dim3(N, B), dim3(256), (size_t)C * 4, st, v->aq, v->ak, v->asc, N, C, scale);
        hipLaunchKernelGGL(k_attn_pv, dim3(N, B), dim3(256), (size_t)N * 4, st, v->asc, v->av, v->ao, N, C);
        if ((rc = launch_status("k_attn"))) return rc;
    }
    if ((rc = run_conv(w.proj, v->ao, T, X, B, H, W, 1, 0, st))) return rc;
    bf.advance(2);
    return WMAR_OK;
}

}  // namespace

extern "C" {

int wmar_vq_create(const wmar_vq_config* cfg, const char* const* names, const void* const* tensors_dev,
                   int32_t n_tensors, void* stream, wmar_vq** out) {
    WMAR_REQUIRE(cfg && names && tensors_dev && out, "vq_create: null argument");
    WMAR_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= 8, "vq_create: bad ch_mult length");
    WMAR_REQUIRE(cfg->max_batch >= 1, "vq_create: max_batch");
    WMAR_REQUIRE(cfg->embed_dim % 8 == 0 && cfg->n_embed % 32 == 0, "embed_dim %% 8 and n_embed %% 32 must be 0");
    WMAR_REQUIRE(cfg->ch % 32 == 0, "ch must be a multiple of 32 (GroupNorm has 32 groups)");
    const int L = cfg->n_levels;
    const int S = cfg->resolution >> (L - 1);
    WMAR_REQUIRE(S >= 8 && S % 8 == 0 && (S << (L - 1)) == cfg->resolution, "latent size %d must be a multiple of 8", S);
    auto* v = new wmar_vq();
    v->cfg = *cfg; v->Bmax = cfg->max_batch; v->S = S;
    ArenaRef arena{&v->allocs, &v->bytes};
    Loader ld;
    ld.v = &arena; ld.st = (hipStream_t)stream;
    for (int i = 0; i < n_tensors; ++i) ld.m[names[i]] = tensors_dev[i];
    const int ch = cfg->ch, z = cfg->z_channels, E = cfg->embed_dim;

    // ---- decoder (model.py:437-505)
    int block_in = ch * cfg->ch_mult[L - 1];
    int res = S;
    ld.conv("post_quant_conv", E, z, 1, v->post_quant);
    ld.conv("decoder.conv_in", z, block_in, 3, v->d_conv_in);
    ld.res("decoder.mid.block_1.", block_in, block_in, v->d_mid1);
    ld.attn("decoder.mid.attn_1.", block_in, v->d_midattn);
    ld.res("decoder.mid.block_2.", block_in, block_in, v->d_mid2);
    v->d_up.resize(L); v->d_upattn.resize(L); v->d_upsample.resize(L);
    for (int lvl = L - 1; lvl >= 0; --lvl) {
        const int block_out = ch * cfg->ch_mult[lvl];
        v->d_up[lvl].resize(cfg->num_res_blocks + 1);
        for (int b = 0; b <= cfg->num_res_blocks; ++b) {
            const std::string p = "decoder.up." + std::to_string(lvl) + ".";
            ld.res(p + "block." + std::to_string(b) + ".", block_in, block_out, v->d_up[lvl][b]);
            block_in = block_out;
            if (in_attn_res(*cfg, res)) {
                v->d_upattn[lvl].emplace_back();
                ld.attn(p + "attn." + std::to_string(b) + ".", block_in, v->d_upattn[lvl].back());
            }
        }
        if (lvl != 0) {
            ld.conv("decoder.up." + std::to_string(lvl) + ".upsample.conv", block_in, block_in, 3, v->d_upsample[lvl]);
            res *= 2;
        }
    }
    ld.norm("decoder.norm_out", block_in, v->d_norm_out);
    ld.conv("decoder.conv_out", block_in, cfg->out_ch, 3, v->d_conv_out);

    // ---- encoder (model.py:343-404)
    ld.conv("encoder.conv_in", cfg->in_channels, ch, 3, v->e_conv_in);
    v->e_down.resize(L); v->e_downattn.resize(L); v->e_downsample.resize(L);
    res = cfg->resolution;
    block_in = ch;
    for (int lvl = 0; lvl < L; ++lvl) {
        block_in = ch * (lvl == 0 ? 1 : cfg->ch_mult[lvl - 1]);
        const int block_out = ch * cfg->ch_mult[lvl];
        v->e_down[lvl].resize(cfg->num_res_blocks);
        for (int b = 0; b < cfg->num_res_blocks; ++b) {
            const std::string p = "encoder.down." + std::to_string(lvl) + ".";
            ld.res(p + "block." + std::to_string(b) + ".", block_in, block_out, v->e_down[lvl][b]);
            block_in = block_out;
            if (in_attn_res(*cfg, res)) {
                v->e_downattn[lvl].emplace_back();
                ld.attn(p + "attn." + std::to_string(b) + ".", block_in, v->e_downattn[lvl].back());
            }
        }
        if (lvl != L - 1) {
            ld.conv("encoder.down." + std::to_string(lvl) + ".downsample.conv", block_in, block_in, 3, v->e_downsample[lvl]);
            res /= 2;
        }
    }
    ld.res("encoder.mid.block_1.", block_in, block_in, v->e_mid1);
    ld.attn("encoder.mid.attn_1.", block_in, v->e_midattn);
    ld.res("encoder.mid.block_2.", block_in, block_in, v->e_mid2);
    ld.norm("encoder.norm_out", block_in, v->e_norm_out);
    ld.conv("encoder.conv_out", block_in, z, 3, v->e_conv_out);
    ld.conv("quant_conv", z, E, 1, v->quant);

    // ---- quantizer
    int rc = ld.rc;
    const float* emb = ld.need("quantize.embedding.weight");
    rc = ld.rc;
    hipStream_t st = (hipStream_t)stream;
#define TRY(x) do { if (rc == WMAR_OK) rc = (x); } while (0)
    TRY(v->alloc(&v->emb, (size_t)cfg->n_embed * E));
    if (rc == WMAR_OK && hipMemcpyAsync(v->emb, emb, (size_t)cfg->n_embed * E * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("embedding copy failed"); rc = WMAR_EHIP;
    }
    TRY(v->alloc(&v->emb_p, (size_t)cfg->n_embed * E / 4));
    TRY(v->alloc(&v->enorm, (size_t)cfg->n_embed));
    if (rc == WMAR_OK) {
        // the codebook as a 1x1 "conv" weight [n_embed][E]: same fragment packing
        size_t n = (size_t)(cfg->n_embed / 32) * (E / 8) * 64;
        hipLaunchKernelGGL(k_pack_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, emb, v->emb_p, cfg->n_embed, E, 1,
                           cfg->n_embed / 32, E / 8);
        hipLaunchKernelGGL(k_row_sqnorm, dim3(cfg->n_embed), dim3(64), 0, st, v->emb, v->enorm, E);
        rc = launch_status("codebook pack");
    }
    // ---- workspaces: the largest activation any layer produces
    size_t maxel = 0;
    {
        int r = cfg->resolution;
        for (int lvl = 0; lvl < L; ++lvl) {
            int cmax = ch * cfg->ch_mult[lvl];
            if (lvl > 0 && ch * cfg->ch_mult[lvl - 1] > cmax) cmax = ch * cfg->ch_mult[lvl - 1];
            if (lvl + 1 < L && ch * cfg->ch_mult[lvl + 1] > cmax) cmax = ch * cfg->ch_mult[lvl + 1];
            size_t el = (size_t)r * r * pad8(cmax);
            if (el > maxel) maxel = el;
            r /= 2;
        }
        size_t el0 = (size_t)cfg->resolution * cfg->resolution * pad8(cfg->in_channels > cfg->out_ch ? cfg->in_channels : cfg->out_ch);
        if (el0 > maxel) maxel = el0;
    }
    v->buf_elems = maxel * v->Bmax;
    for (int i = 0; i < 4; ++i) TRY(v->alloc(&v->buf[i], v->buf_elems));
    const int cattn = ch * cfg->ch_mult[L - 1];
    // attention scratch sized for the largest attention resolution
    int amax = 0;
    for (int i = 0; i < cfg->n_attn_res; ++i) amax = cfg->attn_resolutions[i] > amax ? cfg->attn_resolutions[i] : amax;
    if (amax < S) amax = S;
    const size_t ntok = (size_t)amax * amax;
    int cam = 0;
    for (int lvl = 0; lvl < L; ++lvl) cam = ch * cfg->ch_mult[lvl] > cam ? ch * cfg->ch_mult[lvl] : cam;
    (void)cattn;
    TRY(v->alloc(&v->aq, (size_t)v->Bmax * ntok * cam));
    TRY(v->alloc(&v->ak, (size_t)v->Bmax * ntok * cam));
    TRY(v->alloc(&v->av, (size_t)v->Bmax * ntok * cam));
    TRY(v->alloc(&v->ao, (size_t)v->Bmax * ntok * cam));
    TRY(v->alloc(&v->asc, (size_t)v->Bmax * ntok * ntok));
    if (ntok % 32 == 0 && cam % 32 == 0) {
        TRY(v->alloc(&v->attk, (size_t)v->Bmax * ntok * cam * 3 / 8));     // 3 pieces x 2 bytes per element, in 16-byte units
        TRY(v->alloc(&v->attv, (size_t)v->Bmax * ntok * cam * 3 / 8));
        const size_t nz = ntok > (size_t)cam ? ntok : (size_t)cam;
        TRY(v->alloc(&v->zbias, nz));
        if (rc == WMAR_OK && hipMemsetAsync(v->zbias, 0, nz * 4, st) != hipSuccess) { set_error("vq_create: memset failed"); rc = WMAR_EHIP; }
    }
    TRY(v->alloc(&v->gn_partial, (size_t)GN_MR_DOUBLES + (size_t)v->Bmax * GN_CHUNKS_MAX * 32 * 2));
    v->gn_tiles_cap = (long long)v->Bmax * (cfg->resolution / 8) * (cfg->resolution / 8) * 64;
    TRY(v->alloc(&v->gn_tiles, (size_t)v->gn_tiles_cap));
    TRY(v->alloc(&v->znorm, (size_t)v->Bmax * S * S));
    TRY(v->alloc(&v->vqbest, (size_t)v->Bmax * S * S));
    if (rc == WMAR_OK && hipStreamSynchronize(st) != hipSuccess) { set_error("vq_create: sync failed"); rc = WMAR_EHIP; }
#undef TRY
    if (rc != WMAR_OK) { delete v; return rc; }
    *out = v;
    return WMAR_OK;
}

void wmar_vq_destroy(wmar_vq* v) { delete v; }
int64_t wmar_vq_device_bytes(const wmar_vq* v) { return v ? v->bytes : 0; }

int wmar_vq_decode(wmar_vq* v, const int64_t* codes_dev, int64_t B, float* images_dev, void* stream) {
    WMAR_REQUIRE(v && codes_dev && images_dev, "vq_decode: null argument");
    WMAR_REQUIRE(B >= 1 && B <= v->Bmax, "vq_decode: batch %lld outside 1..%d", (long long)B, v->Bmax);
    hipStream_t st = (hipStream_t)stream;
    const wmar_vq_config& c = v->cfg;
    const int L = c.n_levels, S = v->S, E = c.embed_dim;
    int rc;
    Bufs bf{v->buf};
    g_trk = GnTrack{};
    g_trk.part = v->gn_tiles; g_trk.cap = v->gn_tiles_cap;
    // get_codebook_entry (quantize.py:316-331): z_q in NHWC is just the gathered rows
    const long long npix = (long long)B * S * S;
    hipLaunchKernelGGL(k_codebook_gather, dim3((unsigned)((npix * (E / 4) + 255) / 256)), dim3(256), 0, st,
                       (const long long*)codes_dev, v->emb, bf.X(), npix, E, c.n_embed);
    if ((rc = launch_status("k_codebook_gather"))) return rc;
    if ((rc = run_conv(v->post_quant, bf.X(), bf.other(1), nullptr, (int)B, S, S, 1, 0, st))) return rc;
    bf.advance(1);
    if ((rc = run_conv(v->d_conv_in, bf.X(), bf.other(1), nullptr, (int)B, S, S, 1, 0, st))) return rc;
    bf.advance(1);
    int H = S;
    if ((rc = run_res(v, v->d_mid1, bf, (int)B, H, H, st))) return rc;
    if ((rc = run_attn(v, v->d_midattn, bf, (int)B, H, H, st))) return rc;
    if ((rc = run_res(v, v->d_mid2, bf, (int)B, H, H, st))) return rc;
    for (int lvl = L - 1; lvl >= 0; --lvl) {
        for (int b = 0; b <= c.num_res_blocks; ++b) {
            if ((rc = run_res(v, v->d_up[lvl][b], bf, (int)B, H, H, st))) return rc;
            if (!v->d_upattn[lvl].empty())
                if ((rc = run_attn(v, v->d_upattn[lvl][b], bf, (int)B, H, H, st))) return rc;
        }
        if (lvl != 0) {
            if ((rc = run_conv(v->d_upsample[lvl], bf.X(), bf.other(1), nullptr, (int)B, H, H, 1, 1, st))) return rc;
            bf.advance(1);
            H *= 2;
        }
    }
    GnRef gno{};
    if ((rc = run_gn(v->gn_partial, v->d_norm_out, bf.X(), (int)B, H * H, 1, st, &gno))) return rc;
    if ((rc = run_conv(v->d_conv_out, bf.X(), bf.other(2), nullptr, (int)B, H, H, 1, 0, st, &gno))) return rc;
    const int HW = H * H;
    hipLaunchKernelGGL(k_nhwc_to_nchw_clamp, dim3((HW + 255) / 256, (unsigned)B), dim3(256), 0, st, bf.other(2), images_dev,
                       c.out_ch, HW, v->d_conv_out.cout_s);
    return launch_status("k_nhwc_to_nchw_clamp");
}

int wmar_vq_encode(wmar_vq* v, const float* images_dev, int64_t B, int64_t* codes_dev, float* prequant_dev,
                   void* stream) {
    WMAR_REQUIRE(v && images_dev && codes_dev, "vq_encode: null argument");
    WMAR_REQUIRE(B >= 1 && B <= v->Bmax, "vq_encode: batch %lld outside 1..%d", (long long)B, v->Bmax);
    hipStream_t st = (hipStream_t)stream;
    const wmar_vq_config& c = v->cfg;
    const int L = c.n_levels, S = v->S, E = c.embed_dim;
    int rc;
    Bufs bf{v->buf};
    g_trk = GnTrack{};
    g_trk.part = v->gn_tiles; g_trk.cap = v->gn_tiles_cap;
    int H = c.resolution;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((H * H + 255) / 256, (unsigned)B), dim3(256), 0, st, images_dev, bf.X(),
                       c.in_channels, H * H, v->e_conv_in.cin_s);
    if ((rc = launch_status("k_nchw_to_nhwc"))) return rc;
    if ((rc = run_conv(v->e_conv_in, bf.X(), bf.other(1), nullptr, (int)B, H, H, 1, 0, st))) return rc;
    bf.advance(1);
    for (int lvl = 0; lvl < L; ++lvl) {
        for (int b = 0; b < c.num_res_blocks; ++b) {
            if ((rc = run_res(v, v->e_down[lvl][b], bf, (int)B, H, H, st))) return rc;
            if (!v->e_downattn[lvl].empty())
                if ((rc = run_attn(v, v->e_downattn[lvl][b], bf, (int)B, H, H, st))) return rc;
        }
        if (lvl != L - 1) {
            if ((rc = run_conv(v->e_downsample[lvl], bf.X(), bf.other(1), nullptr, (int)B, H, H, 2, 0, st))) return rc;
            bf.advance(1);
            H /= 2;
        }
    }
    if ((rc = run_res(v, v->e_mid1, bf, (int)B, H, H, st))) return rc;
    if ((rc = run_attn(v, v->e_midattn, bf, (int)B, H, H, st))) return rc;
    if ((rc = run_res(v, v->e_mid2, bf, (int)B, H, H, st))) return rc;
    GnRef gne{};
    if ((rc = run_gn(v->gn_partial, v->e_norm_out, bf.X(), (int)B, H * H, 1, st, &gne))) return rc;
    if ((rc = run_conv(v->e_conv_out, bf.X(), bf.other(2), nullptr, (int)B, H, H, 1, 0, st, &gne))) return rc;
    if ((rc = run_conv(v->quant, bf.other(2), bf.other(3), nullptr, (int)B, H, H, 1, 0, st))) return rc;
    float* zq = bf.other(3);   // [B*S*S][E] (E is a multiple of 8: no channel padding)
    const long long P = (long long)B * S * S;
    if (prequant_dev) WMAR_HIP_CHECK(hipMemcpyAsync(prequant_dev, zq, (size_t)P * E * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)P), dim3(64), 0, st, zq, v->znorm, E);
    VqArgs a{};
    a.z = zq; a.ep = v->emb_p; a.enorm = v->enorm; a.znorm = v->znorm; a.codes = (long long*)codes_dev; a.P = P;
    a.E = E; a.n_embed = c.n_embed;
    return run_vq_argmin(a, v->vqbest, st);
}

}  // extern "C"

// ============================================================================ MaskGIT-VQGAN
// RAR's tokenizer (deps/rar/modeling/modules/maskgit_vqgan.py, deps/rar/modeling/titok.py:41-89):
// the same conv / GroupNorm / quantizer kernels with a different plan -- no attention, bias-free
// 3x3 convs inside the ResnetBlocks, the 1x1 shortcut applied to the block OUTPUT, average-pool
// downsampling, images in [0, 1] on the tokenizer side and [-1, 1] on the wrapper side.
namespace wmar {

// 2x2 average pool, NHWC (DownsamplingBlock.forward, maskgit_vqgan.py:118-119)
__global__ void k_avgpool2(const float* __restrict__ in, float* __restrict__ out, int Ho, int Wo, int C) {
    const long long q4 = C >> 2;
    const long long total = (long long)Ho * Wo * q4;
    const long long b = blockIdx.y;
    const float* ib = in + b * (long long)(2 * Ho) * (2 * Wo) * C;
    float* ob = out + b * (long long)Ho * Wo * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % q4) * 4;
        const long long pix = e / q4;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        const float* p = ib + ((long long)(2 * oy) * (2 * Wo) + 2 * ox) * C + c;
        const float4 a = *(const float4*)p, b2 = *(const float4*)(p + C);
        const float4 c2 = *(const float4*)(p + (long long)2 * Wo * C), d = *(const float4*)(p + (long long)2 * Wo * C + C);
        // F.avg_pool2d: sum of the window (row-major order) divided by the window size
        *(float4*)(ob + pix * C + c) = make_float4((a.x + b2.x + c2.x + d.x) / 4.0f, (a.y + b2.y + c2.y + d.y) / 4.0f,
                                                   (a.z + b2.z + c2.z + d.z) / 4.0f, (a.w + b2.w + c2.w + d.w) / 4.0f);
    }
}

// NCHW [-1,1] -> NHWC [0,1]:  (x + 1) / 2   (rar_wrapper.py:124)
__global__ void k_nchw_to_nhwc01(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, int Cs) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = blockIdx.y;
    if (idx >= HW) return;
    for (int c = 0; c < Cs; ++c)
        dst[((long long)b * HW + idx) * Cs + c] = c < C ? (src[((long long)b * C + c) * HW + idx] + 1.0f) / 2.0f : 0.f;
}

// NHWC -> NCHW: clamp(x, 0, 1) (titok.py:84) then clamp(x*2 - 1, -1, 1) (rar_wrapper.py:113-114)
__global__ void k_nhwc_to_nchw_01(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, int Cs) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = blockIdx.y;
    if (idx >= HW) return;
    for (int c = 0; c < C; ++c) {
        float v = src[((long long)b * HW + idx) * Cs + c];
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        v = v * 2.0f - 1.0f;
        dst[((long long)b * C + c) * HW + idx] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}

}  // namespace wmar

struct wmar_mvq {
    wmar_mvq_config cfg{};
    std::vector<void*> allocs;
    int64_t bytes = 0;
    int Bmax = 0, S = 0;
    ConvW d_conv_in, d_conv_out, e_conv_in, e_conv_out;
    std::vector<ResW> d_mid, e_mid;
    std::vector<std::vector<ResW>> d_up, e_down;
    std::vector<ConvW> d_upconv;
    NormW d_norm_out, e_norm_out;
    float* emb = nullptr; float4* emb_p = nullptr; float* enorm = nullptr;
    float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    double* gn_partial = nullptr;
    double* gn_tiles = nullptr; long long gn_tiles_cap = 0;   // per-tile GroupNorm partial sums written by conv epilogues
    float* znorm = nullptr;
    unsigned long long* vqbest = nullptr;      // packed (distance, code) winners of k_vq_argmin_split
    ~wmar_mvq() { for (void* p : allocs) (void)hipFree(p); }
};

namespace {

void mres_load(Loader& ld, const std::string& p, int cin, int cout, ResW& r) {
    ld.norm(p + "norm1", cin, r.n1);
    ld.conv(p + "conv1", cin, cout, 3, r.c1, false);
    ld.norm(p + "norm2", cout, r.n2);
    ld.conv(p + "conv2", cout, cout, 3, r.c2, false);
    r.has_nin = cin != cout;
    if (r.has_nin) ld.conv(p + "nin_shortcut", cout, cout, 1, r.nin, false);
}

// ResnetBlock.forward (maskgit_vqgan.py:69-87): out = h + (cin != cout ? nin(h) : x),  h = conv2(...)
int run_mres(wmar_mvq* v, const ResW& r, Bufs& bf, int B, int H, int W, hipStream_t st) {
    int rc;
    float *X = bf.X(), *A = bf.other(1), *T = bf.other(2), *C = bf.other(3);
    GnRef gn{};
    if ((rc = run_gn(v->gn_partial, r.n1, X, B, H * W, 1, st, &gn))) return rc;
    if ((rc = run_conv(r.c1, X, T, nullptr, B, H, W, 1, 0, st, &gn))) return rc;
    if ((rc = run_gn(v->gn_partial, r.n2, T, B, H * W, 1, st, &gn))) return rc;
    if (r.has_nin) {
        if ((rc = run_conv(r.c2, T, A, nullptr, B, H, W, 1, 0, st, &gn))) return rc;
        if ((rc = run_conv(r.nin, A, C, A, B, H, W, 1, 0, st))) return rc;      // nin(h) + h
        bf.advance(3);
    } else {
        if ((rc = run_conv(r.c2, T, A, X, B, H, W, 1, 0, st, &gn))) return rc;   // h + x
        bf.advance(1);
    }
    return WMAR_OK;
}

}  // namespace

extern "C" {

int wmar_mvq_create(const wmar_mvq_config* cfg, const char* const* names, const void* const* tensors_dev,
                    int32_t n_tensors, void* stream, wmar_mvq** out) {
    WMAR_REQUIRE(cfg && names && tensors_dev && out, "mvq_create: null argument");
    WMAR_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= 8 && cfg->max_batch >= 1, "mvq_create: bad config");
    WMAR_REQUIRE(cfg->z_channels % 8 == 0 && cfg->num_embeddings % 32 == 0 && cfg->hidden_channels % 32 == 0,
                 "z_channels %% 8, num_embeddings %% 32 and hidden_channels %% 32 must be 0");
    const int R = cfg->n_levels, hc = cfg->hidden_channels, z = cfg->z_channels;
    const int S = cfg->resolution >> (R - 1);
    WMAR_REQUIRE(S >= 8 && S % 8 == 0 && (S << (R - 1)) == cfg->resolution, "latent size %d must be a multiple of 8", S);
    auto* v = new wmar_mvq();
    v->cfg = *cfg; v->Bmax = cfg->max_batch; v->S = S;
    ArenaRef arena{&v->allocs, &v->bytes};
    Loader ld;
    ld.v = &arena; ld.st = (hipStream_t)stream;
    for (int i = 0; i < n_tensors; ++i) ld.m[names[i]] = tensors_dev[i];
    const int mid = hc * cfg->channel_mult[R - 1];
    // decoder (maskgit_vqgan.py:197-245)
    ld.conv("decoder.conv_in", z, mid, 3, v->d_conv_in);
    v->d_mid.resize(cfg->num_res_blocks);
    for (int b = 0; b < cfg->num_res_blocks; ++b) mres_load(ld, "decoder.mid." + std::to_string(b) + ".", mid, mid, v->d_mid[b]);
    v->d_up.resize(R); v->d_upconv.resize(R);
    for (int lvl = 0; lvl < R; ++lvl) {
        int bi = lvl == R - 1 ? hc * cfg->channel_mult[R - 1] : hc * cfg->channel_mult[lvl + 1];
        const int bo = hc * cfg->channel_mult[lvl];
        v->d_up[lvl].resize(cfg->num_res_blocks);
        for (int b = 0; b < cfg->num_res_blocks; ++b) {
            mres_load(ld, "decoder.up." + std::to_string(lvl) + ".block." + std::to_string(b) + ".", bi, bo, v->d_up[lvl][b]);
            bi = bo;
        }
        if (lvl != 0) ld.conv("decoder.up." + std::to_string(lvl) + ".upsample_conv", bo, bo, 3, v->d_upconv[lvl]);
    }
    ld.norm("decoder.norm_out", hc * cfg->channel_mult[0], v->d_norm_out);
    ld.conv("decoder.conv_out", hc * cfg->channel_mult[0], cfg->num_channels, 3, v->d_conv_out);
    // encoder (maskgit_vqgan.py:157-194)
    ld.conv("encoder.conv_in", cfg->num_channels, hc, 3, v->e_conv_in, false);
    v->e_down.resize(R);
    for (int lvl = 0; lvl < R; ++lvl) {
        int bi = hc * (lvl == 0 ? 1 : cfg->channel_mult[lvl - 1]);
        const int bo = hc * cfg->channel_mult[lvl];
        v->e_down[lvl].resize(cfg->num_res_blocks);
        for (int b = 0; b < cfg->num_res_blocks; ++b) {
            mres_load(ld, "encoder.down." + std::to_string(lvl) + ".block." + std::to_string(b) + ".", bi, bo, v->e_down[lvl][b]);
            bi = bo;
        }
    }
    v->e_mid.resize(cfg->num_res_blocks);
    for (int b = 0; b < cfg->num_res_blocks; ++b) mres_load(ld, "encoder.mid." + std::to_string(b) + ".", mid, mid, v->e_mid[b]);
    ld.norm("encoder.norm_out", mid, v->e_norm_out);
    ld.conv("encoder.conv_out", mid, z, 1, v->e_conv_out);
    const float* emb = ld.need("quantize.embedding.weight");
    int rc = ld.rc;
    hipStream_t st = (hipStream_t)stream;
#define TRY(x) do { if (rc == WMAR_OK) rc = (x); } while (0)
    TRY(arena.alloc(&v->emb, (size_t)cfg->num_embeddings * z));
    if (rc == WMAR_OK && hipMemcpyAsync(v->emb, emb, (size_t)cfg->num_embeddings * z * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("embedding copy failed"); rc = WMAR_EHIP;
    }
    TRY(arena.alloc(&v->emb_p, (size_t)cfg->num_embeddings * z / 4));
    TRY(arena.alloc(&v->enorm, (size_t)cfg->num_embeddings));
    if (rc == WMAR_OK) {
        size_t n = (size_t)(cfg->num_embeddings / 32) * (z / 8) * 64;
        hipLaunchKernelGGL(k_pack_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, emb, v->emb_p, cfg->num_embeddings, z, 1,
                           cfg->num_embeddings / 32, z / 8);
        hipLaunchKernelGGL(k_row_sqnorm, dim3(cfg->num_embeddings), dim3(64), 0, st, v->emb, v->enorm, z);
        rc = launch_status("codebook pack");
    }
    size_t maxel = 0;
    {
        int r = cfg->resolution;
        for (int lvl = 0; lvl < R; ++lvl) {
            int cmax = hc * cfg->channel_mult[lvl];
            if (lvl > 0 && hc * cfg->channel_mult[lvl - 1] > cmax) cmax = hc * cfg->channel_mult[lvl - 1];
            if (lvl + 1 < R && hc * cfg->channel_mult[lvl + 1] > cmax) cmax = hc * cfg->channel_mult[lvl + 1];
            if (z > cmax && lvl == R - 1) cmax = z;
            size_t el = (size_t)r * r * pad8(cmax);
            if (el > maxel) maxel = el;
            r /= 2;
        }
    }
    for (int i = 0; i < 4; ++i) TRY(arena.alloc(&v->buf[i], maxel * v->Bmax));
    TRY(arena.alloc(&v->gn_partial, (size_t)GN_MR_DOUBLES + (size_t)v->Bmax * GN_CHUNKS_MAX * 32 * 2));
    v->gn_tiles_cap = (long long)v->Bmax * (cfg->resolution / 8) * (cfg->resolution / 8) * 64;
    TRY(arena.alloc(&v->gn_tiles, (size_t)v->gn_tiles_cap));
    TRY(arena.alloc(&v->znorm, (size_t)v->Bmax * S * S));
    TRY(arena.alloc(&v->vqbest, (size_t)v->Bmax * S * S));
    if (rc == WMAR_OK && hipStreamSynchronize(st) != hipSuccess) { set_error("mvq_create: sync failed"); rc = WMAR_EHIP; }
#undef TRY
    if (rc != WMAR_OK) { delete v; return rc; }
    *out = v;
    return WMAR_OK;
}

void wmar_mvq_destroy(wmar_mvq* v) { delete v; }
int64_t wmar_mvq_device_bytes(const wmar_mvq* v) { return v ? v->bytes : 0; }

int wmar_mvq_decode(wmar_mvq* v, const int64_t* codes_dev, int64_t B, float* images_dev, void* stream) {
    WMAR_REQUIRE(v && codes_dev && images_dev, "mvq_decode: null argument");
    WMAR_REQUIRE(B >= 1 && B <= v->Bmax, "mvq_decode: batch %lld outside 1..%d", (long long)B, v->Bmax);
    hipStream_t st = (hipStream_t)stream;
    const wmar_mvq_config& c = v->cfg;
    const int R = c.n_levels, S = v->S, z = c.z_channels;
    int rc;
    Bufs bf{v->buf};
    g_trk = GnTrack{};
    g_trk.part = v->gn_tiles; g_trk.cap = v->gn_tiles_cap;
    const long long npix = (long long)B * S * S;
    hipLaunchKernelGGL(k_codebook_gather, dim3((unsigned)((npix * (z / 4) + 255) / 256)), dim3(256), 0, st,
                       (const long long*)codes_dev, v->emb, bf.X(), npix, z, c.num_embeddings);
    if ((rc = launch_status("k_codebook_gather"))) return rc;
    if ((rc = run_conv(v->d_conv_in, bf.X(), bf.other(1), nullptr, (int)B, S, S, 1, 0, st))) return rc;
    bf.advance(1);
    int H = S;
    for (auto& r : v->d_mid)
        if ((rc = run_mres(v, r, bf, (int)B, H, H, st))) return rc;
    for (int lvl = R - 1; lvl >= 0; --lvl) {
        for (auto& r : v->d_up[lvl])
            if ((rc = run_mres(v, r, bf, (int)B, H, H, st))) return rc;
        if (lvl != 0) {
            if ((rc = run_conv(v->d_upconv[lvl], bf.X(), bf.other(1), nullptr, (int)B, H, H, 1, 1, st))) return rc;
            bf.advance(1);
            H *= 2;
        }
    }
    GnRef gno{};
    if ((rc = run_gn(v->gn_partial, v->d_norm_out, bf.X(), (int)B, H * H, 1, st, &gno))) return rc;
    if ((rc = run_conv(v->d_conv_out, bf.X(), bf.other(2), nullptr, (int)B, H, H, 1, 0, st, &gno))) return rc;
    const int HW = H * H;
    hipLaunchKernelGGL(k_nhwc_to_nchw_01, dim3((HW + 255) / 256, (unsigned)B), dim3(256), 0, st, bf.other(2), images_dev,
                       c.num_channels, HW, v->d_conv_out.cout_s);
    return launch_status("k_nhwc_to_nchw_01");
}

int wmar_mvq_encode(wmar_mvq* v, const float* images_dev, int64_t B, int64_t* codes_dev, float* prequant_dev, void* stream) {
    WMAR_REQUIRE(v && images_dev && codes_dev, "mvq_encode: null argument");
    WMAR_REQUIRE(B >= 1 && B <= v->Bmax, "mvq_encode: batch %lld outside 1..%d", (long long)B, v->Bmax);
    hipStream_t st = (hipStream_t)stream;
    const wmar_mvq_config& c = v->cfg;
    const int R = c.n_levels, S = v->S, z = c.z_channels;
    int rc;
    Bufs bf{v->buf};
    g_trk = GnTrack{};
    g_trk.part = v->gn_tiles; g_trk.cap = v->gn_tiles_cap;
    int H = c.resolution;
    hipLaunchKernelGGL(k_nchw_to_nhwc01, dim3((H * H + 255) / 256, (unsigned)B), dim3(256), 0, st, images_dev, bf.X(),
                       c.num_channels, H * H, v->e_conv_in.cin_s);
    if ((rc = launch_status("k_nchw_to_nhwc01"))) return rc;
    if ((rc = run_conv(v->e_conv_in, bf.X(), bf.other(1), nullptr, (int)B, H, H, 1, 0, st))) return rc;
    bf.advance(1);
    for (int lvl = 0; lvl < R; ++lvl) {
        for (auto& r : v->e_down[lvl])
            if ((rc = run_mres(v, r, bf, (int)B, H, H, st))) return rc;
        if (lvl != R - 1) {
            const int C = c.hidden_channels * c.channel_mult[lvl];
            const long long total = (long long)(H / 2) * (H / 2) * (C / 4);
            int gx = (int)((total + 255) / 256);
            if (gx > 8192) gx = 8192;
            hipLaunchKernelGGL(k_avgpool2, dim3(gx, (unsigned)B), dim3(256), 0, st, bf.X(), bf.other(1), H / 2, H / 2, C);
            if (g_trk.src == bf.other(1)) g_trk.src = nullptr;
            if ((rc = launch_status("k_avgpool2"))) return rc;
            bf.advance(1);
            H /= 2;
        }
    }
    for (auto& r : v->e_mid)
        if ((rc = run_mres(v, r, bf, (int)B, H, H, st))) return rc;
    GnRef gne{};
    if ((rc = run_gn(v->gn_partial, v->e_norm_out, bf.X(), (int)B, H * H, 1, st, &gne))) return rc;
    if ((rc = run_conv(v->e_conv_out, bf.X(), bf.other(2), nullptr, (int)B, H, H, 1, 0, st, &gne))) return rc;
    float* zq = bf.other(2);
    const long long P = (long long)B * S * S;
    if (prequant_dev) WMAR_HIP_CHECK(hipMemcpyAsync(prequant_dev, zq, (size_t)P * z * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)P), dim3(64), 0, st, zq, v->znorm, z);
    VqArgs a{};
    a.z = zq; a.ep = v->emb_p; a.enorm = v->enorm; a.znorm = v->znorm; a.codes = (long long*)codes_dev; a.P = P;
    a.E = z; a.n_embed = c.num_embeddings;
    return run_vq_argmin(a, v->vqbest, st);
}

}  // extern "C"
