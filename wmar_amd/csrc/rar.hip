// RAR generator decode engine for gfx950 (fp32, exact-f32 MFMA) -- SURVEY.md section 8a row R1.
//
// Reference: deps/rar/modeling/rar.py:56-118 (Attention with qk-norm and KV cache), :138-183
// (adaLN Block), :123-134 (FinalLayer), :319-405 (forward_fn), :408-459 (generate with
// classifier-free guidance).
//
// Same building blocks and data layout as gpt.hip (decoder_kernels.h): fragment-packed weights
// and activations, split-K slabs, fp64 LayerNorm partial sums.  What is specific to RAR:
//   * every row carries a condition vector c = emb[cond] + timestep[p]; SiLU(c) feeds ONE GEMM per
//     position that produces the adaLN shift/scale/gate of all blocks and of the final layer
//     ([M, 6d*L + 2d], row-major);
//   * LayerNorm outputs are modulated per row AND per channel (x*(1+scale)+shift), so the
//     normalised activation is written explicitly (k_modulate) instead of being folded into the
//     weights; residual updates are gated (k_resid_stats with a gate);
//   * q and k are LayerNorm-ed per head inside the attention prologue; head_dim may be 80;
//   * classifier-free guidance doubles the batch: rows [0,B) conditional, [B,2B) unconditional,
//     mixed inside the fused sampler;
//   * position 0 holds the cls token, position 1 the condition token: 257 positions per image.
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "decoder_host.h"

namespace wmar {

// x0 = token vector + pos_embed[p] (+ target_aware_pos_embed[p+1] for p >= 1), LN partial sums;
// sc = SiLU(emb[cond] + timesteps[p])   (forward_fn, rar.py:346-384)
struct RarEmbedArgs {
    float4* x; float4* sc; double* stats;
    u32x4* scq;              // nullable: SiLU(c) as bf16 pieces as well (the adaLN GEMM on the bf16 pipe)
    const float* emb;        // [n_embeddings][d]
    const float* cls;        // [d]
    const float* pos; const float* tape; const float* tstep;   // [*][d]
    const long long* tok;    // explicit token per row (forward_position) or null
    const long long* cond;   // [M] condition id per row
    const long long* ids;    // [B][ids_stride] generated ids (generate)
    long long ids_stride;
    const int* pos_dev;
    int KB, MT, n_chunks, M, Bhalf, K;
    int MTsc;                // row tiles of sc (rows past MTsc*32 take their modulation from the shared table)
};

static __global__ __launch_bounds__(256) void k_rar_embed(RarEmbedArgs a) {
    constexpr int KPW = 4;
    __shared__ double red[4][64][2];
    const int c = blockIdx.x / a.MT, mt = blockIdx.x % a.MT;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kb0 = (int)((unsigned)c * (unsigned)a.KB / (unsigned)a.n_chunks);
    const int kb1 = (int)((unsigned)(c + 1) * (unsigned)a.KB / (unsigned)a.n_chunks);
    const int m = mt * 32 + (lane & 31), half = lane >> 5;
    const int p = *a.pos_dev;
    const int mm = m < a.M ? m : 0;
    long long tk;
    if (a.tok) tk = a.tok[mm];
    else if (p == 0) tk = -1;
    else if (p == 1) tk = a.cond[mm];
    else tk = a.ids[(long long)(mm % a.Bhalf) * a.ids_stride + (p - 2)];
    const float* trow = tk < 0 ? a.cls : a.emb + tk * a.K;
    const float* crow = a.emb + a.cond[mm] * a.K;
    const float* prow = a.pos + (long long)p * a.K;
    const float* arow = a.tape + (long long)(p + 1) * a.K;
    const float* srow = a.tstep + (long long)p * a.K;
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kb = kb0 + w + 4 * i;
        if (kb >= kb1) continue;
        const int k = kb * 8 + 4 * half;
        const long long idx = ((long long)kb * a.MT + mt) * 64 + lane;
        float4 t = *(const float4*)(trow + k), pe = *(const float4*)(prow + k);
        float4 r = make_float4(t.x + pe.x, t.y + pe.y, t.z + pe.z, t.w + pe.w);
        if (p >= 1) {
            float4 ta = *(const float4*)(arow + k);
            r.x += ta.x; r.y += ta.y; r.z += ta.z; r.w += ta.w;
        }
        a.x[idx] = r;
        s += (double)r.x + (double)r.y + (double)r.z + (double)r.w;
        ss += sq4_f64(r);         // never a v_fmac_f64 chain: common.h
        float4 ce = *(const float4*)(crow + k), te = *(const float4*)(srow + k);
        float cv[4] = {ce.x + te.x, ce.y + te.y, ce.z + te.z, ce.w + te.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) cv[j] = cv[j] / (1.0f + expf(-cv[j]));   // SiLU
        if (mt < a.MTsc) {
            a.sc[((long long)kb * a.MTsc + mt) * 64 + lane] = make_float4(cv[0], cv[1], cv[2], cv[3]);
            if (a.scq) bx_store_planes4(a.scq, a.MTsc, kb, half, mt, lane & 31, make_float4(cv[0], cv[1], cv[2], cv[3]));
        }
    }
    red[w][lane][0] = s; red[w][lane][1] = ss;       // all 64 lanes, no shuffle (decoder_kernels.h, k_qkvx_bx's keeper reduction)
    __syncthreads();
    if (threadIdx.x < 32) {
        double ts = 0, tss = 0;
        for (int i = 0; i < 4; ++i) { ts += red[i][threadIdx.x][0] + red[i][threadIdx.x + 32][0]; tss += red[i][threadIdx.x][1] + red[i][threadIdx.x + 32][1]; }
        double* o = a.stats + ((long long)c * a.MT * 32 + mt * 32 + threadIdx.x) * 2;
        o[0] = ts; o[1] = tss;
    }
}

// h = LayerNorm(x; gamma, beta, eps 1e-6) * (1 + scale) + shift      (modulate, rar.py:120-121)
// gamma/beta null: no affine (FinalLayer.norm_final).  scale/shift: row-major [M][mod_stride].
// The per-row modulations (shift / scale / gate of every block) come out of the adaLN GEMM in the PACKED activation layout
// [Ntot/8][MTm][64] float4 -- the same (row = lane % 32, 4 features by lane / 32) mapping as x, so a wave reads them with one
// coalesced 1-KiB load per k-block (row-major [M][Ntot] rows are 1 MB apart: 32 cache lines per load).
struct ModArgs {
    const float4* x; float4* h; const double* stats;
    const float* gamma; const float* beta;
    const float4* modp; int MTm; long long kb_shift, kb_scale;   // packed modulations, their row tiles, k-block offsets of this shift / scale
    const float* shift_u; const float* scale_u; long long mod_stride;   // nullable: rows >= split share row *pos_dev of the row-major [T][mod_stride] table
    const int* pos_dev; int split;
    int KB, MT, n_chunks, K;
    u32x4* hq;      // nullable: write the modulated rows as bf16 pieces (a k_bx GEMM follows) instead of h
};

static __global__ __launch_bounds__(256) void k_modulate(ModArgs a) {
    const int c = blockIdx.x / a.MT, mt = blockIdx.x % a.MT;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kb0 = (int)((unsigned)c * (unsigned)a.KB / (unsigned)a.n_chunks);
    const int kb1 = (int)((unsigned)(c + 1) * (unsigned)a.KB / (unsigned)a.n_chunks);
    const int m = mt * 32 + (lane & 31), half = lane >> 5;
    const bool shared = a.shift_u && m >= a.split;
    const long long urow = (long long)(*a.pos_dev) * a.mod_stride;
    // every load of this wave's (up to) four k-blocks is issued before the first use: one round trip, not four
    constexpr int KPW = 4;   // the host keeps chunks at <= 16 k-blocks (stat_chunks)
    float4 v[KPW], g4[KPW], b4[KPW], sc[KPW], sh[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kb = min(kb0 + w + 4 * i, kb1 - 1);
        const int k = kb * 8 + 4 * half;
        v[i] = a.x[((long long)kb * a.MT + mt) * 64 + lane];
        g4[i] = a.gamma ? *(const float4*)(a.gamma + k) : make_float4(1.f, 1.f, 1.f, 1.f);
        b4[i] = a.gamma ? *(const float4*)(a.beta + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (shared) {
            sc[i] = *(const float4*)(a.scale_u + urow + k);
            sh[i] = *(const float4*)(a.shift_u + urow + k);
        } else {
            sc[i] = a.modp[((a.kb_scale + kb) * a.MTm + mt) * 64 + lane];
            sh[i] = a.modp[((a.kb_shift + kb) * a.MTm + mt) * 64 + lane];
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // the row statistics are fetched while the data loads are in flight
    float mu, rstd;
    {   // eps 1e-6 (RAR's norm_layer); ln_row_stats uses 1e-5, so finish the statistics here
        double sm = 0, sq = 0;
        for (int c0 = 0; c0 < a.n_chunks; c0 += 16) {
            double2 st[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) st[i] = *(const double2*)(a.stats + ((long long)min(c0 + i, a.n_chunks - 1) * a.MT * 32 + m) * 2);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (c0 + i < a.n_chunks) { sm += st[i].x; sq += st[i].y; }
        }
        const double invK = inv_count_f64((double)a.K);
        const double mean = sm * invK;
        mu = (float)mean;
        rstd = rsqrtf((float)var_f64(sq * invK, mean) + 1e-6f);
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kb = kb0 + w + 4 * i;
        if (kb >= kb1) continue;
        float r[4] = {(v[i].x - mu) * rstd, (v[i].y - mu) * rstd, (v[i].z - mu) * rstd, (v[i].w - mu) * rstd};
        if (a.gamma) {
            r[0] = r[0] * g4[i].x + b4[i].x; r[1] = r[1] * g4[i].y + b4[i].y; r[2] = r[2] * g4[i].z + b4[i].z; r[3] = r[3] * g4[i].w + b4[i].w;
        }
        const float4 hv = make_float4(r[0] * (1.0f + sc[i].x) + sh[i].x, r[1] * (1.0f + sc[i].y) + sh[i].y,
                                      r[2] * (1.0f + sc[i].z) + sh[i].z, r[3] * (1.0f + sc[i].w) + sh[i].w);
        if (a.hq) bx_store_planes4(a.hq, a.MT, kb, half, mt, lane & 31, hv);
        else a.h[((long long)kb * a.MT + mt) * 64 + lane] = hv;
    }
}

// ------------------------------------------------------------------ gated residual + adaLN modulation in ONE launch
// x' = x + gate * (bias + sum_s slab[s])  (k_resid_stats), then  h = LayerNorm(x'; eps 1e-6) * (1 + scale) + shift  (k_modulate) of the
// SAME rows: the modulation needs the statistics of whole rows, i.e. of all n_chunks workgroups of a row tile.  They are tiny (32 rows
// x 16 bytes per workgroup), so instead of a second launch (~5 us in the captured step, twice per block) every workgroup publishes
// its partial sums with agent-scope 8-byte atomic stores into a buffer of its own launch site that a memset node poisoned (all ones:
// a NaN no sum can produce) at the start of the position -- the data is the flag -- and polls the other chunks' words of its row tile
// with agent-scope loads until none is poison (at most 64 x 4 workgroups of 256 threads: all resident), then modulates its OWN chunk,
// still in registers.  Summation order is the chunk order (fixed): bit-identical to the two-launch path.
struct ResidModArgs {
    ResidArgs r;                                            // (its row-major gate fields are not used here)
    const float* gamma; const float* beta;                 // nullable: no affine (FinalLayer.norm_final)
    const float4* modp; int MTm; long long kb_gate, kb_shift, kb_scale;   // packed modulations (see ModArgs)
    const float* gate_u; const float* shift_u; const float* scale_u; long long mod_stride; int split;   // nullable: shared rows (see ModArgs)
    float4* h; u32x4* hq;                                   // output: packed fp32 rows, or bf16 pieces (a k_bx GEMM follows)
    unsigned long long* part;                               // [n_chunks][Mpad][2] fp64 bit patterns of THIS launch site, poisoned (all ones) at the start of the position
    unsigned* fail;                                         // set if a wait gives up (never on a healthy device)
};

// KPW = k-blocks per wave: 1 for the usual widths (a chunk = 4 k-blocks: 4 x as many workgroups as k_resid_stats' 16-block chunks -- the
// launch is bound by each workgroup's read of its slab chunk from the other XCDs' L2 / the memory-side cache), 4 beyond 2048 features.
// PHASE 0: the fused launch described above.  PHASE 1 / 2: the same work as TWO launches, with the kernel boundary in the place of the
// in-launch wait -- 1 folds, stores x' and publishes the partial sums; 2 re-reads its x' chunk and the (by then complete) sums and
// modulates.  Same arithmetic and summation order: bit-identical results.  The engine falls back to this pair when a wait of the
// fused launch gave up (rar_verify): a device that does not hold the whole grid at once.
constexpr unsigned RM_MAX_POLLS = 1u << 16;      // ~50 ms of polling against microseconds on a healthy device
template <int S, int KPW, int PHASE = 0>
__global__ __launch_bounds__(256) void k_resid_mod(ResidModArgs a) {
    __shared__ double red[4][64][2];
    __shared__ double grp[8][32][2];
    __shared__ float s_mu[32], s_rs[32];
    const ResidArgs& r = a.r;
    const int c = blockIdx.x / r.MT, mt = blockIdx.x % r.MT;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kb0 = (int)((long long)c * r.KB / r.n_chunks), kb1 = (int)((long long)(c + 1) * r.KB / r.n_chunks);
    const int m = mt * 32 + (lane & 31), half = lane >> 5;
    const int Mpad = r.MT * 32;
    const bool shared = a.shift_u && m >= a.split;
    const long long urow = (long long)(*r.pos_dev) * a.mod_stride;
    // a wait that gave up anywhere (this launch or an earlier one of the call) ends every later wait at once: the call is repeated on
    // the two-launch path, nobody spins through the rest of the captured loop.  (Requested here, with the operands.)
    const unsigned fail0 = PHASE == 0 ? __hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    // every operand of both phases is requested before the first use
    float4 v[KPW], bb[KPW], sl[KPW][S], gg[KPW], g4[KPW], b4[KPW], sc[KPW], sh[KPW];
    int kbs[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kb = kb0 + w + 4 * i;
        kbs[i] = kb < kb1 ? kb : -1;
        const int kk = kb < kb1 ? kb : kb0;
        const long long idx = ((long long)kk * r.MT + mt) * 64 + lane;
        const int k = kk * 8 + 4 * half;
        v[i] = r.x[idx];
        if (PHASE != 2) {
            bb[i] = *(const float4*)(r.bias + k);
#pragma unroll
            for (int si = 0; si < S; ++si) sl[i][si] = r.slabs[(long long)si * r.slab_stride + idx];
        }
        g4[i] = a.gamma ? *(const float4*)(a.gamma + k) : make_float4(1.f, 1.f, 1.f, 1.f);
        b4[i] = a.gamma ? *(const float4*)(a.beta + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (shared) {
            gg[i] = *(const float4*)(a.gate_u + urow + k);
            sc[i] = *(const float4*)(a.scale_u + urow + k);
            sh[i] = *(const float4*)(a.shift_u + urow + k);
        } else {
            gg[i] = a.modp[((a.kb_gate + kk) * a.MTm + mt) * 64 + lane];
            sc[i] = a.modp[((a.kb_scale + kk) * a.MTm + mt) * 64 + lane];
            sh[i] = a.modp[((a.kb_shift + kk) * a.MTm + mt) * 64 + lane];
        }
    }
    double s = 0.0, ss = 0.0;
    float4 rv[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kbs[i] < 0) continue;
        if (PHASE == 2) { rv[i] = v[i]; continue; }      // x' as phase 1 stored it
        float4 acc = sl[i][0];
#pragma unroll
        for (int si = 1; si < S; ++si) { acc.x += sl[i][si].x; acc.y += sl[i][si].y; acc.z += sl[i][si].z; acc.w += sl[i][si].w; }
        const float4 x = make_float4(v[i].x + gg[i].x * (bb[i].x + acc.x), v[i].y + gg[i].y * (bb[i].y + acc.y),
                                     v[i].z + gg[i].z * (bb[i].z + acc.z), v[i].w + gg[i].w * (bb[i].w + acc.w));
        rv[i] = x;
        r.x[((long long)kbs[i] * r.MT + mt) * 64 + lane] = x;
        s += (double)x.x + (double)x.y + (double)x.z + (double)x.w;
        ss += sq4_f64(x);         // never a v_fmac_f64 chain: common.h
    }
    constexpr unsigned long long POISON = ~0ull;
    if (PHASE != 2) {
    red[w][lane][0] = s; red[w][lane][1] = ss;       // all 64 lanes, no shuffle (decoder_kernels.h, k_qkvx_bx's keeper reduction)
    __syncthreads();
    if (w == 0 && lane < 32) {
        double ts = 0, tss = 0;
        for (int i = 0; i < 4; ++i) { ts += red[i][lane][0] + red[i][lane + 32][0]; tss += red[i][lane][1] + red[i][lane + 32][1]; }
        unsigned long long* o = a.part + ((long long)c * Mpad + mt * 32 + lane) * 2;
        __hip_atomic_store(o, (unsigned long long)__double_as_longlong(ts), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, (unsigned long long)__double_as_longlong(tss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PHASE == 1) return;
    }
    {
        // row statistics of this tile's rows: thread (row = lane % 32, group = 2 w + lane / 32) re-reads the words of chunks group,
        // group + 8, ... (its own chunk's included: an L2 round trip) until none is poison; group sums in chunk order, then the eight
        // groups in group order: a fixed summation order
        const int row = lane & 31, gq = 2 * w + (lane >> 5);
        unsigned long long t0[8], t1[8];
        unsigned spins = 0;
        bool dead = fail0 != 0u;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cc = gq + 8 * i;
                if (cc < r.n_chunks) {
                    const unsigned long long* q = a.part + ((long long)cc * Mpad + mt * 32 + row) * 2;
                    t0[i] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t1[i] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && t0[i] != POISON && t1[i] != POISON;
                }
            }
            if (__all(ok) || dead) break;
            if (PHASE == 2) { __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }    // behind a kernel boundary nothing can be missing
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if ((spins & 255u) == 0u) dead = __hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (spins > RM_MAX_POLLS) { __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        double sm = 0, sq = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (gq + 8 * i < r.n_chunks) { sm += __longlong_as_double((long long)t0[i]); sq += __longlong_as_double((long long)t1[i]); }
        grp[gq][row][0] = sm; grp[gq][row][1] = sq;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double sm = 0, sq = 0;
        for (int q = 0; q < 8; ++q) { sm += grp[q][threadIdx.x][0]; sq += grp[q][threadIdx.x][1]; }
        const double invK = inv_count_f64((double)r.K);
        const double mean = sm * invK;
        s_mu[threadIdx.x] = (float)mean;
        s_rs[threadIdx.x] = rsqrtf((float)var_f64(sq * invK, mean) + 1e-6f);
    }
    __syncthreads();
    const float mu = s_mu[lane & 31], rstd = s_rs[lane & 31];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        if (kbs[i] < 0) continue;
        float q[4] = {(rv[i].x - mu) * rstd, (rv[i].y - mu) * rstd, (rv[i].z - mu) * rstd, (rv[i].w - mu) * rstd};
        if (a.gamma) {
            q[0] = q[0] * g4[i].x + b4[i].x; q[1] = q[1] * g4[i].y + b4[i].y; q[2] = q[2] * g4[i].z + b4[i].z; q[3] = q[3] * g4[i].w + b4[i].w;
        }
        const float4 hv = make_float4(q[0] * (1.0f + sc[i].x) + sh[i].x, q[1] * (1.0f + sc[i].y) + sh[i].y,
                                      q[2] * (1.0f + sc[i].z) + sh[i].z, q[3] * (1.0f + sc[i].w) + sh[i].w);
        if (a.hq) bx_store_planes4(a.hq, r.MT, kbs[i], half, mt, lane & 31, hv);
        else a.h[((long long)kbs[i] * r.MT + mt) * 64 + lane] = hv;
    }
}

template <int PHASE>
static int launch_resid_mod_phase(const ResidModArgs& a, int grid, int kpw, hipStream_t st) {
    switch (a.r.S) {
#define WMAR_RM_CASE(N) case N: if (kpw == 1) hipLaunchKernelGGL((k_resid_mod<N, 1, PHASE>), dim3(grid), dim3(256), 0, st, a); \
                                else hipLaunchKernelGGL((k_resid_mod<N, 4, PHASE>), dim3(grid), dim3(256), 0, st, a); break;
        WMAR_RM_CASE(1) WMAR_RM_CASE(2) WMAR_RM_CASE(3) WMAR_RM_CASE(4) WMAR_RM_CASE(5) WMAR_RM_CASE(6) WMAR_RM_CASE(7) WMAR_RM_CASE(8)
#undef WMAR_RM_CASE
        default: set_error("resid_mod: bad slab count %d", a.r.S); return WMAR_EINVAL;
    }
    return launch_status("k_resid_mod");
}
// fused: one launch with the in-launch wait; otherwise the two-launch pair
static int launch_resid_mod(const ResidModArgs& a, int grid, int kpw, bool fused, hipStream_t st) {
    if (fused) return launch_resid_mod_phase<0>(a, grid, kpw, st);
    if (int rc = launch_resid_mod_phase<1>(a, grid, kpw, st)) return rc;
    return launch_resid_mod_phase<2>(a, grid, kpw, st);
}

// SiLU(emb[cond] + timesteps[p]) for p = 0..T-1 in packed layout: the adaLN input of a row whose
// condition never changes (the unconditional half under guidance).
static __global__ void k_rar_cond_rows(float4* sc, const float* emb, const float* tstep, long long cond, int T, int MT, int K) {
    const int kb = blockIdx.x / MT, mt = blockIdx.x % MT;
    const int lane = threadIdx.x, p = mt * 32 + (lane & 31), half = lane >> 5;
    const int k = kb * 8 + 4 * half;
    float cv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p < T) {
        const float4 ce = *(const float4*)(emb + cond * K + k), te = *(const float4*)(tstep + (long long)p * K + k);
        cv[0] = ce.x + te.x; cv[1] = ce.y + te.y; cv[2] = ce.z + te.z; cv[3] = ce.w + te.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) cv[j] = cv[j] / (1.0f + expf(-cv[j]));
    }
    sc[((long long)kb * MT + mt) * 64 + lane] = make_float4(cv[0], cv[1], cv[2], cv[3]);
}

// tok_out[b] replicated to the unconditional half happens implicitly: ids are shared by b % B.
static __global__ void k_set3(int* p, int a, int b, int c) { p[0] = a; p[1] = b; p[2] = c; }

}  // namespace wmar

using namespace wmar;

// split of a k_bx GEMM: K = 64 S PER with S <= MAX_SLABS slabs, as many of the `tiles` x S workgroups as fit one per CU
struct BxShape { int S = 0, PER = 0; };
static BxShape bx_shape(int tiles, int KU) {
    BxShape best{};
    const int pers[] = {4, 5, 8, 10, 16, 20};
    for (int per : pers) {
        if (KU % (4 * per)) continue;
        const int S = KU / (4 * per);
        if (S < 1 || S > MAX_SLABS || tiles * S > 256) continue;
        if (tiles * S > tiles * best.S) best = BxShape{S, per};
    }
    return best;
}
template <int NT>
static int launch_bx4(const BxArgs& a, int N, int PER, hipStream_t st) {
    switch (PER) {
        case 4: return launch_bx<NT, 4, 4>(a, N, st);
        case 5: return launch_bx<NT, 5, 4>(a, N, st);
        case 8: return launch_bx<NT, 8, 4>(a, N, st);
        case 10: return launch_bx<NT, 10, 4>(a, N, st);
        case 16: return launch_bx<NT, 16, 4>(a, N, st);
        case 20: return launch_bx<NT, 20, 4>(a, N, st);
        default: set_error("k_bx: %d steps per wave unsupported", PER); return WMAR_EINVAL;
    }
}

struct RarLayer {
    float4 *wqkv, *wproj, *wfc1, *wfc2;
    float4 *wqkv_bx = nullptr, *wproj_bx = nullptr, *wfc2_bx = nullptr, *wfc1_bx = nullptr;   // k_pack_bx order (128-row steps on the bf16 matrix pipe)
    float *bqkv, *bproj, *bfc1, *bfc2, *n1w, *n1b, *n2w, *n2b, *qnw, *qnb, *knw, *knb;
};

struct wmar_rar {
    wmar_rar_config cfg{};
    DeviceArena mem;
    int D = 0, H = 0, hd = 0, V = 0, L = 0, F = 0, T = 0, Bmax = 0, Mmax = 0, MTmax = 0;
    long long Ntot = 0;
    std::vector<RarLayer> layers;
    float *emb = nullptr, *cls = nullptr, *pos = nullptr, *tape = nullptr, *tstep = nullptr;
    float4 *wada = nullptr, *whead = nullptr;
    float4* wada_bx = nullptr;     // the adaLN weights in k_pack_bx order (64 conditional rows: the GEMM runs as k_bx<4, 20, 2>)
    u32x4* scq = nullptr;          // SiLU(c) of the conditional rows as bf16 pieces
    bool bx_ada = false;
    float *bada = nullptr, *bhead = nullptr;
    // workspaces
    float4 *x = nullptr, *h = nullptr, *y = nullptr, *hbuf = nullptr, *sc = nullptr, *slabs = nullptr, *qkv_slabs = nullptr;
    // 128-row steps (guidance at batch 33..64): QKV, proj and FC2 as k_bx on bf16 pieces; (K slices, 16-k steps per wave) per GEMM
    u32x4 *xq = nullptr, *yq = nullptr, *hq = nullptr;
    BxShape bx_qkv, bx_proj, bx_fc2;
    bool bx_ok = false, no_bx = false;
    bool bx_fc1 = false;    // FC1 + bias + GELU as k_bx with the whole K per workgroup (K / 64 steps per wave: 20 at 1280)
    float* mod = nullptr;
    float* mod_u = nullptr;   // [T][Ntot] modulations of the unconditional row at every position (built on first guided generate)
    bool mod_u_ready = false;
    double* stats = nullptr;
    float *kcache = nullptr, *vcache = nullptr, *logits = nullptr, *scratch = nullptr, *cfg_scale = nullptr;
    long long *ids = nullptr, *cond_ids = nullptr;
    int* ctr = nullptr;   // [pos, step, len]
    unsigned long long* part = nullptr;   // [2 L launch sites][STAT_CHUNKS_MAX][Mpad][2]: k_resid_mod's published partial sums, poisoned at the start of every position
    size_t part_site = 0;                 // words per site
    unsigned* sync_fail = nullptr;
    bool rm_fused = true;          // k_resid_mod as ONE launch with an in-launch wait; false (WMAR_NO_XR=1, an occupancy short of the grid, or a
                                   // wait that gave up): the two-launch pair
    int inject_fail = 0;           // WMAR_INJECT_SYNC_FAIL=1 at creation (tests): the next fused call finds the flag raised
    int fallbacks = 0;
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool pending = false;
    void drop_graph() {
        if (pending && ev) (void)hipEventSynchronize(ev);
        pending = false;
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr;
    }
    template <typename Tp>
    int alloc(Tp** p, size_t n) { return mem.alloc(p, n); }
    ~wmar_rar() {
        drop_graph();
        mem.release();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (ev) (void)hipEventDestroy(ev);
    }
};

namespace {

struct RarPlan {
    wmar_rar* g;
    int M, Bhalf;           // rows this call (2B with guidance), B
    hipStream_t st;
    int MT, D, KBD, KBF, nch;
    long long act;
    int S_proj, S_fc2;
    const long long* tok;   // explicit tokens (forward_position) or null
    bool shared_u;          // rows [Bhalf, M) all carry the "none" condition: their adaLN modulation comes from g->mod_u
    int MTc;                // row tiles of the adaLN GEMM
    bool bx = false;        // 128 rows: QKV / proj / FC2 on the bf16 matrix pipe

    RarPlan(wmar_rar* g_, int M_, int Bhalf_, const long long* tok_, hipStream_t st_, bool shared_u_ = false)
        : g(g_), M(M_), Bhalf(Bhalf_), st(st_), tok(tok_), shared_u(shared_u_) {
        MT = mt_for(M); D = g->D; KBD = D / 8; KBF = g->F / 8; nch = stat_chunks(KBD);
        MTc = shared_u ? mt_for(Bhalf) : MT;
        act = (long long)KBD * MT * 64;
        const int tiles = MT % 2 == 0 ? (D / 32) * (MT / 2) : (D / 32) * MT;
        S_proj = pick_split(tiles, KBD, 4);
        S_fc2 = pick_split(tiles, KBF, 4);
        bx = MT == 4 && g->bx_ok && !g->no_bx;
        if (bx) { S_proj = g->bx_proj.S; S_fc2 = g->bx_fc2.S; }
    }
    GemmArgs base() const {
        GemmArgs a{};
        a.MT = MT; a.B = M; a.stats = g->stats; a.n_chunks = nch; a.K = D;
        a.pos_dev = g->ctr; a.D = D; a.H = g->H; a.hd = g->hd; a.Tmax = g->T;
        return a;
    }
    int embed() {
        RarEmbedArgs e{};
        e.x = g->x; e.sc = g->sc; e.stats = g->stats; e.emb = g->emb; e.cls = g->cls; e.pos = g->pos; e.tape = g->tape;
        e.tstep = g->tstep; e.tok = tok; e.cond = g->cond_ids; e.ids = g->ids; e.ids_stride = g->cfg.image_seq_len;
        e.pos_dev = g->ctr; e.KB = KBD; e.MT = MT; e.n_chunks = nch; e.M = M; e.Bhalf = Bhalf; e.K = D; e.MTsc = MTc;
        e.scq = ada_bx() ? g->scq : nullptr;
        hipLaunchKernelGGL(k_rar_embed, dim3(nch * MT), dim3(256), 0, st, e);
        return launch_status("k_rar_embed");
    }
    // all adaLN modulations of this position: mod[M][Ntot] = SiLU(c) W_ada^T + b_ada
    // 33..64 adaLN rows at hidden size 1280: on the bf16 pipe (round 5)
    bool ada_bx() const { return g->bx_ada && MTc == 2; }
    int adaln() {
        if (ada_bx()) {
            // k_bx<4, 20, 2>: 128 columns x the whole K per workgroup (1940 workgroups), its four waves a K quarter each; weights fp32
            // (split in registers), SiLU(c) as pieces from k_rar_embed, bias in the epilogue, output in the packed layout the
            // modulation consumers read.  The fp32-input MFMA GEMM it replaces was bound by that pipe (259 us at peak for
            // 40.7 GFLOP, 381 measured); here the 1.27 GB weight stream is (159 us at 8 TB/s).
            BxArgs x{};
            x.Wq = g->wada_bx; x.Xq = g->scq; x.out = (float4*)g->mod; x.slab_stride = 0; x.KU = D / 16; x.S = 1; x.bias = g->bada;
            return launch_bx<4, 20, 2, false>(x, (int)g->Ntot, st);
        }
        GemmArgs a = base();
        a.Wp = g->wada; a.Xp = g->sc; a.KB = KBD; a.NT = (int)(g->Ntot / 32); a.bias = g->bada;
        a.out_packed = (float4*)g->mod; a.slab_stride = 0;        // packed [Ntot/8][MTc][64]: read like an activation by k_modulate / k_resid_mod
        if (shared_u) { a.MT = MTc; a.B = Bhalf; }
        // (round 4: four column tiles per workgroup -- a quarter of the activation re-reads -- is 0.4 % SLOWER per step: at 1 wave per SIMD
        // the fp32-MFMA-bound launch, 259 us at peak, loses more latency hiding than the bytes buy)
        return gemm_dispatch<EPI_PACKED, false>(a, false, st);
    }
    int modulate(const float* gamma, const float* beta, long long off_shift, long long off_scale, u32x4* planes = nullptr) {
        ModArgs m{};
        m.hq = planes;
        m.x = g->x; m.h = g->h; m.stats = g->stats; m.gamma = gamma; m.beta = beta;
        m.modp = (const float4*)g->mod; m.MTm = MTc; m.kb_shift = off_shift / 8; m.kb_scale = off_scale / 8; m.mod_stride = g->Ntot;
        if (shared_u) { m.shift_u = g->mod_u + off_shift; m.scale_u = g->mod_u + off_scale; m.split = Bhalf; }
        m.pos_dev = g->ctr;
        m.KB = KBD; m.MT = MT; m.n_chunks = nch; m.K = D;
        hipLaunchKernelGGL(k_modulate, dim3(nch * MT), dim3(256), 0, st, m);
        return launch_status("k_modulate");
    }
    // gated residual fold of the S slabs in g->slabs + the adaLN modulation of the new rows for the NEXT GEMM (one launch)
    int resid_mod(int site, const float* bias, int S, long long off_gate, const float* gamma, const float* beta, long long off_shift,
                  long long off_scale, u32x4* planes) {
        ResidModArgs a{};
        ResidArgs& r = a.r;
        const int kpw = KBD <= 256 ? 1 : 4, nrm = (KBD + 4 * kpw - 1) / (4 * kpw);      // chunks of 4 (or 16) k-blocks: <= 64 per row tile
        r.x = g->x; r.stats = g->stats; r.KB = KBD; r.MT = MT; r.n_chunks = nrm; r.B = M; r.K = D;
        r.slabs = g->slabs; r.slab_stride = act; r.S = S; r.bias = bias;
        r.pos_dev = g->ctr;
        a.gamma = gamma; a.beta = beta; a.modp = (const float4*)g->mod; a.MTm = MTc;
        a.kb_gate = off_gate / 8; a.kb_shift = off_shift / 8; a.kb_scale = off_scale / 8; a.mod_stride = g->Ntot;
        if (shared_u) { a.gate_u = g->mod_u + off_gate; a.shift_u = g->mod_u + off_shift; a.scale_u = g->mod_u + off_scale; a.split = Bhalf; }
        a.h = g->h; a.hq = planes;
        a.part = g->part + (size_t)site * site_words(); a.fail = g->sync_fail;
        return launch_resid_mod(a, nrm * MT, kpw, g->rm_fused, st);
    }
    // words of one k_resid_mod launch site at THIS plan's row count: [chunks][32 MT][2]; the sites are packed at this size, so the
    // poison memset of a position covers exactly what the launches poll
    size_t site_words() const { const int kpw = KBD <= 256 ? 1 : 4; return (size_t)((KBD + 4 * kpw - 1) / (4 * kpw)) * MT * 32 * 2; }
    static bool att80() { static int v = -1; if (v < 0) { const char* e = getenv("WMAR_RAR_ATT80"); v = (e && atoi(e) == 0) ? 0 : 1; } return v != 0; }
    bool fc1_bx() const {
        return bx && g->bx_fc1
#ifdef WMAR_DEV_KNOBS
               && !getenv("WMAR_RAR_NO_FC1_BX")
#endif
            ;
    }
    // The block's input rows arrive already modulated for the QKV projection (planes in g->xq when bx, else fp32 in g->h): by the
    // position's first k_modulate (block 0) or by the previous block's second k_resid_mod.
    int layer(int l) {
        const RarLayer& w = g->layers[l];
        const long long o = (long long)l * 6 * D;   // chunk(6): shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        int rc, S = 1;
        int S_qkv = 1;
        bool qkv_rm = false;
        if (bx) {
            BxArgs x{};
            x.Wq = w.wqkv_bx; x.Xq = g->xq; x.out = g->qkv_slabs; x.slab_stride = 3 * act; x.KU = D / 16; x.S = g->bx_qkv.S;
            // head_dim 80: the pieces go out row-major, so that the attention prologue of (row, head) reads 320 contiguous bytes per piece
            // and q / k / v instead of 20 lines of the packed layout (k_attn_decode80 takes either; WMAR_RAR_QKV_PACKED=1: A/B)
            static const bool qkv_packed = getenv("WMAR_RAR_QKV_PACKED") != nullptr;
            qkv_rm = g->hd == 80 && att80() && !qkv_packed;
            x.rowmajor = qkv_rm ? 1 : 0; x.N = 3 * D;
            if ((rc = launch_bx4<2>(x, 3 * D, g->bx_qkv.PER, st))) return rc;     // 64-column groups: half the activation reads per column
            S_qkv = g->bx_qkv.S;
        } else {
            GemmArgs a = base();
            a.Wp = w.wqkv; a.Xp = g->h; a.KB = KBD; a.NT = 3 * D / 32; a.out_packed = g->qkv_slabs; a.slab_stride = 3 * act;
            if ((rc = gemm_split(a, &S, st, 1))) return rc;
        }
        AttnArgs t{};
        const long long lstride = (long long)g->Mmax * g->H * g->T * g->hd;
        t.qkv_slabs = g->qkv_slabs; t.slab_stride = 3 * act; t.S = S_qkv; t.stats = g->stats; t.n_chunks = nch; t.K = D; t.invK = 1.0 / (double)D;
        t.rowmajor = qkv_rm ? 1 : 0;
        t.bias = w.bqkv; t.mode = 1; t.qn_w = w.qnw; t.qn_b = w.qnb; t.kn_w = w.knw; t.kn_b = w.knb;
        t.kcache = g->kcache + l * lstride; t.vcache = g->vcache + l * lstride; t.y = g->y; t.yq = bx ? g->yq : nullptr; t.pos_dev = g->ctr;
        t.D = D; t.H = g->H; t.Tmax = g->T; t.MT = MT; t.scale = 1.0f / sqrtf((float)g->hd);
        const dim3 grid((unsigned)(M * g->H));
        int nwa = 1;      // measured at 128 rows, head_dim 80, 256 positions: 4.58 / 4.75 / 4.83 ms per step with 1 / 2 / 4 waves per (sequence, head)
#ifdef WMAR_DEV_KNOBS
        { const char* e = getenv("WMAR_RAR_ATT_NW"); if (e) nwa = atoi(e); }
#endif
#define WMAR_RAR_ATT(HDV)                                                                                   \
        if (nwa == 1) hipLaunchKernelGGL((k_attn_decode<HDV, 1, false, 1, false>), grid, dim3(64), 0, st, t);                \
        else if (nwa == 4) hipLaunchKernelGGL((k_attn_decode<HDV, 4, false, 1, false>), grid, dim3(256), 0, st, t);          \
        else hipLaunchKernelGGL((k_attn_decode<HDV, 2, false, 1, false>), grid, dim3(128), 0, st, t);
        switch (g->hd) {
            case 32: WMAR_RAR_ATT(32) break;
            case 48: WMAR_RAR_ATT(48) break;
            case 64: WMAR_RAR_ATT(64) break;
            case 80:
                // all 64 lanes busy (k_attn_decode80); WMAR_RAR_ATT80=0 at run time keeps the 20-of-32-lane form (A/B)
                if (att80()) {
                    if (nwa == 1) hipLaunchKernelGGL((k_attn_decode80<1>), grid, dim3(64), 0, st, t);
                    else if (nwa == 4) hipLaunchKernelGGL((k_attn_decode80<4>), grid, dim3(256), 0, st, t);
                    else hipLaunchKernelGGL((k_attn_decode80<2>), grid, dim3(128), 0, st, t);
                } else { WMAR_RAR_ATT(80) }
                break;
            case 88: WMAR_RAR_ATT(88) break;
            default: WMAR_RAR_ATT(128) break;
        }
#undef WMAR_RAR_ATT
        if ((rc = launch_status("k_attn_decode"))) return rc;
        if (bx) {
            BxArgs x{};
            x.Wq = w.wproj_bx; x.Xq = g->yq; x.out = g->slabs; x.slab_stride = act; x.KU = D / 16; x.S = g->bx_proj.S;
            if ((rc = launch_bx4<1>(x, D, g->bx_proj.PER, st))) return rc;
        } else {
            GemmArgs p = base();
            p.Wp = w.wproj; p.Xp = g->y; p.KB = KBD; p.NT = D / 32; p.out_packed = g->slabs; p.slab_stride = act;
            if ((rc = gemm_split(p, &S, st, S_proj))) return rc;
        }
        const bool fc1x = fc1_bx();
        // round 5: eight waves x 10 steps instead of four x 20 (WMAR_RAR_FC1_NW4=1: the round-2 form, A/B); the operand as packed fp32
        // rows split in registers instead of bf16 planes (two thirds of the activation bytes; WMAR_RAR_FC1_PLANES=1: A/B)
        static const bool nw4 = getenv("WMAR_RAR_FC1_NW4") != nullptr;
        static const bool fc1_planes = getenv("WMAR_RAR_FC1_PLANES") != nullptr;
        const bool fc1_xf = fc1x && !nw4 && !fc1_planes;
        // x += gate_msa * (proj + bias);  norm2 + modulate(shift_mlp, scale_mlp) -> the FC1 operand
        if ((rc = resid_mod(2 * l, w.bproj, S_proj, o + 2 * D, w.n2w, w.n2b, o + 3 * D, o + 4 * D, (fc1x && !fc1_xf) ? g->xq : nullptr))) return rc;
        if (fc1x) {
            BxArgs x{};
            x.Wq = w.wfc1_bx; x.Xq = fc1_xf ? (const u32x4*)g->h : g->xq; x.KU = D / 16; x.S = 1; x.bias = w.bfc1; x.outq = g->hq;
            if (nw4) { if ((rc = launch_bx<1, 20, 4, true>(x, g->F, st))) return rc; }
            else if (fc1_xf) { if ((rc = launch_bx<1, 10, 4, true, 8, true>(x, g->F, st))) return rc; }
            else if ((rc = launch_bx<1, 10, 4, true, 8>(x, g->F, st))) return rc;
        } else {
            GemmArgs f = base();
            f.Wp = w.wfc1; f.Xp = g->h; f.bias = w.bfc1; f.KB = KBD; f.NT = g->F / 32; f.out_packed = g->hbuf;
            f.out_planes = bx ? g->hq : nullptr;
            if ((rc = gemm_dispatch<EPI_GELU, false>(f, false, st))) return rc;
        }
        if (bx) {
            BxArgs x{};
            x.Wq = w.wfc2_bx; x.Xq = g->hq; x.out = g->slabs; x.slab_stride = act; x.KU = g->F / 16; x.S = g->bx_fc2.S;
            if ((rc = launch_bx4<2>(x, D, g->bx_fc2.PER, st))) return rc;
        } else {
            GemmArgs q = base();
            q.Wp = w.wfc2; q.Xp = g->hbuf; q.KB = KBF; q.NT = D / 32; q.out_packed = g->slabs; q.slab_stride = act;
            if ((rc = gemm_split(q, &S, st, S_fc2))) return rc;
        }
        // x += gate_mlp * (FC2 + bias);  then the NEXT consumer's modulation: block l+1's norm1 (shift_msa, scale_msa), or the
        // FinalLayer's affine-free norm (scale first, then shift: rar.py:132) into fp32 rows for the vocabulary head
        if (l + 1 < g->L) {
            const RarLayer& nx = g->layers[l + 1];
            const long long o2 = o + 6 * D;
            return resid_mod(2 * l + 1, w.bfc2, S_fc2, o + 5 * D, nx.n1w, nx.n1b, o2, o2 + D, bx ? g->xq : nullptr);
        }
        const long long of = (long long)g->L * 6 * D;
        return resid_mod(2 * l + 1, w.bfc2, S_fc2, o + 5 * D, nullptr, nullptr, of + D, of, nullptr);
    }
    int head(float* logits_out) {
        GemmArgs a = base();
        a.Wp = g->whead; a.Xp = g->h; a.KB = KBD; a.NT = g->V / 32; a.bias = g->bhead; a.logits = logits_out; a.V = g->V;
        return gemm_dispatch<EPI_LOGITS, false>(a, false, st);
    }
    int position(bool with_head, float* logits_out) {
        int rc;
        if (hipMemsetAsync(g->part, 0xff, (size_t)2 * g->L * site_words() * 8, st) != hipSuccess) {     // poison: "not published yet"
            set_error("rar position: poisoning the partial sums failed"); return WMAR_EHIP;
        }
        if ((rc = embed())) return rc;
        if ((rc = adaln())) return rc;
        { const RarLayer& w0 = g->layers[0]; if ((rc = modulate(w0.n1w, w0.n1b, 0, D, bx ? g->xq : nullptr))) return rc; }
        for (int l = 0; l < g->L; ++l)
            if ((rc = layer(l))) return rc;
        return with_head ? head(logits_out) : WMAR_OK;
    }
};

}  // namespace

extern "C" {

int wmar_rar_create(const wmar_rar_config* cfg, const char* const* names, const void* const* tensors_dev,
                    int32_t n_tensors, void* stream, wmar_rar** out) {
    WMAR_REQUIRE(cfg && names && tensors_dev && out, "rar_create: null argument");
    const int D = cfg->hidden_size, H = cfg->num_attention_heads, L = cfg->num_hidden_layers, F = cfg->intermediate_size;
    const int V = cfg->codebook_size;
    WMAR_REQUIRE(D % H == 0, "hidden_size %% heads != 0");
    const int hd = D / H;
    WMAR_REQUIRE(D % 32 == 0 && F % 32 == 0 && V % 32 == 0 && D <= 8192, "hidden (<=8192), intermediate and codebook sizes must be multiples of 32");
    WMAR_REQUIRE(hd == 32 || hd == 48 || hd == 64 || hd == 80 || hd == 88 || hd == 128, "head_dim %d unsupported", hd);
    WMAR_REQUIRE(cfg->max_batch >= 1 && cfg->max_batch <= 64, "max_batch must be in 1..64 (rows double under guidance)");
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) tm.m[names[i]] = tensors_dev[i];
    hipStream_t st = (hipStream_t)stream;
    auto* g = new wmar_rar();
    g->cfg = *cfg; g->D = D; g->H = H; g->hd = hd; g->V = V; g->L = L; g->F = F;
    g->T = cfg->image_seq_len + 2; g->Bmax = cfg->max_batch; g->Mmax = 2 * cfg->max_batch; g->MTmax = mt_for(g->Mmax);
    g->Ntot = (long long)6 * D * L + 2 * D;
    int rc = WMAR_OK;
    auto need = [&](const std::string& k) -> const float* {
        const float* p = tm.get(k);
        if (!p && rc == WMAR_OK) { set_error("checkpoint tensor '%s' is missing", k.c_str()); rc = WMAR_EMISSING; }
        return p;
    };
#define TRY(x) do { if (rc == WMAR_OK) rc = (x); } while (0)
    const int nemb = V + 1 + cfg->condition_num_classes + 1;
    const float *e = need("embeddings.weight"), *cl = need("cls_token"), *pe = need("pos_embed"),
                *ta = need("target_aware_pos_embed"), *ts = need("timesteps_embeddings"), *hw = need("lm_head.weight"),
                *hb = need("lm_head.bias"), *fw = need("adaln_before_head.adaLN_modulation.1.weight"),
                *fb = need("adaln_before_head.adaLN_modulation.1.bias");
    if (rc == WMAR_OK) {
        TRY(copy_vec(g, &g->emb, e, (size_t)nemb * D, st));
        TRY(copy_vec(g, &g->cls, cl, (size_t)D, st));
        TRY(copy_vec(g, &g->pos, pe, (size_t)(cfg->image_seq_len + 1024) * D, st));
        TRY(copy_vec(g, &g->tape, ta, (size_t)(cfg->image_seq_len + 1024) * D, st));
        TRY(copy_vec(g, &g->tstep, ts, (size_t)(cfg->image_seq_len + 100) * D, st));
        TRY(g->alloc(&g->whead, (size_t)V * D / 4));
        TRY(pack(hw, g->whead, V, D, 0, st));
        TRY(copy_vec(g, &g->bhead, hb, (size_t)V, st));
        TRY(g->alloc(&g->wada, (size_t)g->Ntot * D / 4));
        TRY(g->alloc(&g->bada, (size_t)g->Ntot));
        // WMAR_RAR_ADA_FP32=1 (A/B) keeps the per-step adaLN GEMM on the fp32-input MFMA
        g->bx_ada = D == 1280 && g->Ntot % 128 == 0 && g->MTmax >= 2 && getenv("WMAR_NO_BX") == nullptr && getenv("WMAR_RAR_ADA_FP32") == nullptr;
        if (g->bx_ada) { TRY(g->alloc(&g->wada_bx, (size_t)g->Ntot * D / 4)); TRY(g->alloc(&g->scq, (size_t)D / 16 * 2 * 3 * 64)); }
    }
    if (g->MTmax >= 4 && D % 32 == 0 && F % 32 == 0) {
        g->bx_qkv = bx_shape(3 * D / 64, D / 16); g->bx_proj = bx_shape(D / 32, D / 16); g->bx_fc2 = bx_shape(D / 64, F / 16);
        g->bx_fc1 = D == 1280 && F % 32 == 0 && F / 32 <= 256;
        g->bx_ok = D % 64 == 0 && g->bx_qkv.S > 0 && g->bx_qkv.S <= QKV_SLABS_MAX && g->bx_proj.S > 0 && g->bx_fc2.S > 0;
    }
    // WMAR_NO_BX=1 at engine creation (any build, as in gpt.hip): every GEMM stays on the fp32-input MFMA kernels
    g->no_bx = getenv("WMAR_NO_BX") != nullptr;
    g->layers.resize(L);
    for (int l = 0; l < L && rc == WMAR_OK; ++l) {
        const std::string p = "blocks." + std::to_string(l) + ".";
        RarLayer& w = g->layers[l];
        const float *qw = need(p + "attn.qkv.weight"), *qb = need(p + "attn.qkv.bias"), *pw = need(p + "attn.proj.weight"),
                    *pb = need(p + "attn.proj.bias"), *f1w = need(p + "mlp.fc1.weight"), *f1b = need(p + "mlp.fc1.bias"),
                    *f2w = need(p + "mlp.fc2.weight"), *f2b = need(p + "mlp.fc2.bias"), *n1w = need(p + "norm1.weight"),
                    *n1b = need(p + "norm1.bias"), *n2w = need(p + "norm2.weight"), *n2b = need(p + "norm2.bias"),
                    *qnw = need(p + "attn.q_norm.weight"), *qnb = need(p + "attn.q_norm.bias"),
                    *knw = need(p + "attn.k_norm.weight"), *knb = need(p + "attn.k_norm.bias"),
                    *aw = need(p + "adaLN_modulation.1.weight"), *ab = need(p + "adaLN_modulation.1.bias");
        if (rc != WMAR_OK) break;
        TRY(g->alloc(&w.wqkv, (size_t)3 * D * D / 4)); TRY(pack(qw, w.wqkv, 3 * D, D, 0, st));
        TRY(copy_vec(g, &w.bqkv, qb, (size_t)3 * D, st));
        TRY(g->alloc(&w.wproj, (size_t)D * D / 4)); TRY(pack(pw, w.wproj, D, D, 0, st));
        TRY(copy_vec(g, &w.bproj, pb, (size_t)D, st));
        TRY(g->alloc(&w.wfc1, (size_t)F * D / 4)); TRY(pack(f1w, w.wfc1, F, D, 0, st));
        TRY(copy_vec(g, &w.bfc1, f1b, (size_t)F, st));
        TRY(g->alloc(&w.wfc2, (size_t)F * D / 4)); TRY(pack(f2w, w.wfc2, D, F, 0, st));
        TRY(copy_vec(g, &w.bfc2, f2b, (size_t)D, st));
        if (g->bx_ok) {
            auto pack_bx = [&](const float* W, float4* Wq, int N, int K) -> int {
                const long long total = (long long)(N / 32) * (K / 16) * 128;
                hipLaunchKernelGGL(k_pack_bx, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, (const float*)nullptr, Wq, N, K, 0);
                return launch_status("k_pack_bx");
            };
            TRY(g->alloc(&w.wqkv_bx, (size_t)3 * D * D / 4)); TRY(pack_bx(qw, w.wqkv_bx, 3 * D, D));
            TRY(g->alloc(&w.wproj_bx, (size_t)D * D / 4)); TRY(pack_bx(pw, w.wproj_bx, D, D));
            TRY(g->alloc(&w.wfc2_bx, (size_t)F * D / 4)); TRY(pack_bx(f2w, w.wfc2_bx, D, F));
            if (g->bx_fc1) { TRY(g->alloc(&w.wfc1_bx, (size_t)F * D / 4)); TRY(pack_bx(f1w, w.wfc1_bx, F, D)); }
        }
        TRY(copy_vec(g, &w.n1w, n1w, (size_t)D, st)); TRY(copy_vec(g, &w.n1b, n1b, (size_t)D, st));
        TRY(copy_vec(g, &w.n2w, n2w, (size_t)D, st)); TRY(copy_vec(g, &w.n2b, n2b, (size_t)D, st));
        TRY(copy_vec(g, &w.qnw, qnw, (size_t)hd, st)); TRY(copy_vec(g, &w.qnb, qnb, (size_t)hd, st));
        TRY(copy_vec(g, &w.knw, knw, (size_t)hd, st)); TRY(copy_vec(g, &w.knb, knb, (size_t)hd, st));
        // this block's 6d adaLN rows go to their slice of the one big modulation GEMM
        TRY(pack(aw, g->wada, 6 * D, D, l * (6 * D / 32), st));
        if (g->bx_ada && rc == WMAR_OK) {
            const long long total = (long long)(6 * D / 32) * (D / 16) * 128;
            hipLaunchKernelGGL(k_pack_bx, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, aw, (const float*)nullptr, g->wada_bx, 6 * D, D, l * (6 * D / 32));
            rc = launch_status("k_pack_bx");
        }
        if (rc == WMAR_OK && hipMemcpyAsync(g->bada + (size_t)l * 6 * D, ab, (size_t)6 * D * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("adaLN bias copy failed"); rc = WMAR_EHIP;
        }
    }
    if (rc == WMAR_OK) {
        TRY(pack(fw, g->wada, 2 * D, D, L * (6 * D / 32), st));
        if (g->bx_ada && rc == WMAR_OK) {
            const long long total = (long long)(2 * D / 32) * (D / 16) * 128;
            hipLaunchKernelGGL(k_pack_bx, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, fw, (const float*)nullptr, g->wada_bx, 2 * D, D, L * (6 * D / 32));
            rc = launch_status("k_pack_bx");
        }
        if (rc == WMAR_OK && hipMemcpyAsync(g->bada + (size_t)L * 6 * D, fb, (size_t)2 * D * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("final adaLN bias copy failed"); rc = WMAR_EHIP;
        }
    }
    const size_t Mpad = (size_t)g->MTmax * 32;
    TRY(g->alloc(&g->x, Mpad * D / 4));
    TRY(g->alloc(&g->h, Mpad * D / 4));
    TRY(g->alloc(&g->y, Mpad * D / 4));
    TRY(g->alloc(&g->sc, Mpad * D / 4));
    TRY(g->alloc(&g->hbuf, Mpad * F / 4));
    TRY(g->alloc(&g->slabs, (size_t)MAX_SLABS * Mpad * D / 4));
    TRY(g->alloc(&g->qkv_slabs, (size_t)(g->bx_ok ? g->bx_qkv.S : 1) * Mpad * 3 * D / 4));
    if (g->bx_ok) {
        TRY(g->alloc(&g->xq, (size_t)D / 16 * 4 * 3 * 64));
        TRY(g->alloc(&g->yq, (size_t)D / 16 * 4 * 3 * 64));
        TRY(g->alloc(&g->hq, (size_t)F / 16 * 4 * 3 * 64));
        if (rc == WMAR_OK && hipMemsetAsync(g->yq, 0, (size_t)D / 16 * 4 * 3 * 64 * 16, st) != hipSuccess) { set_error("rar_create: memset failed"); rc = WMAR_EHIP; }
    }
    TRY(g->alloc(&g->mod, Mpad * (size_t)g->Ntot));
    TRY(g->alloc(&g->mod_u, (size_t)((g->T + 31) / 32) * 32 * (size_t)g->Ntot));
    TRY(g->alloc(&g->stats, (size_t)STAT_CHUNKS_MAX * Mpad * 2));
    const size_t kv = (size_t)L * g->Mmax * H * g->T * hd;
    TRY(g->alloc(&g->kcache, kv));
    TRY(g->alloc(&g->vcache, kv));
    TRY(g->alloc(&g->logits, (size_t)g->Mmax * V));
    TRY(g->alloc(&g->scratch, (size_t)g->Bmax * V));
    TRY(g->alloc(&g->cfg_scale, (size_t)cfg->image_seq_len));
    TRY(g->alloc(&g->ids, (size_t)g->Bmax * cfg->image_seq_len));
    TRY(g->alloc(&g->cond_ids, (size_t)g->Mmax));
    TRY(g->alloc(&g->ctr, 4));
    { const int kbd = D / 8, kpw = kbd <= 256 ? 1 : 4; g->part_site = (size_t)((kbd + 4 * kpw - 1) / (4 * kpw)) * g->MTmax * 32 * 2; }
    TRY(g->alloc(&g->part, (size_t)2 * L * g->part_site));
    TRY(g->alloc(&g->sync_fail, 4));
    if (rc == WMAR_OK && hipMemsetAsync(g->sync_fail, 0, 16, st) != hipSuccess) { set_error("rar_create: memset failed"); rc = WMAR_EHIP; }
    if (rc == WMAR_OK) {
        // k_resid_mod's workgroups wait for each other's partial sums inside ONE launch: every workgroup of its grid (chunks x row
        // tiles, at most 64 x 4) must be resident at the same time.  Checked here against what the device can hold (the occupancy
        // the runtime reports for the widest instantiation x the compute units it exposes -- a CU mask or a partition mode shrinks
        // it); a device that cannot runs the two-launch pair (k_resid_mod<., ., 1> + <., ., 2>: same results) from the start.  A
        // wait that gives up at run time raises sync_fail: the call is then repeated on the two-launch pair (rar_sync_failed).
        int nb = 0, dev = 0, cus = 0;
        hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_resid_mod<8, 4, 0>, 256, 0);
        if (e1 == hipSuccess) e1 = hipGetDevice(&dev);
        if (e1 == hipSuccess) e1 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int kbd = D / 8, kpw = kbd <= 256 ? 1 : 4;
        const long long need = (long long)((kbd + 4 * kpw - 1) / (4 * kpw)) * g->MTmax;
        if (e1 != hipSuccess) { set_error("rar_create: occupancy query failed: %s", hipGetErrorString(e1)); rc = WMAR_EHIP; }
        else g->rm_fused = (long long)nb * cus >= need && getenv("WMAR_NO_XR") == nullptr;
        { const char* e = getenv("WMAR_INJECT_SYNC_FAIL"); g->inject_fail = (e && atoi(e) > 0) ? 1 : 0; }
    }
    if (rc == WMAR_OK) {
        hipError_t er = hipMemsetAsync(g->x, 0, Mpad * D * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->h, 0, Mpad * D * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->y, 0, Mpad * D * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->sc, 0, Mpad * D * 4, st);
        if (er == hipSuccess && g->scq) er = hipMemsetAsync(g->scq, 0, (size_t)D / 16 * 2 * 3 * 64 * 16, st);      // rows past the batch are never written
        if (er == hipSuccess) er = hipMemsetAsync(g->hbuf, 0, Mpad * F * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->mod, 0, Mpad * (size_t)g->Ntot * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->kcache, 0, kv * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->vcache, 0, kv * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->ids, 0, (size_t)g->Bmax * cfg->image_seq_len * 8, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->cond_ids, 0, (size_t)g->Mmax * 8, st);
        if (er == hipSuccess) er = hipStreamCreateWithFlags(&g->cap_stream, hipStreamNonBlocking);
        if (er == hipSuccess) er = hipEventCreate(&g->ev);
        if (er == hipSuccess) er = hipStreamSynchronize(st);
        if (er != hipSuccess) { set_error("rar_create: %s", hipGetErrorString(er)); rc = WMAR_EHIP; }
    }
#undef TRY
    if (rc != WMAR_OK) { delete g; return rc; }
    *out = g;
    return WMAR_OK;
}

void wmar_rar_destroy(wmar_rar* g) { delete g; }

// The in-launch waits of k_resid_mod raise g->sync_fail when they give up.  Returns 1 when the flag was up: the engine has then been
// switched to the two-launch pair, its graph dropped and the flag cleared -- the caller repeats its work (deterministic in its
// inputs).  0: clean.  < 0: HIP error.
// `always`: read the flag on the two-launch pair as well (wmar_rar_check, which the Python engine calls behind every generation): there
// the second launch raises it when it reads a still-poisoned partial sum, i.e. an ORDERING bug of the pair -- nothing to fall back to,
// returns 2.  The generation loops themselves pass false and stay asynchronous on the pair.
static int rar_sync_failed(wmar_rar* g, hipStream_t st, bool always = false) {
    if (!g->rm_fused && !always) return 0;
    unsigned f = 0;
    WMAR_HIP_CHECK(hipMemcpyAsync(&f, g->sync_fail, 4, hipMemcpyDeviceToHost, st));
    WMAR_HIP_CHECK(hipStreamSynchronize(st));
    if (!f) return 0;
    if (!g->rm_fused) { WMAR_HIP_CHECK(hipMemsetAsync(g->sync_fail, 0, 4, st)); return 2; }
    g->drop_graph();
    WMAR_HIP_CHECK(hipMemsetAsync(g->sync_fail, 0, 4, st));
    g->rm_fused = false;
    g->fallbacks += 1;
    return 1;
}
static int rar_inject(wmar_rar* g, hipStream_t st) {
    if (!g->inject_fail || !g->rm_fused) return WMAR_OK;
    g->inject_fail = 0;
    hipLaunchKernelGGL(k_set3, dim3(1), dim3(1), 0, st, (int*)g->sync_fail, 1, 0, 0);
    return launch_status("k_set3");
}
int wmar_rar_check(wmar_rar* g, void* stream) {
    WMAR_REQUIRE(g, "rar_check: null argument");
    // generate / forward_position verify and recover by themselves; kept for callers that want to know whether the fused launch still runs
    const int f = rar_sync_failed(g, (hipStream_t)stream, true);
    if (f < 0) return f;
    if (f == 2) {
        set_error("rar: the second launch of the two-launch residual pair read a partial sum its first launch had not written (poisoned "
                  "word): the launches since the last check are invalid");
        return WMAR_EHIP;
    }
    if (f > 0) {
        set_error("rar: an in-launch wait of k_resid_mod gave up (its workgroups were not co-resident): the launches since the last check "
                  "are invalid (the engine continues on the two-launch pair)");
        return WMAR_EHIP;
    }
    return WMAR_OK;
}
int wmar_rar_launch_status(const wmar_rar* g, int32_t* fused, int32_t* fallbacks) {
    WMAR_REQUIRE(g, "rar_launch_status: null argument");
    if (fused) *fused = g->rm_fused ? 1 : 0;
    if (fallbacks) *fallbacks = g->fallbacks;
    return WMAR_OK;
}
int64_t wmar_rar_device_bytes(const wmar_rar* g) { return g ? g->mem.bytes : 0; }

int wmar_rar_forward_position(wmar_rar* g, const int64_t* tok_dev, const int64_t* cond_ids_dev, int64_t M, int32_t pos,
                              float* logits_dev, void* stream) {
    WMAR_REQUIRE(g && tok_dev && cond_ids_dev && logits_dev, "rar_forward_position: null argument");
    WMAR_REQUIRE(M >= 1 && M <= g->Mmax, "rar_forward_position: rows %lld outside 1..%d", (long long)M, g->Mmax);
    WMAR_REQUIRE(pos >= 0 && pos < g->T, "rar_forward_position: position %d outside 0..%d", pos, g->T - 1);
    hipStream_t st = (hipStream_t)stream;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (int rc = rar_inject(g, st)) return rc;
        WMAR_HIP_CHECK(hipMemcpyAsync(g->cond_ids, cond_ids_dev, (size_t)M * 8, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_set3, dim3(1), dim3(1), 0, st, g->ctr, (int)pos, 0, 0);
        RarPlan p(g, (int)M, (int)M, (const long long*)tok_dev, st);
        if (int rc = p.position(true, logits_dev)) return rc;
        // on the fused path the call waits for the position and checks the in-launch waits; a failed position is repeated on the
        // two-launch pair (a pure function of the tokens, the position and the cache rows below it)
        const int f = rar_sync_failed(g, st);
        if (f < 0) return f;
        if (f == 0) return WMAR_OK;
    }
    set_error("rar_forward_position: the in-launch wait flag is up on the two-launch pair");
    return WMAR_EHIP;
}

static int rar_generate_once(wmar_rar* g, const wmar_wm_ctx* wm, const int64_t* class_ids_dev, int64_t B,
                             const float* cfg_scale_host, int32_t use_guidance, float temperature, const float* q_dev,
                             const float* log_rs_dev, float top_p, int32_t top_k,
                             int64_t* tokens_out_dev, int32_t use_graph, void* stream) {
    WMAR_REQUIRE(g && class_ids_dev && (q_dev || log_rs_dev) && tokens_out_dev, "rar_generate: null argument");
    WMAR_REQUIRE(B >= 1 && B <= g->Bmax, "rar_generate: batch %lld outside 1..%d", (long long)B, g->Bmax);
    WMAR_REQUIRE(!use_guidance || cfg_scale_host, "rar_generate: guidance scales missing");
    if (wm) WMAR_REQUIRE(wm->table_dev && wm->vocab_size == g->V, "rar_generate: watermark vocab mismatch");
    hipStream_t st = (hipStream_t)stream;
    const int L = g->cfg.image_seq_len, V = g->V;
    const int M = use_guidance ? 2 * (int)B : (int)B;
    g->drop_graph();
    if (int rc = rar_inject(g, st)) return rc;
    // condition ids: class + codebook_size + 1, unconditional rows get the "none" id (rar.py:303-312)
    std::vector<long long> hc((size_t)B);
    WMAR_HIP_CHECK(hipMemcpyAsync(hc.data(), class_ids_dev, (size_t)B * 8, hipMemcpyDeviceToHost, st));
    WMAR_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<long long> ci((size_t)M);
    const long long none_id = (long long)g->cfg.condition_num_classes + V + 1;
    for (int b = 0; b < (int)B; ++b) {
        WMAR_REQUIRE(hc[b] >= 0 && hc[b] < g->cfg.condition_num_classes, "class id %lld out of range", hc[b]);
        ci[b] = hc[b] + V + 1;
        if (use_guidance) ci[B + b] = none_id;
    }
    WMAR_HIP_CHECK(hipMemcpyAsync(g->cond_ids, ci.data(), (size_t)M * 8, hipMemcpyHostToDevice, st));
    if (use_guidance) WMAR_HIP_CHECK(hipMemcpyAsync(g->cfg_scale, cfg_scale_host, (size_t)L * 4, hipMemcpyHostToDevice, st));
    WMAR_HIP_CHECK(hipStreamSynchronize(st));   // host staging vectors go out of scope

    if (use_guidance && !g->mod_u_ready) {
        // every unconditional row has the same condition: its adaLN modulations depend on the position only
        const int MTt = (g->T + 31) / 32;
        float4* tmp = nullptr;
        WMAR_HIP_CHECK(hipMalloc(&tmp, (size_t)MTt * 32 * g->D * 4));
        hipLaunchKernelGGL(k_rar_cond_rows, dim3((unsigned)(g->D / 8 * MTt)), dim3(64), 0, st, tmp, g->emb, g->tstep, none_id, g->T, MTt, g->D);
        GemmArgs a{};
        a.MT = MTt; a.B = MTt * 32; a.K = g->D; a.D = g->D; a.Wp = g->wada; a.Xp = tmp; a.KB = g->D / 8; a.NT = (int)(g->Ntot / 32);
        a.bias = g->bada; a.logits = g->mod_u; a.V = (int)g->Ntot; a.pos_dev = g->ctr;
        int rc = gemm_dispatch<EPI_LOGITS, false>(a, false, st);
        hipError_t e = hipStreamSynchronize(st);
        (void)hipFree(tmp);
        if (rc) return rc;
        if (e != hipSuccess) { set_error("unconditional adaLN table: %s", hipGetErrorString(e)); return WMAR_EHIP; }
        g->mod_u_ready = true;
    }
    const bool shared_u = use_guidance != 0;
    RarPlan p(g, M, (int)B, nullptr, st, shared_u);
    // position 0: the cls token (no logits needed)
    hipLaunchKernelGGL(k_set3, dim3(1), dim3(1), 0, st, g->ctr, 0, 0, 0);
    if (int rc = p.position(false, nullptr)) return rc;
    hipLaunchKernelGGL(k_set3, dim3(1), dim3(1), 0, st, g->ctr, 1, 0, 0);   // pos = 1, step = 0, len(ids) = 0

    SampArgs a{};
    a.wm = make_wm(wm);
    a.logits = g->logits; a.V = V; a.past = g->ids; a.past_stride = L; a.t_dev = g->ctr + 2;
    a.temperature = temperature; a.top_k = 0; a.use_top_p = 0; a.top_p_thr = 0.f;
    a.q = q_dev; a.q_step_stride = (long long)B * V; a.step_dev = g->ctr + 1;
    a.scratch = a.V > 65536 ? g->scratch : nullptr;   /* rows up to 65536 entries live in the sampler's registers */
    a.tok_out = (long long*)tokens_out_dev; a.tok_out_stride = L;
    a.past_append = g->ids; a.trace = nullptr; a.B = B;
    if (use_guidance) { a.logits_uncond = g->logits + (long long)B * V; a.cfg_scale = g->cfg_scale; }
    GumbelArgs ga{};
    ga.logits = g->logits; ga.logits_uncond = a.logits_uncond; ga.cfg_scale = a.cfg_scale; ga.step_dev = g->ctr + 1;
    ga.t_dev = g->ctr + 2; ga.V = V; ga.B = B; ga.log_rs = log_rs_dev; ga.key_row_stride = 0;
    ga.use_sampling = 1; ga.temp = temperature; ga.top_p = top_p; ga.top_k = top_k;
    ga.tok_out = (long long*)tokens_out_dev; ga.tok_out_stride = L; ga.past_append = g->ids; ga.past_stride = L;

    auto one_step = [&](hipStream_t s) -> int {
        RarPlan q(g, M, (int)B, nullptr, s, shared_u);
        int rc = q.position(true, g->logits);
        if (rc) return rc;
        if ((rc = log_rs_dev ? launch_gumbel_sample(ga, s) : launch_sample_fused(a, s))) return rc;
        hipLaunchKernelGGL(k_advance3, dim3(1), dim3(1), 0, s, g->ctr);
        return launch_status("k_advance3");
    };
    if (use_graph) {
        WMAR_HIP_CHECK(hipStreamBeginCapture(g->cap_stream, hipStreamCaptureModeThreadLocal));
        // groups of four positions per captured graph (the seam between two replays is ~10 us, gpt.hip)
        const int gs = (L % 4 == 0) ? 4 : 1;
        int rc = WMAR_OK;
        for (int k = 0; k < gs && rc == WMAR_OK; ++k) rc = one_step(g->cap_stream);
        hipError_t e = hipStreamEndCapture(g->cap_stream, &g->graph);
        if (rc) { g->drop_graph(); return rc; }
        if (e != hipSuccess) { g->drop_graph(); set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return WMAR_EHIP; }
        e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { g->drop_graph(); set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return WMAR_EHIP; }
        for (int n = 0; n < L && e == hipSuccess; n += gs) e = hipGraphLaunch(g->exec, st);
        if (e == hipSuccess) e = hipEventRecord(g->ev, st);
        if (e != hipSuccess) { set_error("graph replay failed: %s", hipGetErrorString(e)); return WMAR_EHIP; }
        g->pending = true;
    } else {
        for (int n = 0; n < L; ++n)
            if (int rc = one_step(st)) return rc;
    }
    return WMAR_OK;
}

static int rar_generate_impl(wmar_rar* g, const wmar_wm_ctx* wm, const int64_t* class_ids_dev, int64_t B,
                             const float* cfg_scale_host, int32_t use_guidance, float temperature, const float* q_dev,
                             const float* log_rs_dev, float top_p, int32_t top_k,
                             int64_t* tokens_out_dev, int32_t use_graph, void* stream) {
    WMAR_REQUIRE(g, "rar_generate: null argument");
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (int rc = rar_generate_once(g, wm, class_ids_dev, B, cfg_scale_host, use_guidance, temperature, q_dev, log_rs_dev, top_p, top_k,
                                       tokens_out_dev, use_graph, stream)) return rc;
        // on the fused path the call waits for its replays and reads the wait flag; a run that raised it is repeated on the
        // two-launch pair (same inputs, same noise: the same tokens)
        const int f = rar_sync_failed(g, (hipStream_t)stream);
        if (f < 0) return f;
        if (f == 0) return WMAR_OK;
    }
    set_error("rar_generate: the in-launch wait flag is up on the two-launch pair");
    return WMAR_EHIP;
}

int wmar_rar_generate(wmar_rar* g, const wmar_wm_ctx* wm, const int64_t* class_ids_dev, int64_t B,
                      const float* cfg_scale_host, int32_t use_guidance, float temperature, const float* q_dev,
                      int64_t* tokens_out_dev, int32_t use_graph, void* stream) {
    WMAR_REQUIRE(q_dev, "rar_generate: null argument");
    return rar_generate_impl(g, wm, class_ids_dev, B, cfg_scale_host, use_guidance, temperature, q_dev, nullptr, 0.f, 0,
                             tokens_out_dev, use_graph, stream);
}

int wmar_rar_generate_gumbel(wmar_rar* g, const int64_t* class_ids_dev, int64_t B, const float* cfg_scale_host,
                             int32_t use_guidance, float temperature, float top_p, int32_t top_k,
                             const float* log_rs_dev, int64_t* tokens_out_dev, int32_t use_graph, void* stream) {
    WMAR_REQUIRE(log_rs_dev, "rar_generate_gumbel: null key");
    WMAR_REQUIRE(g && g->V <= 16384, "rar_generate_gumbel: codebook larger than 16384");
    return rar_generate_impl(g, nullptr, class_ids_dev, B, cfg_scale_host, use_guidance, temperature, nullptr, log_rs_dev, top_p,
                             top_k, tokens_out_dev, use_graph, stream);
}

}  // extern "C"
