// Chameleon / Anole decode engine for gfx950 (bf16 weights and activations, fp32 accumulation)
// -- SURVEY.md section 8a row C1.
//
// Reference: deps/chameleon/inference/transformer.py:288-353 (Transformer.forward_with_attn_bias),
// :37-160 (Attention), :163-217 (FeedForward), :220-285 (TransformerBlock, swin_norm = False);
// model_adapter.py:72-119 (ragged prefill, then one token per sequence per step);
// chameleon.py:299-389 (ImageDecoder: three guidance streams, 1024 image tokens);
// logits_processor.py:312-336 (InBatchInstructCFG), :135-156 (AllowOnlyTokens);
// token_selector.py:26-47 (multinomial on the first stream, replicated).
//
// One step = every row (sequence) consumes one token at its own position: the prompt is fed
// token by token through the same captured step (prompts are a few dozen tokens against 1024
// generated ones), rows of shorter prompts idle until their turn (AlignPromptRight).
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "cham_kernels.h"
#include "sampler.h"

namespace wmar {

struct ChamLayer {
    uint4 *wqkv, *wo, *w13, *w2;
    float *qnw, *qnb, *knw, *knb;
};

}  // namespace wmar

using namespace wmar;

struct wmar_cham {
    wmar_cham_config cfg{};
    DeviceArena mem;
    int D = 0, H = 0, Hkv = 0, hd = 0, V = 0, L = 0, F = 0, T = 0, Mmax = 0, MT = 0, Dkv = 0;
    std::vector<ChamLayer> layers;
    uint16_t* emb = nullptr;
    uint4* whead = nullptr;
    // workspaces
    uint4 *x = nullptr, *y = nullptr, *hbuf = nullptr;
    float *slabs = nullptr, *big_slabs = nullptr, *logits = nullptr, *scratch = nullptr;   // pieces of D-wide / of QKV and w13 outputs
    double* ssq = nullptr;
    uint16_t *kcache = nullptr, *vcache = nullptr;
    long long *tok = nullptr, *ids = nullptr;
    float2* rope = nullptr;
    int *pos = nullptr, *ctr = nullptr;     // ctr: [unused, step, len]
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
    ~wmar_cham() {
        drop_graph();
        mem.release();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (ev) (void)hipEventDestroy(ev);
    }
};

namespace {

// workgroups of the stream-K GEMM: one per CU (measured: 256 -> 5.3 ms/step, 512 -> 5.9, 1024 -> 6.5 on the 7B model; the kernel
// runs at one wave per SIMD, so more workgroups only add pieces and prologues).  Dev builds (-DWMAR_DEV_KNOBS): WMAR_CHAM_GRID overrides.
int cham_grid() {
#ifdef WMAR_DEV_KNOBS
    static int g = 0;
    if (!g) {
        const char* e = getenv("WMAR_CHAM_GRID");
        g = e ? atoi(e) : 256;
        if (g < 1) g = 256;
    }
    return g;
#else
    return 256;
#endif
}

SkInfo sk_for(int NT, int KB, bool whole_groups) {
    SkInfo k;
    const int groups = (NT + BG_TG - 1) / BG_TG;
    k.C = (KB + BG_KC - 1) / BG_KC;
    k.U = groups * k.C;
    const int target = cham_grid();
    if (whole_groups) {
        // every workgroup owns whole groups: the largest grid <= target that divides the group count
        int per = (groups + target - 1) / target;
        while (groups % per) ++per;
        k.G = groups / per;
    } else {
        // at most BG_MAXP pieces per group (the slab buffers are sized for that)
        auto max_pieces = [&](int G) {
            SkInfo t = k;
            t.G = G;
            int mx = 0;
            for (int g = 0; g < groups; ++g) mx = std::max(mx, sk_count(t, g));
            return mx;
        };
        k.G = std::min(target, k.U);
        while (k.G > 1 && max_pieces(k.G) > BG_MAXP) k.G -= (k.G > 64 ? 16 : 1);
    }
    return k;
}

template <int EPI>
int launch_bgemm(const BGemmArgs& a, int MT, hipStream_t st) {
    const dim3 grid((unsigned)a.sk.G);
    switch (MT) {
        case 1: hipLaunchKernelGGL((k_bgemm<1, EPI>), grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((k_bgemm<2, EPI>), grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((k_bgemm<3, EPI>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((k_bgemm<4, EPI>), grid, dim3(256), 0, st, a); break;
    }
    return launch_status("k_bgemm");
}

struct ChamPlan {
    wmar_cham* g;
    int M;
    hipStream_t st;
    int MT, KBD, KBF, nch;
    long long act8;     // floats per fp32 piece of a D-wide output
    SkInfo sk_qkv, sk_o, sk_13, sk_2, sk_head;
    ChamPlan(wmar_cham* g_, int M_, hipStream_t st_) : g(g_), M(M_), st(st_) {
        MT = g->MT; KBD = g->D / 16; KBF = g->F / 16; nch = (KBD + CHAM_STAT_KB - 1) / CHAM_STAT_KB;      // (<= 128: checked at creation)
        act8 = (long long)g->D * MT * 32;
        sk_qkv = sk_for((g->D + 2 * g->Dkv) / 32, KBD, false);
        sk_o = sk_for(g->D / 32, KBD, false);
        sk_13 = sk_for(g->F / 16, KBD, false);
        sk_2 = sk_for(g->D / 32, KBF, false);
        sk_head = sk_for(g->V / 32, KBD, true);
    }
    BGemmArgs base() const {
        BGemmArgs a{};
        a.M = M; a.ssq = g->ssq; a.n_chunks = nch; a.K = g->D; a.eps = g->cfg.norm_eps;
        return a;
    }
    int embed() {
        ChamResidArgs r{};
        r.x = g->x; r.emb = g->emb; r.tok = g->tok; r.ssq = g->ssq; r.KB = KBD; r.MT = MT; r.M = M; r.K = g->D;
        hipLaunchKernelGGL((k_cham_resid<true>), dim3((unsigned)(nch * MT)), dim3(256), 0, st, r);
        return launch_status("k_cham_resid<embed>");
    }
    int resid(const SkInfo& sk) {
        ChamResidArgs r{};
        r.x = g->x; r.slabs = g->slabs; r.slab_stride = act8; r.sk = sk; r.ssq = g->ssq; r.KB = KBD; r.MT = MT; r.M = M; r.K = g->D;
        hipLaunchKernelGGL((k_cham_resid<false>), dim3((unsigned)(nch * MT)), dim3(256), 0, st, r);
        return launch_status("k_cham_resid");
    }
    int layer(int l) {
        const ChamLayer& w = g->layers[l];
        int rc;
        const int Nqkv = g->D + 2 * g->Dkv;
        BGemmArgs q = base();
        q.Wp = w.wqkv; q.Xp = g->x; q.KB = KBD; q.NT = Nqkv / 32; q.sk = sk_qkv; q.slabs = g->big_slabs;
        q.slab_stride = (long long)Nqkv * MT * 32;
        if ((rc = launch_bgemm<BEPI_SLAB>(q, MT, st))) return rc;
        ChamAttnArgs t{};
        const long long lstride = (long long)g->Mmax * g->Hkv * g->T * g->hd;
        t.qkv_slabs = g->big_slabs; t.slab_stride = q.slab_stride; t.sk = sk_qkv; t.ssq = g->ssq; t.n_chunks = nch; t.K = g->D;
        t.eps = g->cfg.norm_eps; t.qn_w = w.qnw; t.qn_b = w.qnb; t.kn_w = w.knw; t.kn_b = w.knb;
        t.kcache = g->kcache + l * lstride; t.vcache = g->vcache + l * lstride; t.y = g->y; t.pos = g->pos;
        t.D = g->D; t.H = g->H; t.Hkv = g->Hkv; t.Tmax = g->T; t.MT = MT; t.scale = 1.0f / sqrtf((float)g->hd);
        t.rope = g->rope;
        const dim3 grid((unsigned)(M * g->H));
        static const int att_nw = getenv("WMAR_CHAM_NWA") ? atoi(getenv("WMAR_CHAM_NWA")) : 2;     // dev A/B: waves per (sequence, head)
        if (g->hd == 128 && att_nw == 1) hipLaunchKernelGGL((k_cham_attn<128, 1>), grid, dim3(64), 0, st, t);
        else if (g->hd == 128 && att_nw == 4) hipLaunchKernelGGL((k_cham_attn<128, 4>), grid, dim3(256), 0, st, t);
        else if (g->hd == 128) hipLaunchKernelGGL((k_cham_attn<128, 2>), grid, dim3(128), 0, st, t);
        else hipLaunchKernelGGL((k_cham_attn<64, 2>), grid, dim3(128), 0, st, t);
        if ((rc = launch_status("k_cham_attn"))) return rc;
        BGemmArgs o = base();
        o.Wp = w.wo; o.Xp = g->y; o.KB = KBD; o.NT = g->D / 32; o.sk = sk_o; o.slabs = g->slabs; o.slab_stride = act8;
        if ((rc = launch_bgemm<BEPI_SLAB>(o, MT, st))) return rc;
        if ((rc = resid(sk_o))) return rc;
        BGemmArgs f = base();
        f.Wp = w.w13; f.Xp = g->x; f.KB = KBD; f.NT = g->F / 16; f.sk = sk_13; f.slabs = g->big_slabs;
        f.slab_stride = (long long)2 * g->F * MT * 32;
        if ((rc = launch_bgemm<BEPI_SLAB>(f, MT, st))) return rc;
        SwigluArgs sw{};
        sw.slabs = g->big_slabs; sw.slab_stride = f.slab_stride; sw.sk = sk_13; sw.out = g->hbuf; sw.ssq = g->ssq; sw.n_chunks = nch;
        sw.K = g->D; sw.eps = g->cfg.norm_eps; sw.NT = g->F / 16; sw.MT = MT;
        hipLaunchKernelGGL(k_cham_swiglu, dim3((unsigned)((sw.NT + 3) / 4), (unsigned)MT), dim3(256), 0, st, sw);
        if ((rc = launch_status("k_cham_swiglu"))) return rc;
        BGemmArgs d = base();
        d.Wp = w.w2; d.Xp = g->hbuf; d.KB = KBF; d.NT = g->D / 32; d.sk = sk_2; d.slabs = g->slabs; d.slab_stride = act8;
        if ((rc = launch_bgemm<BEPI_SLAB>(d, MT, st))) return rc;
        return resid(sk_2);
    }
    int head(float* logits_out) {
        BGemmArgs a = base();
        a.Wp = g->whead; a.Xp = g->x; a.KB = KBD; a.NT = g->V / 32; a.sk = sk_head; a.logits = logits_out; a.V = g->V;
        return launch_bgemm<BEPI_LOGITS>(a, MT, st);
    }
    int step(bool with_head, float* logits_out) {
        int rc;
        if ((rc = embed())) return rc;
        for (int l = 0; l < g->L; ++l)
            if ((rc = layer(l))) return rc;
        return with_head ? head(logits_out) : WMAR_OK;
    }
};

int bpack(const void* src, const void* gamma, uint4* dst, int N, int K, int mode, int Hd, int src_bf16, hipStream_t st) {
    BPackArgs a{};
    a.src = src; a.gamma = gamma; a.dst = dst; a.N = N; a.K = K; a.mode = mode; a.Hd = Hd; a.src_bf16 = src_bf16;
    const int NT = mode == 0 ? (N + 31) / 32 : Hd / 16;
    hipLaunchKernelGGL(k_bpack, dim3((unsigned)((long long)NT * (K / 16))), dim3(64), 0, st, a);
    return launch_status("k_bpack");
}

}  // namespace

extern "C" {

int wmar_cham_create(const wmar_cham_config* cfg, const char* const* names, const void* const* tensors_dev,
                     int32_t n_tensors, void* stream, wmar_cham** out) {
    WMAR_REQUIRE(cfg && names && tensors_dev && out, "cham_create: null argument");
    const int D = cfg->dim, H = cfg->n_heads, Hkv = cfg->n_kv_heads, L = cfg->n_layers, F = cfg->ffn_hidden, V = cfg->vocab_size;
    WMAR_REQUIRE(H > 0 && Hkv > 0 && D % H == 0 && H % Hkv == 0, "cham_create: bad head counts");
    const int hd = D / H;
    WMAR_REQUIRE(hd == 64 || hd == 128, "cham_create: head_dim %d unsupported (64, 128)", hd);
    WMAR_REQUIRE(D % 32 == 0 && F % 16 == 0 && V % 32 == 0 && (Hkv * hd) % 16 == 0, "cham_create: dim / ffn / vocab alignment");
    WMAR_REQUIRE(!cfg->swin_norm, "cham_create: swin_norm (Chameleon-30B block order) is not supported");
    WMAR_REQUIRE(cfg->max_rows >= 1 && cfg->max_rows <= 128, "cham_create: max_rows must be in 1..128");
    WMAR_REQUIRE(cfg->max_seq_len >= 2, "cham_create: max_seq_len too small");
    WMAR_REQUIRE((D / 16 + CHAM_STAT_KB - 1) / CHAM_STAT_KB <= 128, "cham_create: dim %d gives more than 128 statistics chunks (k_cham_attn reads two per lane)", D);
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) tm.m[names[i]] = tensors_dev[i];
    hipStream_t st = (hipStream_t)stream;
    auto* g = new wmar_cham();
    g->cfg = *cfg; g->D = D; g->H = H; g->Hkv = Hkv; g->hd = hd; g->V = V; g->L = L; g->F = F; g->Dkv = Hkv * hd;
    g->T = cfg->max_seq_len; g->Mmax = cfg->max_rows; g->MT = (cfg->max_rows + 31) / 32;
    const int bf = cfg->tensors_bf16;
    int rc = WMAR_OK;
    auto need = [&](const std::string& k) -> const void* {
        auto it = tm.m.find(k);
        if (it == tm.m.end()) {
            if (rc == WMAR_OK) { set_error("checkpoint tensor '%s' is missing", k.c_str()); rc = WMAR_EMISSING; }
            return nullptr;
        }
        return it->second;
    };
#define TRY(x) do { if (rc == WMAR_OK) rc = (x); } while (0)
    auto vec_f32 = [&](float** dst, const void* src, size_t n) -> int {
        if (int r = g->mem.alloc(dst, n)) return r;
        hipLaunchKernelGGL(k_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, *dst, (long long)n, bf);
        return launch_status("k_to_f32");
    };
    const void *e = need("tok_embeddings.weight"), *nw = need("norm.weight"), *ow = need("output.weight");
    if (rc == WMAR_OK) {
        TRY(g->mem.alloc(&g->emb, (size_t)V * D));
        if (rc == WMAR_OK) {
            const long long n = (long long)V * D;
            hipLaunchKernelGGL(k_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, e, g->emb, n, bf);
            rc = launch_status("k_to_bf16");
        }
        TRY(g->mem.alloc(&g->whead, (size_t)V * D / 8));
        TRY(bpack(ow, nw, g->whead, V, D, 0, 0, bf, st));
    }
    g->layers.resize(L);
    const int Nqkv = D + 2 * g->Dkv;
    for (int l = 0; l < L && rc == WMAR_OK; ++l) {
        const std::string p = "layers." + std::to_string(l) + ".";
        ChamLayer& w = g->layers[l];
        const void *qkv = need(p + "attention.wqkv.weight"), *wo = need(p + "attention.wo.weight"),
                   *w13 = need(p + "feed_forward.w13.weight"), *w2 = need(p + "feed_forward.w2.weight"),
                   *an = need(p + "attention_norm.weight"), *fn = need(p + "ffn_norm.weight");
        const void *qw = nullptr, *qb = nullptr, *kw = nullptr, *kb = nullptr;
        if (cfg->qk_normalization) {
            qw = need(p + "attention.q_normalization.weight"); qb = need(p + "attention.q_normalization.bias");
            kw = need(p + "attention.k_normalization.weight"); kb = need(p + "attention.k_normalization.bias");
        }
        if (rc != WMAR_OK) break;
        TRY(g->mem.alloc(&w.wqkv, (size_t)Nqkv * D / 8)); TRY(bpack(qkv, an, w.wqkv, Nqkv, D, 0, 0, bf, st));
        TRY(g->mem.alloc(&w.wo, (size_t)D * D / 8)); TRY(bpack(wo, nullptr, w.wo, D, D, 0, 0, bf, st));
        TRY(g->mem.alloc(&w.w13, (size_t)2 * F * D / 8)); TRY(bpack(w13, fn, w.w13, 2 * F, D, 1, F, bf, st));
        TRY(g->mem.alloc(&w.w2, (size_t)D * F / 8)); TRY(bpack(w2, nullptr, w.w2, D, F, 0, 0, bf, st));
        w.qnw = w.qnb = w.knw = w.knb = nullptr;
        if (cfg->qk_normalization) {
            TRY(vec_f32(&w.qnw, qw, (size_t)hd)); TRY(vec_f32(&w.qnb, qb, (size_t)hd));
            TRY(vec_f32(&w.knw, kw, (size_t)hd)); TRY(vec_f32(&w.knb, kb, (size_t)hd));
        }
    }
    const size_t Mpad = (size_t)g->MT * 32;
    TRY(g->mem.alloc(&g->x, Mpad * D / 8));
    TRY(g->mem.alloc(&g->y, Mpad * D / 8));
    TRY(g->mem.alloc(&g->hbuf, Mpad * F / 8));
    TRY(g->mem.alloc(&g->slabs, (size_t)BG_MAXP * Mpad * D));
    TRY(g->mem.alloc(&g->big_slabs, (size_t)BG_MAXP * Mpad * (size_t)std::max(Nqkv, 2 * F)));
    TRY(g->mem.alloc(&g->ssq, (size_t)((D / 16 + CHAM_STAT_KB - 1) / CHAM_STAT_KB) * Mpad));
    const size_t kv = (size_t)L * g->Mmax * Hkv * g->T * hd;
    TRY(g->mem.alloc(&g->kcache, kv));
    TRY(g->mem.alloc(&g->vcache, kv));
    TRY(g->mem.alloc(&g->logits, (size_t)g->Mmax * V));
    TRY(g->mem.alloc(&g->scratch, (size_t)g->Mmax * V));
    TRY(g->mem.alloc(&g->tok, Mpad));
    TRY(g->mem.alloc(&g->pos, Mpad));
    TRY(g->mem.alloc(&g->ids, (size_t)g->Mmax * (size_t)(g->T + 4)));
    TRY(g->mem.alloc(&g->ctr, 4));
    TRY(g->mem.alloc(&g->rope, (size_t)g->T * (hd / 2)));
    if (rc == WMAR_OK) {
        const long long n = (long long)g->T * (hd / 2);
        hipLaunchKernelGGL(k_rope_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g->rope, g->T, hd / 2, cfg->rope_theta);
        rc = launch_status("k_rope_table");
    }
    if (rc == WMAR_OK) {
        hipError_t er = hipMemsetAsync(g->x, 0, Mpad * D * 2, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->y, 0, Mpad * D * 2, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->slabs, 0, (size_t)BG_MAXP * Mpad * D * 4, st);                       // padding rows are never written
        if (er == hipSuccess) er = hipMemsetAsync(g->big_slabs, 0, (size_t)BG_MAXP * Mpad * (size_t)std::max(Nqkv, 2 * F) * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->hbuf, 0, Mpad * F * 2, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->kcache, 0, kv * 2, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->vcache, 0, kv * 2, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->tok, 0, Mpad * 8, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->pos, 0, Mpad * 4, st);
        if (er == hipSuccess) er = hipMemsetAsync(g->ctr, 0, 16, st);
        if (er == hipSuccess) er = hipStreamCreateWithFlags(&g->cap_stream, hipStreamNonBlocking);
        if (er == hipSuccess) er = hipEventCreate(&g->ev);
        if (er == hipSuccess) er = hipStreamSynchronize(st);
        if (er != hipSuccess) { set_error("cham_create: %s", hipGetErrorString(er)); rc = WMAR_EHIP; }
    }
#undef TRY
    if (rc != WMAR_OK) { delete g; return rc; }
    *out = g;
    return WMAR_OK;
}

void wmar_cham_destroy(wmar_cham* g) { delete g; }
int64_t wmar_cham_device_bytes(const wmar_cham* g) { return g ? g->mem.bytes : 0; }

int wmar_cham_forward_tokens(wmar_cham* g, const int64_t* tok_dev, const int32_t* pos_dev, int64_t M, float* logits_dev,
                             void* stream) {
    WMAR_REQUIRE(g && tok_dev && pos_dev, "cham_forward_tokens: null argument");
    WMAR_REQUIRE(M >= 1 && M <= g->Mmax, "cham_forward_tokens: rows %lld outside 1..%d", (long long)M, g->Mmax);
    hipStream_t st = (hipStream_t)stream;
    if (g->pending) { (void)hipEventSynchronize(g->ev); g->pending = false; }
    WMAR_HIP_CHECK(hipMemcpyAsync(g->tok, tok_dev, (size_t)M * 8, hipMemcpyDeviceToDevice, st));
    WMAR_HIP_CHECK(hipMemcpyAsync(g->pos, pos_dev, (size_t)M * 4, hipMemcpyDeviceToDevice, st));
    ChamPlan p(g, (int)M, st);
    return p.step(logits_dev != nullptr, logits_dev);
}

int wmar_cham_generate_image(wmar_cham* g, const wmar_wm_ctx* wm, const int64_t* prompt_tokens_host,
                             const int32_t* prompt_lens_host, int64_t B, const wmar_cham_sample_params* sp,
                             const uint32_t* allow_dev, const int32_t* allow_ids_dev, int32_t n_allow, const float* q_dev,
                             int32_t n_tokens, int64_t* tokens_out_dev, void* stream) {
    WMAR_REQUIRE(g && prompt_tokens_host && prompt_lens_host && sp && q_dev && tokens_out_dev, "cham_generate_image: null argument");
    WMAR_REQUIRE(B >= 1 && 3 * B <= g->Mmax, "cham_generate_image: batch %lld needs %lld rows, engine has %d", (long long)B,
                 (long long)(3 * B), g->Mmax);
    WMAR_REQUIRE(n_tokens >= 1, "cham_generate_image: n_tokens");
    WMAR_REQUIRE(!(sp->top_p >= 0) || sp->top_p <= 1.0, "`top_p` has to be a float > 0 and < 1, but is %f", sp->top_p);
    if (wm) WMAR_REQUIRE(wm->table_dev && wm->vocab_size == g->V, "cham_generate_image: watermark vocab mismatch");
    if (wm) WMAR_REQUIRE(wm->seed_strategy != WMAR_SEED_SPATIAL && wm->context_size <= 3,
                         "Chameleon supports fixed / linear seeding with context size <= 3 (generate.py of the reference: fixed / linear only)");
    hipStream_t st = (hipStream_t)stream;
    const int M = 3 * (int)B, V = g->V;
    int maxlen = 0;
    for (int m = 0; m < M; ++m) {
        WMAR_REQUIRE(prompt_lens_host[m] >= 1, "cham_generate_image: empty prompt in row %d", m);
        maxlen = std::max(maxlen, (int)prompt_lens_host[m]);
    }
    WMAR_REQUIRE(maxlen + n_tokens <= g->T, "cham_generate_image: %d prompt + %d image tokens exceed max_seq_len %d", maxlen,
                 n_tokens, g->T);
    g->drop_graph();
    // right-aligned prompt tables (alignment.py:27-52): row m idles (token 0 at position 0) until its prompt starts
    // the watermark sees the whole padded row: keep its last CTX entries (prompt tokens, pad_id where the row is still padding)
    const int CTX = maxlen < 3 ? maxlen : 3;
    std::vector<long long> tt((size_t)maxlen * M), first_ctx((size_t)B * 3, 0);
    std::vector<int> tp((size_t)maxlen * M);
    size_t off = 0;
    for (int m = 0; m < M; ++m) {
        const int len = prompt_lens_host[m];
        for (int j = 0; j < maxlen; ++j) {
            const int i = j - (maxlen - len);
            const long long tk = i >= 0 ? prompt_tokens_host[off + i] : 0;
            WMAR_REQUIRE(tk >= 0 && tk < V, "cham_generate_image: prompt token %lld out of range", tk);
            tt[(size_t)j * M + m] = tk;
            tp[(size_t)j * M + m] = i >= 0 ? i : 0;
        }
        if (m < B)
            for (int c = 0; c < CTX; ++c) {
                const int i = len - CTX + c;
                first_ctx[(size_t)m * 3 + c] = i >= 0 ? prompt_tokens_host[off + i] : (long long)sp->pad_id;
            }
        off += len;
    }
    long long* tab_tok = nullptr;
    int* tab_pos = nullptr;
    WMAR_HIP_CHECK(hipMalloc(&tab_tok, tt.size() * 8));
    if (hipMalloc(&tab_pos, tp.size() * 4) != hipSuccess) {
        (void)hipFree(tab_tok);
        set_error("cham_generate_image: out of device memory for the prompt tables");
        return WMAR_ENOMEM;
    }
    int rc = WMAR_OK;
    hipError_t e = hipMemcpyAsync(tab_tok, tt.data(), tt.size() * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(tab_pos, tp.data(), tp.size() * 4, hipMemcpyHostToDevice, st);
    // the watermark context is the whole input row: the tail of the padded prompt, then the generated tokens
    const long long ids_stride = g->T + 4;
    if (e == hipSuccess) e = hipMemcpy2DAsync(g->ids, ids_stride * 8, first_ctx.data(), 3 * 8, (size_t)CTX * 8, (size_t)B, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("cham_generate_image: %s", hipGetErrorString(e)); rc = WMAR_EHIP; }
    ChamPlan p(g, M, st);
    for (int j = 0; j < maxlen && rc == WMAR_OK; ++j) {
        hipLaunchKernelGGL(k_cham_step, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, g->tok, g->pos, tab_tok, tab_pos, j, M,
                           (const long long*)nullptr, 0ll, (const int*)nullptr, (int)B, 0);
        rc = p.step(j == maxlen - 1, g->logits);
    }
    hipLaunchKernelGGL(k_cham_set3, dim3(1), dim3(1), 0, st, g->ctr, 0, 0, CTX);    // step = 0, len(ids row) = CTX
    if (rc == WMAR_OK) rc = launch_status("k_set3");

    SampArgs a{};
    a.wm = make_wm(wm);
    a.logits = g->logits; a.V = V; a.past = g->ids; a.past_stride = ids_stride; a.t_dev = g->ctr + 2;
    a.temperature = sp->temperature; a.top_k = 0; a.use_top_p = sp->top_p >= 0; a.top_p_thr = (float)(1.0 - sp->top_p);
    a.q = q_dev; a.q_step_stride = (long long)B * V; a.step_dev = g->ctr + 1;
    a.scratch = a.V > 65536 ? g->scratch : nullptr;   /* rows up to 65536 entries live in the sampler's registers */
    a.tok_out = (long long*)tokens_out_dev; a.tok_out_stride = n_tokens;
    a.past_append = g->ids; a.trace = nullptr; a.B = B;
    a.logits_img = g->logits + (long long)B * V; a.logits_uncond = g->logits + 2ll * B * V;
    a.g_text = sp->guidance_scale_text; a.g_image = sp->guidance_scale_image; a.allow = allow_dev;
    if (allow_ids_dev && n_allow > 0) { a.gather = allow_ids_dev; a.Vsrc = V; a.V = n_allow; }

    auto sample = [&](hipStream_t s) -> int {
        int r = launch_sample_fused(a, s);
        if (r) return r;
        hipLaunchKernelGGL(k_advance3, dim3(1), dim3(1), 0, s, g->ctr);
        return launch_status("k_advance3");
    };
    auto one_step = [&](hipStream_t s) -> int {      // consume the token sampled last, produce the next one
        hipLaunchKernelGGL(k_cham_step, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, s, g->tok, g->pos, (const long long*)nullptr,
                           (const int*)nullptr, 0, M, (const long long*)tokens_out_dev, (long long)n_tokens, (const int*)(g->ctr + 1),
                           (int)B, 1);
        ChamPlan q(g, M, s);
        int r = q.step(true, g->logits);
        if (r) return r;
        return sample(s);
    };
    if (rc == WMAR_OK) rc = sample(st);                 // first image token from the prefill logits
    if (rc == WMAR_OK && n_tokens > 1) {
        if (sp->use_graph) {
            // groups of four steps per captured graph (the seam between two replays is ~10 us, gpt.hip); the steps that do not
            // fill a group are enqueued directly in front of the replays
            const int gs = 4, lead = (n_tokens - 1) % gs;
            for (int n = 0; n < lead && rc == WMAR_OK; ++n) rc = one_step(st);
            if (rc == WMAR_OK && n_tokens - 1 - lead > 0) {
                e = hipStreamBeginCapture(g->cap_stream, hipStreamCaptureModeThreadLocal);
                if (e == hipSuccess) {
                    for (int k = 0; k < gs && rc == WMAR_OK; ++k) rc = one_step(g->cap_stream);
                    e = hipStreamEndCapture(g->cap_stream, &g->graph);
                }
                if (rc == WMAR_OK && e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
                for (int n = 1 + lead; n < n_tokens && rc == WMAR_OK && e == hipSuccess; n += gs) e = hipGraphLaunch(g->exec, st);
            }
            if (rc == WMAR_OK && e != hipSuccess) { set_error("cham_generate_image graph: %s", hipGetErrorString(e)); rc = WMAR_EHIP; }
        } else {
            for (int n = 1; n < n_tokens && rc == WMAR_OK; ++n) rc = one_step(st);
        }
    }
    // the prompt tables are read by the prefill only; it has been enqueued on `st`, free after it ran
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(tab_tok);
    (void)hipFree(tab_pos);
    if (rc == WMAR_OK && e2 != hipSuccess) { set_error("cham_generate_image: %s", hipGetErrorString(e2)); rc = WMAR_EHIP; }
    if (rc != WMAR_OK) g->drop_graph();
    return rc;
}

}  // extern "C"
