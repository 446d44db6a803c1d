// minGPT decode engine for gfx950 (fp32, exact-f32 MFMA).
//
// Reference: deps/taming/modules/transformer/mingpt.py:42-214 (CausalSelfAttention, Block,
// GPT.forward_with_past) and :326-368 (sample_with_past).
//
// Data layout in HBM (everything fp32):
//   * Linear weights W[N][K] are repacked once into MFMA-fragment order
//         Wp[nt][kb][lane][4]   nt = n/32, kb = k/8, lane = (n%32) + 32*((k%8)/4), i = k%4
//     so that ONE wave-wide float4 load (1 KiB contiguous) feeds four
//     v_mfma_f32_32x32x2_f32 instructions as the A operand.
//   * Activations X[M][K] (M = batch rows, padded to 32) live in the SAME fragment order
//         Xp[kb][mt][lane][4]   mt = m/32, lane = (m%32) + 32*((k%8)/4)
//     and feed the B operand.  With A = W and B = X^T the MFMA result registers
//     (n = 8*(reg/4) + 4*(lane/32) + reg%4, m = lane%32) ARE the packed layout of the next
//     layer's activation: every GEMM epilogue stores whole float4s, 1 KiB per wave.
//   * KV cache [L][B][H][Tmax][hd]; one wave per (sequence, head) streams its K and V rows
//     with 1 KiB coalesced loads.
//   * The step position lives in device memory so one captured hipGraph replays for all steps.
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "decoder_host.h"
#include "decode_small.h"
#include "decode_persist.h"


using namespace wmar;

// ------------------------------------------------------------------------------ engine
struct LayerW {
    float4 *wqkv, *wproj, *wfc1, *wfc2;
    float4* wfc1x16 = nullptr;   // FC1 in k_fc1x's 24-column packing (null: shape not eligible)
    float2* wfc1x8 = nullptr;
    float4* wqkvx_bx = nullptr;  // QKV weights in k_qkvx_bx's 16-k step order (null: shape not eligible)
    float4* wproj_bx = nullptr;  // output projection in k_pack_bx order for k_bx (null: shape not eligible)
    float *bqkv, *bproj, *bfc1, *bfc2, *cqkv, *cfc1;
};

// Row-major weights of the small-batch path (decode_small.h): the checkpoint's own layout, LayerNorm NOT folded.
struct SmallW {
    float *wqkv = nullptr, *bqkv = nullptr, *wproj = nullptr, *bproj = nullptr, *wfc1 = nullptr, *bfc1 = nullptr, *wfc2 = nullptr, *bfc2 = nullptr;
    float *ln1w = nullptr, *ln1b = nullptr, *ln2w = nullptr, *ln2b = nullptr;
};

struct wmar_gpt {
    wmar_gpt_config cfg{};
    int D = 0, H = 0, hd = 0, V = 0, L = 0, Tmax = 0, Bmax = 0, MTmax = 0;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    std::vector<LayerW> layers;
    // small-batch path (1..12 rows, n_embd 1536, head_dim 64): streaming kernels on row-major weights, five launches per layer
    std::vector<SmallW> sw;
    float *whead_rm = nullptr, *lnfw = nullptr, *lnfb = nullptr;
    float *xs = nullptr, *ys = nullptr, *hs = nullptr, *qs = nullptr;     // [8][D], [8][D], [8][4D], [8][D] row-major
    bool small_ok = false;
    // ... and for 1..5 rows the whole step as ONE persistent launch (decode_persist.h): needs all 256 workgroups resident and the
    // blockIdx % 8 -> XCD grouping; a barrier that gives up switches the engine back to the five-launch plan (the call is re-run)
    float* sw_arena = nullptr;         // all layers' small-batch weights, SS_LAYER_FLOATS per layer (SmallW points into it)
    unsigned* ss_bar = nullptr;        // barrier words + fail flag (SS_BAR_WORDS)
    bool persist_ok = false;
    bool ps_enqueued = false;          // a persistent launch was enqueued since the last flag check
    bool xr_enqueued = false;      // a fused projection launch (k_bx_xr) was enqueued since the last flag check
    float *tok_emb = nullptr, *pos_emb = nullptr, *bhead = nullptr, *chead = nullptr;
    float4* whead = nullptr;
    // workspaces
    float4 *x = nullptr, *x2 = nullptr, *y = nullptr, *hbuf = nullptr, *slabs = nullptr, *qkv_slabs = nullptr;
    u32x4* yq = nullptr;           // attention output as bf16 pieces (k_bx output projection; 64-row steps)
    float* qbuf = nullptr;
    double* stats = nullptr;
    double* stats_q = nullptr;   // [QKV_SLABS_MAX][Mpad][2]: LN1 row sums per K slice, written by k_qkvx
    float *kcache = nullptr, *vcache = nullptr;
    float *logits = nullptr, *scratch = nullptr;
    long long* past = nullptr;  // [Bmax][Tmax+1]
    int *pos_dev = nullptr, *step_dev = nullptr;
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // One captured step per attention phase: the decode attention runs 1 / 2 / 4 waves per (sequence, head) while the
    // cache is short / medium / long (att_phase(): fixed cost 7.0 / 9.8 / 13.5 us against loads in flight).
    static constexpr int N_PHASE = 3;
    hipGraph_t graph[N_PHASE] = {nullptr, nullptr, nullptr};
    hipGraphExec_t exec[N_PHASE] = {nullptr, nullptr, nullptr};
    int att_nw = 2;                // waves per attention workgroup of the step being enqueued
    unsigned long long graph_key[12] = {0};
    bool pending = false;          // replays of `exec` may still be running (ev1 marks their end)
    void drop_graph() {
        if (pending && ev1) (void)hipEventSynchronize(ev1);
        pending = false;
        for (int i = 0; i < N_PHASE; ++i) {
            if (exec[i]) (void)hipGraphExecDestroy(exec[i]);
            if (graph[i]) (void)hipGraphDestroy(graph[i]);
            exec[i] = nullptr; graph[i] = nullptr;
        }
    }
    // phase of the step that attends to `kv` cached rows (including the new one)
    // Measured at batch 64 with the QKV projection arriving in 7 split-K pieces (round 2): one wave per (sequence, head) is the
    // fastest variant at every cache length up to 256 (the prologue that sums the pieces runs in wave 0 while the others
    // wait), so by default every step is phase 0.  wmar_gpt_set_attention_phases moves the thresholds.
    // Small batches (fewer (sequence, head) pairs than two per CU) are the other regime: with the chip mostly idle, 2 waves per pair
    // up to 128 cached rows and 4 beyond are 10-35 % faster (round 3, batch 1 / 8 / 16: scripts/perf_attn_nw.py).
    int att_t1 = 1 << 30, att_t2 = 1 << 30;
    bool att_user = false;         // thresholds set through wmar_gpt_set_attention_phases: used at every batch size
    int att_phase(int kv, int64_t B) const {
        if (!att_user && B * H < 512) return kv <= 128 ? 1 : 2;
        return kv <= att_t1 ? 0 : (kv <= att_t2 ? 1 : 2);
    }
    static int phase_waves(int ph) { return ph == 0 ? 1 : (ph == 1 ? 2 : 4); }
    int timing = 0;
    bool no_bx_qkv = false, no_bx_proj = false;   // dev knobs WMAR_NO_BX_QKV / WMAR_NO_BX_PROJ
    // fused output projection + residual fold + LN2 statistics (k_bx_xr): needs the block -> XCD grouping probed at creation
    unsigned* xsync = nullptr;     // [8][64] barrier words, [8*64 ..] = fail flags (placement, timeout)
    bool xcd_ok = false;           // blocks with equal blockIdx % 8 share an XCD, eight distinct XCDs (k_xcc_probe)
    bool no_xr = false;            // WMAR_NO_XR=1 at creation: keep the two-launch path
    int inject_fail = 0;           // WMAR_INJECT_SYNC_FAIL=1 at creation (tests): the next fused call finds the timeout flag raised
    int fallbacks = 0;             // calls re-run on the two-launch path after a failed in-launch barrier (wmar_gpt_plan_info reports it)
    unsigned long long* dbg_sums = nullptr; int dbg_slot = 0;   // dev only: see k_dbg_sum
    unsigned long long* stamps = nullptr;   // dev only (-DWMAR_STAMPS, WMAR_STAMPS=1 at creation): [L][5 roles][WMAR_STAMP_UNITS][8] timeline stamps
    unsigned long long* stamp_slot(int l, int role) const {
        return stamps ? stamps + ((long long)(l * 5 + role) * WMAR_STAMP_UNITS) * 8 : nullptr;
    }
    bool no_bx = false;           // dev knob WMAR_NO_BX: keep the fp32-MFMA k_qkvx
    int force_s[3] = {0, 0, 0};   // tuning knobs: WMAR_S_QKV / WMAR_S_PROJ / WMAR_S_FC2
    double step_ms = 0.0;
    // per-launch event timing (eager mode only)
    struct Span { int cls; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    bool span_on = false;
    double cls_us[WMAR_T_NCLASS] = {0};
    int64_t cls_calls[WMAR_T_NCLASS] = {0};
    hipEvent_t next_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }
    void span_begin(int cls, hipStream_t st) {
        if (!span_on) return;
        Span s{cls, next_event(), next_event()};
        (void)hipEventRecord(s.a, st);
        spans.push_back(s);
    }
    void span_end(hipStream_t st) {
        if (!span_on) return;
        (void)hipEventRecord(spans.back().b, st);
    }
    void span_collect() {
        for (int i = 0; i < WMAR_T_NCLASS; ++i) { cls_us[i] = 0; cls_calls[i] = 0; }
        for (auto& s : spans) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { cls_us[s.cls] += ms * 1000.0; cls_calls[s.cls] += 1; }
        }
        spans.clear();
        ev_used = 0;
    }

    template <typename T>
    int alloc(T** p, size_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, n * sizeof(T));
        if (e != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
            return WMAR_ENOMEM;
        }
        allocs.push_back(q);
        bytes += (int64_t)(n * sizeof(T));
        *p = (T*)q;
        return WMAR_OK;
    }
    ~wmar_gpt() {
        drop_graph();
        for (void* p : allocs) (void)hipFree(p);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    }
};

namespace {

#ifdef WMAR_DEV_KNOBS
// dev only (WMAR_DBG_SUMS=1): order-independent 64-bit checksum of a buffer after every launch of a step -- two runs of the same step
// on the same state must agree slot by slot; the first slot that differs names the launch (scripts/stress_logits.py)
__global__ void k_dbg_sum(const uint32_t* p, long long n, unsigned long long* slot) {
    unsigned long long s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        s += (unsigned long long)p[i] * (unsigned long long)(2 * i + 1);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(slot, s);
}
#endif

struct StepIO {
    const long long* tok;  // row m's token: tok[m*stride + (use_pos ? pos : 0)]
    long long tok_stride;
    int tok_use_pos;
    float* logits;
};

// One decode step = the launches below, in order.  Each role is its own function so that
// bench.py can also replay a single role back to back (wmar_gpt_profile_role).
struct StepPlan {
    wmar_gpt* g;
    int64_t B;
    StepIO io;
    hipStream_t st;
    int MT, D, KBD, KBF, nch;
    long long act;
    ResidArgs r{};
    int S_qkv = 1, S_proj = 1, S_fc2 = 1;
    int fc2_hi = 0;      // FC2: the first fc2_hi column tiles take S_fc2 + 1 K slices so that exactly 256 workgroups exist
    // fused residual fold + QKV projection (k_qkvx): column groups x K slices; 0 = not applicable to this shape (k_gemm path)
    int S_qx = 0;
    float4* xcur = nullptr;   // residual stream buffer the next launch reads (the fused fold ping-pongs between x and x2)
    bool proj_bx = false;     // 33..64 rows, n_embd a multiple of 384: output projection as k_bx on the attention's bf16 pieces
    bool proj_xr = false;     // ... with the residual fold + LN2 statistics inside the launch (k_bx_xr): no k_resid_stats behind it
    int nch_ln2 = 0;          // statistics chunks the FC1 launch reads (16 = two per XCD group behind k_bx_xr)
    bool small = false;       // 1..12 rows on an eligible engine: the streaming path of decode_small.h
    bool persist = false;     // 1..5 rows: the whole step as one persistent launch (decode_persist.h)

    StepPlan(wmar_gpt* g_, int64_t B_, const StepIO& io_, hipStream_t st_) : g(g_), B(B_), io(io_), st(st_) {
        small = B <= SG_MAX_ROWS && g->small_ok;
        persist = small && B <= SS_MAX_ROWS && g->persist_ok;
        MT = mt_for(B); D = g->D; KBD = D / 8; KBF = 4 * D / 8; nch = stat_chunks(KBD);
        act = (long long)KBD * MT * 64;  // float4 units of one [M][D] packed activation
        r.x = g->x; r.stats = g->stats; r.KB = KBD; r.MT = MT; r.n_chunks = nch;
        r.tok_emb = g->tok_emb; r.pos_emb = g->pos_emb; r.tok = io.tok; r.tok_stride = io.tok_stride;
        r.tok_use_pos = io.tok_use_pos; r.pos_dev = g->pos_dev; r.B = (int)B; r.K = D;
        r.slabs = g->slabs; r.slab_stride = act;
        xcur = g->x;
        // split factors (fixed per shape so that a role can be replayed on its own)
        S_qkv = g->force_s[0] > 0 ? (g->force_s[0] > QKV_SLABS_MAX ? QKV_SLABS_MAX : g->force_s[0]) : 1;
        S_proj = g->force_s[1] > 0 ? g->force_s[1] : split_for(D / 32, KBD);
        S_fc2 = g->force_s[2] > 0 ? g->force_s[2] : ((MT % 2 == 0 && KBF >= 128) ? 4 : split_for(D / 32, KBF));
        // one row-tile group (M = 64): split the tiles S / S+1 ways so that the grid is exactly one workgroup per CU
        // (48 tiles x 4 slices = 192 workgroups leave a quarter of the CUs idle; 16 x 6 + 32 x 5 = 256)
        if (g->force_s[2] <= 0 && MT == 2 && KBF >= 256) {
            const int NTd = D / 32;
            const int Slo = 256 / NTd;
            if (Slo >= 2 && Slo < MAX_SLABS && NTd * Slo <= 256) { S_fc2 = Slo; fc2_hi = 256 - NTd * Slo; if (fc2_hi >= NTd) fc2_hi = 0; }
        }
        // k_qkvx: 128-column groups x S slices of K, as close to one workgroup per CU as the shape allows.  Shapes it does not
        // fit (n_embd not a multiple of 128, more than 64 rows) keep the k_gemm path.
        const int NTq = 3 * D / 32;
        if (g->force_s[0] <= 0 && NTq % 4 == 0 && MT <= 2 && S_fc2 + (fc2_hi > 0 ? 1 : 0) <= MAX_SLABS) {
            const int G = NTq / 4;
            int S = 256 / G;
            if (S > QKV_SLABS_MAX) S = QKV_SLABS_MAX;
            if (S > KBD / QX_CK) S = KBD / QX_CK;
            if (S >= 1) S_qx = S;
        }
        proj_bx = MT == 2 && g->yq && g->layers[0].wproj_bx && !g->no_bx && !g->no_bx_proj && g->force_s[1] <= 0;
        if (proj_bx) S_proj = D / BX_KSLICE;
        proj_xr = proj_bx && g->xcd_ok && !g->no_xr && S_proj == 4 && (D / 32) % 8 == 0 && (D / 32 / 8) * S_proj <= 32 && (D / 32 / 8) % 2 == 0;
        nch_ln2 = proj_xr ? 16 : nch;
    }
    void dbg(const void* p, long long bytes) {
#ifdef WMAR_DEV_KNOBS
        if (!g->dbg_sums || g->dbg_slot >= 4096) return;
        hipLaunchKernelGGL(k_dbg_sum, dim3(512), dim3(256), 0, st, (const uint32_t*)p, bytes / 4, g->dbg_sums + g->dbg_slot++);
#else
        (void)p; (void)bytes;
#endif
    }
    // the predicates the launches below use (wmar_gpt_plan_info reports from the same ones)
    bool qkv_bx() const { return S_qx > 0 && MT == 2 && g->layers[0].wqkvx_bx && !g->no_bx && !g->no_bx_qkv; }
    bool fc1_x() const { return MT == 2 && g->layers[0].wfc1x16 && nch_ln2 <= 16; }      // (k_fc1x: each wave fetches four of at most 16 statistics chunks)
    int qx_nkeep() const { const int G = 3 * D / 128; return G >= 4 && S_qx * 4 <= STAT_CHUNKS_MAX ? 4 : 1; }   // keeper groups per K slice (k_qkvx_bx)
    static bool head_narrow() { static int v = -1; if (v < 0) v = getenv("WMAR_HEAD_NARROW") ? 1 : 0; return v != 0; }   // A/B: the two-launch 32-column head
    int split_for(int NT, int KB) const { return pick_split(MT % 2 == 0 ? NT * (MT / 2) : NT * MT, KB, 4); }
    GemmArgs base() const {
        GemmArgs a{};
        a.MT = MT; a.B = (int)B; a.stats = g->stats; a.n_chunks = nch; a.K = D;
        a.pos_dev = g->pos_dev; a.D = D; a.H = g->H; a.hd = g->hd; a.Tmax = g->Tmax;
        return a;
    }
    SgArgs sg_base() const {
        SgArgs a{};
        a.pos_dev = g->pos_dev; a.D = D; a.H = g->H; a.Tmax = g->Tmax;
        return a;
    }
    int embed() {
        if (small) {
            SeArgs e{g->tok_emb, g->pos_emb, g->xs, io.tok, io.tok_stride, io.tok_use_pos, g->pos_dev, D};
            g->span_begin(WMAR_T_EMBED, st);
            hipLaunchKernelGGL(k_sembed, dim3((unsigned)((D / 4 + 255) / 256), (unsigned)B), dim3(256), 0, st, e);
            g->span_end(st);
            return launch_status("k_sembed");
        }
        r.x = xcur;
        g->span_begin(WMAR_T_EMBED, st);
        hipLaunchKernelGGL((k_resid_stats<true, 0>), dim3(nch * MT), dim3(256), 0, st, r);
        g->span_end(st);
        return launch_status("k_resid_stats<embed>");
    }
    // x += bias + sum of the S partial slabs; LN statistics of the new rows
    int resid(const float* bias, int S, int n_hi = 0) {
        r.x = xcur;
        r.S = S + (n_hi > 0 ? 1 : 0); r.bias = bias; r.n_hi = n_hi;
        g->span_begin(WMAR_T_RESID, st);
        int rc = launch_resid(r, nch * MT, st);
        g->span_end(st);
        return rc;
    }
    // QKV projection of the RAW residual stream into slabs; LN1, bias and the KV-cache append are
    // finished per (sequence, head) in the attention kernel's prologue.  Measured: one slab (no K
    // split across workgroups) beats every split for this shape -- more, thinner workgroups per CU
    // contend for the same L1 fill path.
    // Layer l's QKV projection with the residual fold of layer l-1's FC2 (bias `prev_bias`, slabs in g->slabs) staged on the
    // fly; prev_bias == nullptr: nothing to fold (layer 0 reads the embedding).  Switches xcur to the other buffer.
    int qkvx(int l, const float* prev_bias) {
        QkvxArgs q{};
        const LayerW& w = g->layers[l];
        float4* xout = (xcur == g->x) ? g->x2 : g->x;
        q.Wp = w.wqkv; q.x_in = xcur; q.x_out = xout;
        q.slabs = g->slabs; q.slab_stride = act;
        const int S_in = prev_bias ? S_fc2 + (fc2_hi > 0 ? 1 : 0) : 0;
        q.n_hi = prev_bias ? fc2_hi : 0; q.bias = prev_bias;
        q.stats = g->stats_q; q.out = g->qkv_slabs; q.out_stride = 3 * act;
        q.KB = KBD; q.NT = 3 * D / 32; q.S = S_qx;
        q.cap = ((q.NT / 4) * q.S + 7) / 8;
        q.nkeep = qx_nkeep();
        q.trace = g->stamp_slot(l, 0);
        g->span_begin(WMAR_T_QKV, st);
        int rc;
        if (qkv_bx()) { q.Wp = w.wqkvx_bx; rc = launch_qkvx_bx(q, S_in, st); }   // 33..64 rows: bf16 matrix pipe
        else rc = launch_qkvx(q, MT, S_in, st);
        g->span_end(st);
        xcur = xout;
        return rc;
    }
    int qkv_small(int l) {
        const SmallW& w = g->sw[l];
        const long long lstride = (long long)g->Bmax * g->H * g->Tmax * g->hd;
        SgArgs a = sg_base();
        a.W = w.wqkv; a.bias = g->layers[l].bqkv; a.c1 = g->layers[l].cqkv; a.x = g->xs; a.gamma = w.ln1w; a.out = g->qs;      // (bias folded: b + W beta)
        a.kcache = g->kcache + l * lstride; a.vcache = g->vcache + l * lstride; a.N = 3 * D; a.K = D;
        g->span_begin(WMAR_T_QKV, st);
        const int rc = launch_sgemv<3, 3, SG_QKV>(a, (int)B, st);
        g->span_end(st);
        return rc;
    }
    int qkv(int l) {
        GemmArgs a = base();
        const LayerW& w = g->layers[l];
        a.Wp = w.wqkv; a.Xp = xcur; a.KB = KBD; a.NT = 3 * D / 32;
        a.out_packed = g->qkv_slabs; a.slab_stride = 3 * act;
        int S = 1;
        g->span_begin(WMAR_T_QKV, st);
        int rc = gemm_split(a, &S, st, S_qkv);
        g->span_end(st);
        return rc;
    }
    int attn(int l) {
        const LayerW& w = g->layers[l];
        const long long lstride = (long long)g->Bmax * g->H * g->Tmax * g->hd;
        if (small) {
            SaArgs s{g->qs, g->kcache + l * lstride, g->vcache + l * lstride, g->ys, g->pos_dev, g->H, g->Tmax, D, 1.0f / sqrtf((float)g->hd)};
            g->span_begin(WMAR_T_ATTN, st);
            hipLaunchKernelGGL(k_sattn, dim3((unsigned)(B * g->H)), dim3(256), 0, st, s);
            g->span_end(st);
            return launch_status("k_sattn");
        }
        AttnArgs t{};
        t.qkv_slabs = g->qkv_slabs; t.slab_stride = 3 * act; t.K = D; t.invK = 1.0 / (double)D;
        if (S_qx > 0) { t.S = S_qx; t.stats = g->stats_q; t.n_chunks = qkv_bx() ? S_qx * qx_nkeep() : S_qx; }
        else { t.S = S_qkv; t.stats = g->stats; t.n_chunks = nch; }
        t.c1 = w.cqkv; t.bias = w.bqkv;
        t.rowmajor = qkv_bx() ? 1 : 0;      // k_qkvx_bx writes its pieces row-major (decoder_kernels.h)
        t.kcache = g->kcache + l * lstride; t.vcache = g->vcache + l * lstride; t.y = g->y; t.yq = proj_bx ? g->yq : nullptr; t.pos_dev = g->pos_dev;
        t.D = D; t.H = g->H; t.Tmax = g->Tmax; t.MT = MT; t.scale = 1.0f / sqrtf((float)g->hd);
        t.trace = g->stamp_slot(l, 1);
        int nwa = g->att_nw;   // chosen per phase by the caller (generate / profile_role)
#ifdef WMAR_DEV_KNOBS
        { const char* e = getenv("WMAR_ATT_DBG"); t.dbg = e ? atoi(e) : 0; }
        { const char* e = getenv("WMAR_ATT_NW"); if (e) nwa = atoi(e); }
#endif
        const dim3 grid((unsigned)(B * g->H));
        g->span_begin(WMAR_T_ATTN, st);
#define WMAR_ATT_LAUNCH3(HDV, PF, RMV)                                                                    \
        if (nwa == 1) hipLaunchKernelGGL((k_attn_decode<HDV, 1, PF, 0, RMV>), grid, dim3(64), 0, st, t);       \
        else if (nwa == 2) hipLaunchKernelGGL((k_attn_decode<HDV, 2, PF, 0, RMV>), grid, dim3(128), 0, st, t); \
        else hipLaunchKernelGGL((k_attn_decode<HDV, 4, PF, 0, RMV>), grid, dim3(256), 0, st, t);
#define WMAR_ATT_LAUNCH2(HDV, PF) if (t.rowmajor) { WMAR_ATT_LAUNCH3(HDV, PF, true) } else { WMAR_ATT_LAUNCH3(HDV, PF, false) }
#define WMAR_ATT_LAUNCH(HDV) if (pf2) { WMAR_ATT_LAUNCH2(HDV, true) } else { WMAR_ATT_LAUNCH2(HDV, false) }
#ifdef WMAR_ATT_PF2
        const bool pf2 = true;
#else
        const bool pf2 = false;    // re-measured with exact waits (round 3) and with non-temporal K/V loads (round 4): +-1 % either way
#endif
        if (g->hd == 64) { WMAR_ATT_LAUNCH(64) }
        else if (g->hd == 32) { WMAR_ATT_LAUNCH(32) }
        else { WMAR_ATT_LAUNCH(128) }
#undef WMAR_ATT_LAUNCH3
#undef WMAR_ATT_LAUNCH2
#undef WMAR_ATT_LAUNCH
        g->span_end(st);
        return launch_status("k_attn_decode");
    }
    // attention output projection (split-K slabs; bias + residual folded by the next resid())
    int proj(int l) {
        if (small) {
            SgArgs a = sg_base();
            a.W = g->sw[l].wproj; a.bias = g->layers[l].bproj; a.x = g->ys; a.out = g->xs; a.N = D; a.K = D;
            g->span_begin(WMAR_T_PROJ, st);
            const int rc = launch_sgemv<3, 1, SG_PROJ>(a, (int)B, st);
            g->span_end(st);
            return rc;
        }
        if (proj_xr) {
            BxrArgs q{};
            BxArgs& x = q.bx;
            x.Wq = g->layers[l].wproj_bx; x.Xq = g->yq; x.out = g->slabs; x.slab_stride = act; x.KU = D / 16; x.S = 4;
            q.x = xcur; q.bias = g->layers[l].bproj; q.stats = g->stats; q.sync = g->xsync; q.fail = g->xsync + 8 * 64;
            q.tiles_per_group = D / 32 / 8;
            q.trace = g->stamp_slot(l, 2);
            g->span_begin(WMAR_T_PROJ, st);
            // round 5: eight waves x 3 steps in phase 1 (its steps are bound by the latency of their loads: 3.817 -> 3.796 ms per step,
            // same-box A/B); WMAR_XR_NW4=1 keeps four waves x 6 steps
            static const bool xr4 = getenv("WMAR_XR_NW4") != nullptr;
            const int rc = xr4 ? launch_bx_xr<BX_PER, 4>(q, D, st) : launch_bx_xr<BX_PER / 2, 4, 8>(q, D, st);
            g->xr_enqueued = true;
            g->span_end(st);
            return rc;
        }
        if (proj_bx) {
            BxArgs x{};
            x.Wq = g->layers[l].wproj_bx; x.Xq = g->yq; x.out = g->slabs; x.slab_stride = act; x.KU = D / 16; x.S = D / BX_KSLICE;
            g->span_begin(WMAR_T_PROJ, st);
            const int rc = launch_bx<1>(x, D, st);
            g->span_end(st);
            return rc;
        }
        GemmArgs p = base();
        p.Wp = g->layers[l].wproj; p.Xp = g->y; p.KB = KBD; p.NT = D / 32;
        p.out_packed = g->slabs; p.slab_stride = act;
        int S = 1;
        g->span_begin(WMAR_T_PROJ, st);
        int rc = gemm_split(p, &S, st, S_proj);
        g->span_end(st);
        return rc;
    }
    // LN2 -> FC1 (+bias, GELU) -> packed hidden
    int fc1(int l) {
        if (small) {
            const SmallW& sw = g->sw[l];
            SgArgs a = sg_base();
            a.W = sw.wfc1; a.bias = g->layers[l].bfc1; a.c1 = g->layers[l].cfc1; a.x = g->xs; a.gamma = sw.ln2w; a.out = g->hs; a.N = 4 * D; a.K = D;
            g->span_begin(WMAR_T_FC1, st);
            const int rc = launch_sgemv<3, 4, SG_FC1>(a, (int)B, st);
            g->span_end(st);
            return rc;
        }
        GemmArgs f = base();
        const LayerW& w = g->layers[l];
        f.Wp = w.wfc1; f.Xp = xcur; f.bias = w.bfc1; f.c1 = w.cfc1; f.KB = KBD; f.NT = 4 * D / 32;
        f.out_packed = g->hbuf; f.slab_stride = 0; f.n_chunks = nch_ln2;
        g->span_begin(WMAR_T_FC1, st);
        int rc;
        if (fc1_x()) {
            Fc1xArgs x{};
            x.W16 = w.wfc1x16; x.W8 = w.wfc1x8; x.Xp = xcur; x.bias = w.bfc1; x.c1 = w.cfc1; x.stats = g->stats; x.n_chunks = nch_ln2; x.K = D;
            x.out = g->hbuf; x.KU = D / 16;
            x.trace = g->stamp_slot(l, 3);
            hipLaunchKernelGGL(k_fc1x, dim3((unsigned)(4 * D / 24)), dim3(256), 0, st, x);
            rc = launch_status("k_fc1x");
        } else {
            rc = gemm_dispatch<EPI_GELU, true>(f, false, st);
        }
        g->span_end(st);
        return rc;
    }
    int fc2(int l) {
        if (small) {
            SgArgs a = sg_base();
            a.W = g->sw[l].wfc2; a.bias = g->layers[l].bfc2; a.x = g->hs; a.out = g->xs; a.N = D; a.K = 4 * D;
            g->span_begin(WMAR_T_FC2, st);
            const int rc = launch_sgemv<3, 2, SG_FC2>(a, (int)B, st);
            g->span_end(st);
            return rc;
        }
        GemmArgs q = base();
        q.Wp = g->layers[l].wfc2; q.Xp = g->hbuf; q.KB = KBF; q.NT = D / 32;
        q.out_packed = g->slabs; q.slab_stride = act; q.n_hi = fc2_hi;
        q.trace = g->stamp_slot(l, 4);
        int S = 1;
        g->span_begin(WMAR_T_FC2, st);
        int rc = gemm_split(q, &S, st, S_fc2);
        g->span_end(st);
        return rc;
    }
    // ln_f -> vocabulary head
    int head() {
        if (small) {
            SgArgs a = sg_base();
            a.W = g->whead_rm; a.bias = g->bhead; a.c1 = g->chead; a.x = g->xs; a.gamma = g->lnfw; a.out = io.logits; a.N = g->V; a.K = D;
            g->span_begin(WMAR_T_HEAD, st);
            const int rc = launch_sgemv<4, 8, SG_HEAD>(a, (int)B, st);
            g->span_end(st);
            return rc;
        }
        GemmArgs h = base();
        h.Wp = g->whead; h.Xp = xcur; h.KB = KBD; h.NT = g->V / 32;
        h.bias = g->bhead; h.c1 = g->chead; h.logits = io.logits; h.V = g->V;
        g->span_begin(WMAR_T_HEAD, st);
        int rc = WMAR_OK;
        if (MT == 2 && h.NT % 2 == 0 && h.NT / 2 <= 512 && !head_narrow()) {
            // 64 rows: TWO column tiles per workgroup (round 4) -- the launch was bound by the bytes a CU pulls (393 KB of activation
            // beside 196 KB of weights per 32-column workgroup, two workgroups per CU: 42 us for 100 MB of weights); every activation
            // fragment now feeds two weight fragments, one launch of V / 64 workgroups
            h.S = 1;
            rc = launch_gemm<2, 4, EPI_LOGITS, true, 0, GEMM_STAGE, true, 2>(h, st);
        } else {
        // at most one workgroup per CU per launch: two workgroups on a CU share its MFMA pipes and L1 fill path, so 512 column
        // tiles run faster as two launches of 256 than as one of 512 (measured 56 us -> see DESIGN.md section 6)
        const int MG = (MT % 2 == 0) ? MT / 2 : MT;
        const int per = 256 / MG > 0 ? 256 / MG : 1;
        const int NTall = h.NT;
        for (int nt0 = 0; nt0 < NTall && rc == WMAR_OK; nt0 += per) {
            GemmArgs p = h;
            p.NT = NTall - nt0 < per ? NTall - nt0 : per;
            p.Wp = h.Wp + (long long)nt0 * h.KB * 64;
            p.bias = h.bias ? h.bias + (long long)nt0 * 32 : nullptr;
            p.c1 = h.c1 + (long long)nt0 * 32;
            p.logits = h.logits + (long long)nt0 * 32;
            rc = gemm_dispatch<EPI_LOGITS, true>(p, false, st);
        }
        }
        g->span_end(st);
        return rc;
    }
};

int enqueue_step(wmar_gpt* g, int64_t B, const StepIO& io, hipStream_t st) {
    StepPlan p(g, B, io, st);
    int rc;
    const long long actb = p.act * 16, Mpad = p.MT * 32;
#ifdef WMAR_DEV_KNOBS
    if (g->dbg_sums) { g->dbg_slot = 0; if (hipMemsetAsync(g->dbg_sums, 0, 4096 * 8, st) != hipSuccess) return WMAR_EHIP; }
#endif
    if (p.persist) {
        SsArgs a{};
        a.arena = g->sw_arena; a.L = g->L; a.tok_emb = g->tok_emb; a.pos_emb = g->pos_emb; a.whead = g->whead_rm; a.lnfw = g->lnfw; a.lnfb = g->lnfb;
        a.tok = io.tok; a.tok_stride = io.tok_stride; a.tok_use_pos = io.tok_use_pos;
        a.xs = g->xs; a.ys = g->ys; a.hs = g->hs; a.qs = g->qs; a.kcache = g->kcache; a.vcache = g->vcache;
        a.lstride = (long long)g->Bmax * g->H * g->Tmax * g->hd; a.logits = io.logits; a.V = g->V; a.pos_dev = g->pos_dev;
        a.D = g->D; a.H = g->H; a.Tmax = g->Tmax; a.bar = g->ss_bar;
        g->span_begin(WMAR_T_QKV, st);
        rc = launch_sstep(a, (int)B, st);
        g->span_end(st);
        g->ps_enqueued = true;
        return rc;
    }
    if ((rc = p.embed())) return rc;
    if (p.small) {
        // 1..12 rows: five streaming launches per layer, the residual stream updated in place (decode_small.h)
        for (int l = 0; l < g->L; ++l) {
            if ((rc = p.qkv_small(l))) return rc;
            if ((rc = p.attn(l))) return rc;
            if ((rc = p.proj(l))) return rc;
            if ((rc = p.fc1(l))) return rc;
            if ((rc = p.fc2(l))) return rc;
        }
        return p.head();
    }
    p.dbg(p.xcur, actb); p.dbg(g->stats, p.nch * Mpad * 16);                                   // slots 0, 1
    for (int l = 0; l < g->L; ++l) {                                                            // then 11 per layer:
        if (p.S_qx > 0) {
            if ((rc = p.qkvx(l, l > 0 ? g->layers[l - 1].bfc2 : nullptr))) return rc;
            p.dbg(p.xcur, actb); p.dbg(g->stats_q, p.S_qx * Mpad * 16); p.dbg(g->qkv_slabs, p.S_qx * 3 * actb);   // +0 x', +1 stats, +2 QKV pieces
        } else {
            if (l > 0 && (rc = p.resid(g->layers[l - 1].bfc2, p.S_fc2, p.fc2_hi))) return rc;
            if ((rc = p.qkv(l))) return rc;
            p.dbg(p.xcur, actb); p.dbg(g->stats, p.nch * Mpad * 16); p.dbg(g->qkv_slabs, 3 * actb);
        }
        if ((rc = p.attn(l))) return rc;
        if (p.proj_bx) p.dbg(g->yq, (long long)p.D / 16 * 2 * 3 * 64 * 16); else p.dbg(g->y, actb);              // +3 attention output
        {
            const long long lstride = (long long)g->Bmax * g->H * g->Tmax * g->hd;
            p.dbg(g->kcache + l * lstride, lstride * 4); p.dbg(g->vcache + l * lstride, lstride * 4);            // +4, +5 caches
        }
        if ((rc = p.proj(l))) return rc;
        p.dbg(g->slabs, p.S_proj * actb);                                                                        // +6 proj slabs
        if (!p.proj_xr && (rc = p.resid(g->layers[l].bproj, p.S_proj))) return rc;             // k_bx_xr has folded and summed already
        p.dbg(p.xcur, actb); p.dbg(g->stats, p.nch_ln2 * Mpad * 16);                                             // +7, +8
        if ((rc = p.fc1(l))) return rc;
        p.dbg(g->hbuf, 4 * actb);                                                                                // +9 hidden
        if ((rc = p.fc2(l))) return rc;
        p.dbg(g->slabs, p.S_fc2 * actb);                                                                         // +10 FC2 slabs (uniform part)
    }
    if ((rc = p.resid(g->layers[g->L - 1].bfc2, p.S_fc2, p.fc2_hi))) return rc;
    p.dbg(p.xcur, actb); p.dbg(g->stats, p.nch * Mpad * 16);
    return p.head();
}

}  // namespace

extern "C" {

int wmar_gpt_create(const wmar_gpt_config* cfg, const char* const* names, const void* const* tensors_dev,
                    int32_t n_tensors, void* stream, wmar_gpt** out) {
    WMAR_REQUIRE(cfg && names && tensors_dev && out, "gpt_create: null argument");
    WMAR_REQUIRE(cfg->n_embd % cfg->n_head == 0, "n_embd %% n_head != 0");
    const int D = cfg->n_embd, H = cfg->n_head, hd = D / H, V = cfg->vocab_size, L = cfg->n_layer;
    WMAR_REQUIRE(D % 32 == 0 && V % 32 == 0 && D <= 8192, "n_embd (<= 8192) and vocab_size must be multiples of 32 (got %d, %d)", D, V);
    WMAR_REQUIRE(hd == 32 || hd == 64 || hd == 128, "head_dim %d unsupported (32, 64, 128)", hd);
    WMAR_REQUIRE(cfg->max_batch >= 1 && cfg->max_batch <= 128, "max_batch must be in 1..128");
    WMAR_REQUIRE(cfg->block_size >= 1 && L >= 1, "bad block_size / n_layer");
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) tm.m[names[i]] = tensors_dev[i];
    hipStream_t st = (hipStream_t)stream;
    auto* g = new wmar_gpt();
    g->cfg = *cfg; g->D = D; g->H = H; g->hd = hd; g->V = V; g->L = L; g->Tmax = cfg->block_size; g->Bmax = cfg->max_batch;
    g->MTmax = mt_for(cfg->max_batch);
#ifdef WMAR_DEV_KNOBS
    {
        const char* names_s[3] = {"WMAR_S_QKV", "WMAR_S_PROJ", "WMAR_S_FC2"};
        for (int i = 0; i < 3; ++i) {
            const char* e = getenv(names_s[i]);
            if (e) g->force_s[i] = atoi(e) > MAX_SLABS ? MAX_SLABS : atoi(e);
        }
    }
#endif
    // WMAR_NO_BX=1 at engine creation (any build): QKV and the output projection stay on the fp32-input MFMA (k_qkvx / k_gemm).  The
    // bf16-piece split turns an infinite operand into NaN where fp32 arithmetic gives +-inf (bx_split.h).
    g->no_bx = getenv("WMAR_NO_BX") != nullptr;
#ifdef WMAR_STAMPS
    if (getenv("WMAR_STAMPS")) {
        const size_t n = (size_t)L * 5 * WMAR_STAMP_UNITS * 8;
        if (hipMalloc(&g->stamps, n * 8) != hipSuccess || hipMemset(g->stamps, 0, n * 8) != hipSuccess) g->stamps = nullptr;
    }
#endif
#ifdef WMAR_DEV_KNOBS
    if (getenv("WMAR_DBG_SUMS")) { if (hipMalloc(&g->dbg_sums, 4096 * 8) != hipSuccess) g->dbg_sums = nullptr; }
    g->no_bx_qkv = getenv("WMAR_NO_BX_QKV") != nullptr; g->no_bx_proj = getenv("WMAR_NO_BX_PROJ") != nullptr;   // one role at a time (bisecting)
#endif
    int rc = WMAR_OK;
    auto need = [&](const std::string& k) -> const float* {
        const float* p = tm.get(k);
        if (!p && rc == WMAR_OK) { set_error("checkpoint tensor '%s' is missing", k.c_str()); rc = WMAR_EMISSING; }
        return p;
    };
#define TRY(x) do { if (rc == WMAR_OK) rc = (x); } while (0)
    const float* te = need("tok_emb.weight");
    const float* pe = need("pos_emb");
    const float* hw = need("head.weight");
    const float *lfw = need("ln_f.weight"), *lfb = need("ln_f.bias");
    if (rc == WMAR_OK) {
        TRY(copy_vec(g, &g->tok_emb, te, (size_t)V * D, st));
        TRY(copy_vec(g, &g->pos_emb, pe, (size_t)cfg->block_size * D, st));
        TRY(g->alloc(&g->whead, (size_t)V * D / 4));
        TRY(pack(hw, g->whead, V, D, 0, st, lfw));
        TRY(g->alloc(&g->bhead, (size_t)V));
        TRY(fold_bias(hw, nullptr, lfb, g->bhead, V, D, st));
        TRY(g->alloc(&g->chead, (size_t)V));
        TRY(fold_bias(hw, nullptr, lfw, g->chead, V, D, st));
    }
    g->layers.resize(L);
    for (int l = 0; l < L && rc == WMAR_OK; ++l) {
        std::string p = "blocks." + std::to_string(l) + ".";
        LayerW& w = g->layers[l];
        const float *qw = need(p + "attn.query.weight"), *kw = need(p + "attn.key.weight"), *vw = need(p + "attn.value.weight");
        const float *qb = need(p + "attn.query.bias"), *kb = need(p + "attn.key.bias"), *vb = need(p + "attn.value.bias");
        const float *pw = need(p + "attn.proj.weight"), *pb = need(p + "attn.proj.bias");
        const float *f1w = need(p + "mlp.0.weight"), *f1b = need(p + "mlp.0.bias");
        const float *f2w = need(p + "mlp.2.weight"), *f2b = need(p + "mlp.2.bias");
        const float *l1w = need(p + "ln1.weight"), *l1b = need(p + "ln1.bias"), *l2w = need(p + "ln2.weight"), *l2b = need(p + "ln2.bias");
        if (rc != WMAR_OK) break;
        TRY(g->alloc(&w.wqkv, (size_t)3 * D * D / 4));
        TRY(pack(qw, w.wqkv, D, D, 0, st, l1w));
        TRY(pack(kw, w.wqkv, D, D, D / 32, st, l1w));
        TRY(pack(vw, w.wqkv, D, D, 2 * D / 32, st, l1w));
        if (g->MTmax >= 2 && D % 16 == 0 && (3 * D / 32) % 4 == 0) {
            auto pack_bx = [&](const float* W, int tile_off) -> int {
                const long long total = (long long)(D / 32) * (D / 16) * 128;
                hipLaunchKernelGGL(k_pack_bx, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, l1w, w.wqkvx_bx, D, D, tile_off);
                return launch_status("k_pack_bx");
            };
            TRY(g->alloc(&w.wqkvx_bx, (size_t)3 * D * D / 4));
            TRY(pack_bx(qw, 0));
            TRY(pack_bx(kw, D / 32));
            TRY(pack_bx(vw, 2 * D / 32));
        }
        TRY(g->alloc(&w.bqkv, (size_t)3 * D));
        TRY(fold_bias(qw, qb, l1b, w.bqkv, D, D, st));
        TRY(fold_bias(kw, kb, l1b, w.bqkv + D, D, D, st));
        TRY(fold_bias(vw, vb, l1b, w.bqkv + 2 * D, D, D, st));
        TRY(g->alloc(&w.cqkv, (size_t)3 * D));
        TRY(fold_bias(qw, nullptr, l1w, w.cqkv, D, D, st));
        TRY(fold_bias(kw, nullptr, l1w, w.cqkv + D, D, D, st));
        TRY(fold_bias(vw, nullptr, l1w, w.cqkv + 2 * D, D, D, st));
        TRY(g->alloc(&w.wproj, (size_t)D * D / 4));
        TRY(pack(pw, w.wproj, D, D, 0, st));
        TRY(copy_vec(g, &w.bproj, pb, D, st));
        if (g->MTmax >= 2 && D % BX_KSLICE == 0 && D / BX_KSLICE <= MAX_SLABS) {
            const long long total = (long long)(D / 32) * (D / 16) * 128;
            TRY(g->alloc(&w.wproj_bx, (size_t)D * D / 4));
            if (rc == WMAR_OK) {
                hipLaunchKernelGGL(k_pack_bx, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pw, (const float*)nullptr, w.wproj_bx, D, D, 0);
                rc = launch_status("k_pack_bx");
            }
        }
        TRY(g->alloc(&w.wfc1, (size_t)4 * D * D / 4));
        TRY(pack(f1w, w.wfc1, 4 * D, D, 0, st, l2w));
        if ((4 * D) % 24 == 0 && (D / 16) % 16 == 0 && g->MTmax >= 2) {
            // batches of 33..64 rows run FC1 on 24-column tiles (k_fc1x: one workgroup per CU at n_embd 1536); the 32-column
            // packing above serves the other batch sizes
            const size_t units = (size_t)(4 * D / 24) * (D / 16) * 64;
            TRY(g->alloc(&w.wfc1x16, units));
            TRY(g->alloc(&w.wfc1x8, units));
            if (rc == WMAR_OK) {
                hipLaunchKernelGGL(k_pack_fc1x, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, f1w, l2w, w.wfc1x16, w.wfc1x8, 4 * D, D);
                rc = launch_status("k_pack_fc1x");
            }
        }
        TRY(g->alloc(&w.bfc1, (size_t)4 * D));
        TRY(fold_bias(f1w, f1b, l2b, w.bfc1, 4 * D, D, st));
        TRY(g->alloc(&w.cfc1, (size_t)4 * D));
        TRY(fold_bias(f1w, nullptr, l2w, w.cfc1, 4 * D, D, st));
        TRY(g->alloc(&w.wfc2, (size_t)4 * D * D / 4));
        TRY(pack(f2w, w.wfc2, D, 4 * D, 0, st));
        TRY(copy_vec(g, &w.bfc2, f2b, D, st));
    }
    // small-batch path (decode_small.h): the weights once more in the checkpoint's row-major layout (LayerNorm not folded), 5.5 GB at
    // 48 layers x 1536.  Shapes: n_embd = two K segments of 768, head_dim 64.  WMAR_NO_SMALL=1 at creation keeps 1..12 rows on the
    // matrix-core plan (A/B, tests).
    if (rc == WMAR_OK && D == 2 * SG_SEG && hd == 64 && !getenv("WMAR_NO_SMALL")) {
        g->sw.resize(L);
        const size_t DD = (size_t)D * D;
        // ONE allocation, fixed per-layer stride (decode_persist.h forms every address from this base)
        TRY(g->alloc(&g->sw_arena, (size_t)L * SS_LAYER_FLOATS));
        for (int l = 0; l < L && rc == WMAR_OK; ++l) {
            std::string p = "blocks." + std::to_string(l) + ".";
            SmallW& w = g->sw[l];
            float* base = g->sw_arena + (size_t)l * SS_LAYER_FLOATS;
            w.wqkv = base + SS_OFF_WQKV; w.wproj = base + SS_OFF_WPROJ; w.wfc1 = base + SS_OFF_WFC1; w.wfc2 = base + SS_OFF_WFC2;
            w.bqkv = base + SS_OFF_BQKV; w.bproj = base + SS_OFF_BPROJ; w.bfc1 = base + SS_OFF_BFC1; w.bfc2 = base + SS_OFF_BFC2;
            w.ln1w = base + SS_OFF_LN1W; w.ln1b = base + SS_OFF_LN1B; w.ln2w = base + SS_OFF_LN2W; w.ln2b = base + SS_OFF_LN2B;
            hipError_t e = hipSuccess;
            auto cp = [&](float* dst, const std::string& key, size_t n) {
                if (e == hipSuccess) e = hipMemcpyAsync(dst, tm.get(p + key), n * 4, hipMemcpyDeviceToDevice, st);
            };
            cp(w.wqkv, "attn.query.weight", DD); cp(w.wqkv + DD, "attn.key.weight", DD); cp(w.wqkv + 2 * DD, "attn.value.weight", DD);
            cp(w.bqkv, "attn.query.bias", D); cp(w.bqkv + D, "attn.key.bias", D); cp(w.bqkv + 2 * D, "attn.value.bias", D);
            cp(w.wproj, "attn.proj.weight", DD); cp(w.bproj, "attn.proj.bias", D);
            cp(w.wfc1, "mlp.0.weight", 4 * DD); cp(w.bfc1, "mlp.0.bias", (size_t)4 * D);
            cp(w.wfc2, "mlp.2.weight", 4 * DD); cp(w.bfc2, "mlp.2.bias", D);
            cp(w.ln1w, "ln1.weight", D); cp(w.ln1b, "ln1.bias", D); cp(w.ln2w, "ln2.weight", D); cp(w.ln2b, "ln2.bias", D);
            if (e != hipSuccess) { set_error("gpt_create: %s", hipGetErrorString(e)); rc = WMAR_EHIP; }
        }
        TRY(copy_vec(g, &g->whead_rm, hw, (size_t)V * D, st));
        TRY(copy_vec(g, &g->lnfw, lfw, D, st));
        TRY(copy_vec(g, &g->lnfb, lfb, D, st));
        TRY(g->alloc(&g->xs, (size_t)SG_MAX_ROWS * D));
        TRY(g->alloc(&g->ys, (size_t)SG_MAX_ROWS * D));
        TRY(g->alloc(&g->hs, (size_t)SG_MAX_ROWS * 4 * D));
        TRY(g->alloc(&g->qs, (size_t)SG_MAX_ROWS * D));
        g->small_ok = rc == WMAR_OK;
        // the persistent step: per-layer pointer table, barrier words; all 256 workgroups of 320 threads must be resident at once and
        // workgroups with equal blockIdx % 8 must share an XCD (probed below with the kernel's own grid).  WMAR_NO_PERSIST=1: off.
        static_assert(SSD == 2 * SG_SEG, "arena offsets are for n_embd 1536");
        // OPT-IN (WMAR_PERSIST=1 at creation): measured SLOWER than the five launches it replaces (round 6, DESIGN section 6a: 2.01 against
        // 1.63 ms per step at batch 1, 2.85 against 1.89 at batch 5) -- the barrier's atomics and polls travel through the same in-order
        // per-CU memory pipeline as the weight prefetch they are meant to overlap with; kept as a tested experiment, not the default plan.
        if (g->small_ok && getenv("WMAR_PERSIST") && !getenv("WMAR_NO_PERSIST")) {
            TRY(g->alloc(&g->ss_bar, (size_t)SS_BAR_WORDS));
            if (rc == WMAR_OK && hipMemsetAsync(g->ss_bar, 0, SS_BAR_WORDS * 4, st) != hipSuccess) { set_error("gpt_create: memset failed"); rc = WMAR_EHIP; }
            if (rc == WMAR_OK) {
                int dev = 0, cus = 0;
                bool ok = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
                ok = ok && (long long)sstep_blocks_per_cu() * cus >= SS_WGS;
                unsigned* tmp = nullptr;
                ok = ok && hipMalloc(&tmp, SS_WGS * 4) == hipSuccess;
                for (int rep = 0; rep < 3 && ok; ++rep) {
                    unsigned h[SS_WGS];
                    hipLaunchKernelGGL(k_xcc_probe, dim3(SS_WGS), dim3(SS_THREADS), 0, st, tmp);
                    ok = hipMemcpyAsync(h, tmp, sizeof h, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
                    unsigned seen = 0;
                    for (int b = 0; b < SS_WGS && ok; ++b) ok = h[b] == h[b & 7] && h[b] < 16;
                    for (int x = 0; x < 8 && ok; ++x) { ok = !(seen & (1u << h[x])); seen |= 1u << h[x]; }
                }
                if (tmp) (void)hipFree(tmp);
                g->persist_ok = ok;
            }
        }
    }
    const size_t Mpad = (size_t)g->MTmax * 32;
    TRY(g->alloc(&g->x, Mpad * D / 4));
    TRY(g->alloc(&g->x2, Mpad * D / 4));
    TRY(g->alloc(&g->y, Mpad * D / 4));
    TRY(g->alloc(&g->hbuf, Mpad * 4 * D / 4));
    TRY(g->alloc(&g->slabs, (size_t)MAX_SLABS * Mpad * D / 4));
    TRY(g->alloc(&g->qkv_slabs, (size_t)MAX_SLABS * Mpad * 3 * D / 4));
    if (g->MTmax >= 2 && D % BX_KSLICE == 0) TRY(g->alloc(&g->yq, (size_t)D / 16 * 2 * 3 * 64));
    TRY(g->alloc(&g->qbuf, Mpad * D));
    TRY(g->alloc(&g->stats, (size_t)STAT_CHUNKS_MAX * Mpad * 2));
    TRY(g->alloc(&g->stats_q, (size_t)QKV_SLABS_MAX * 4 * Mpad * 2));      // x 4: k_qkvx_bx's keeper groups per K slice
    const size_t kv = (size_t)L * g->Bmax * H * g->Tmax * hd;
    TRY(g->alloc(&g->kcache, kv));
    TRY(g->alloc(&g->vcache, kv));
    TRY(g->alloc(&g->logits, (size_t)g->Bmax * V));
    TRY(g->alloc(&g->scratch, (size_t)g->Bmax * V));
    TRY(g->alloc(&g->past, (size_t)g->Bmax * (g->Tmax + 1)));
    TRY(g->alloc(&g->pos_dev, 4));
    TRY(g->alloc(&g->xsync, 8 * 64 + 64));
    g->step_dev = g->pos_dev + 1;
    if (rc == WMAR_OK) {
        // padded rows of the packed buffers must hold finite numbers
        hipError_t e = hipMemsetAsync(g->x, 0, Mpad * D * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->x2, 0, Mpad * D * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->kcache, 0, kv * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->vcache, 0, kv * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->hbuf, 0, Mpad * 4 * D * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->past, 0, (size_t)g->Bmax * (g->Tmax + 1) * 8, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->y, 0, Mpad * D * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->qbuf, 0, Mpad * D * 4, st);
        if (e == hipSuccess && g->yq) e = hipMemsetAsync(g->yq, 0, (size_t)D / 16 * 2 * 3 * 64 * 16, st);   // rows past the batch are never written
        if (e == hipSuccess) e = hipMemsetAsync(g->xsync, 0, (8 * 64 + 64) * 4, st);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->cap_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&g->ev0);
        if (e == hipSuccess) e = hipEventCreate(&g->ev1);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("gpt_create: %s", hipGetErrorString(e)); rc = WMAR_EHIP; }
    }
#undef TRY
    if (rc == WMAR_OK && g->MTmax >= 2 && g->yq) {
        // k_bx_xr keeps a split-K reduction inside an XCD: probe that the 24 blocks of a 192-block grid with equal blockIdx % 8
        // share an XCC id and that the eight groups land on eight different XCDs (three launches: the id a group gets rotates with
        // the launches before it, the grouping must not).  Anything else keeps the two-launch path.
        g->no_xr = getenv("WMAR_NO_XR") != nullptr;
        unsigned* tmp = nullptr;
        bool ok = hipMalloc(&tmp, 192 * 4) == hipSuccess;
        for (int rep = 0; rep < 3 && ok; ++rep) {
            unsigned h[192];
            hipLaunchKernelGGL(k_xcc_probe, dim3(192), dim3(256), 0, st, tmp);
            ok = hipMemcpyAsync(h, tmp, sizeof h, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            unsigned seen = 0;
            for (int b = 0; b < 192 && ok; ++b) ok = h[b] == h[b & 7] && h[b] < 16;
            for (int x = 0; x < 8 && ok; ++x) { ok = !(seen & (1u << h[x])); seen |= 1u << h[x]; }
        }
        if (tmp) (void)hipFree(tmp);
        // ... and every workgroup of its grid must be resident at once: one per CU at least (a CU mask or a partition mode shrinks
        // what the runtime reports)
        // (the instantiation that will run: WMAR_XR_NW4=1 selects the four-wave variant, another register / LDS footprint)
        int nb = 0, dev = 0, cus = 0;
        const bool xr4 = getenv("WMAR_XR_NW4") != nullptr;
        if (ok) ok = (xr4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bx_xr<BX_PER, 4>, 256, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bx_xr<BX_PER / 2, 4, 8>, 512, 0)) == hipSuccess && hipGetDevice(&dev) == hipSuccess &&
                     hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                     (long long)nb * cus >= (long long)(D / 32) * 4;
        g->xcd_ok = ok;
        { const char* e = getenv("WMAR_INJECT_SYNC_FAIL"); g->inject_fail = (e && atoi(e) > 0) ? 1 : 0; }
    }
    if (rc != WMAR_OK) { delete g; return rc; }
    *out = g;
    return WMAR_OK;
}

// The fused projection launch (k_bx_xr) raises device flags when its XCD-local barrier gives up or a block sits on a foreign XCD.
// Returns 1 when a flag was up: the engine has then been switched to the two-launch path (k_bx + k_resid_stats), its graphs dropped
// and the flags cleared -- the caller re-runs its work, which is deterministic in its inputs.  0: clean.  < 0: HIP error.
// the persistent step's barrier (decode_persist.h): 1 when a wait gave up -- the engine is switched to the five-launch plan, its graphs
// dropped, the barrier words cleared; the caller re-runs its work
static int gpt_persist_failed(wmar_gpt* g, hipStream_t st) {
    if (!g->ss_bar || !g->persist_ok || !g->ps_enqueued) return 0;
    g->ps_enqueued = false;
    unsigned f = 0;
    WMAR_HIP_CHECK(hipMemcpyAsync(&f, g->ss_bar + 8 * 64 + 64, 4, hipMemcpyDeviceToHost, st));
    WMAR_HIP_CHECK(hipStreamSynchronize(st));
    if (!f) return 0;
    g->drop_graph();
    WMAR_HIP_CHECK(hipMemsetAsync(g->ss_bar, 0, SS_BAR_WORDS * 4, st));
    g->persist_ok = false;
    g->fallbacks += 1;
    fprintf(stderr, "wmar_amd: the persistent decode step (k_sstep) gave up waiting at its device-wide barrier; this engine continues on the "
                    "five-launch small-batch plan (the call is re-run; wmar_gpt_plan_info reports barrier_fallbacks)\n");
    return 1;
}
static int gpt_sync_failed(wmar_gpt* g, hipStream_t st) {
    if (const int pf = gpt_persist_failed(g, st)) return pf;
    // only a fused launch can raise the flags: calls that enqueued none (two-launch path, WMAR_NO_XR, <= 32 rows, the small-batch
    // path) stay asynchronous and legal inside a caller's stream capture
    if (!g->xsync || !g->xcd_ok || g->no_xr || !g->xr_enqueued) return 0;
    g->xr_enqueued = false;
    unsigned f[2] = {0, 0};
    WMAR_HIP_CHECK(hipMemcpyAsync(f, g->xsync + 8 * 64, 8, hipMemcpyDeviceToHost, st));
    WMAR_HIP_CHECK(hipStreamSynchronize(st));
    if (!f[0] && !f[1]) return 0;
    g->drop_graph();
    WMAR_HIP_CHECK(hipMemsetAsync(g->xsync, 0, (8 * 64 + 64) * 4, st));
    g->xcd_ok = false;                  // the two-launch path from here on
    g->fallbacks += 1;
    fprintf(stderr, "wmar_amd: the fused projection launch (k_bx_xr) %s; this engine continues on the two-launch path (the call is re-run; "
                    "wmar_gpt_plan_info reports barrier_fallbacks)\n", f[0] ? "found a workgroup on a foreign XCD" : "gave up waiting at its XCD-local barrier");
    return 1;
}
// tests: raise the timeout flag in front of the next fused call, once
static int gpt_inject(wmar_gpt* g, hipStream_t st) {
    if (!g->inject_fail || !g->xcd_ok) return WMAR_OK;
    g->inject_fail = 0;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, (int*)(g->xsync + 8 * 64 + 1), 1);
    return launch_status("k_set_int");
}
int wmar_gpt_check(wmar_gpt* g, void* stream) {
    WMAR_REQUIRE(g, "gpt_check: null argument");
    // generate / decode_step verify and recover by themselves; this entry point remains for callers that enqueue single roles
    // (wmar_gpt_profile_role) and want to know whether the fused launch still runs
    const int f = gpt_sync_failed(g, (hipStream_t)stream);
    if (f < 0) return f;
    if (f > 0) {
        set_error("gpt: the XCD-local barrier of the fused projection launch gave up or found a block on a foreign XCD: the launches since "
                  "the last check are invalid (the engine continues on the two-launch path)");
        return WMAR_EHIP;
    }
    return WMAR_OK;
}

void wmar_gpt_destroy(wmar_gpt* g) { delete g; }
int64_t wmar_gpt_device_bytes(const wmar_gpt* g) { return g ? g->bytes : 0; }

int wmar_gpt_decode_step(wmar_gpt* g, const int64_t* tok_dev, int64_t B, int32_t pos, float* logits_dev, void* stream) {
    WMAR_REQUIRE(g && tok_dev && logits_dev, "decode_step: null argument");
    WMAR_REQUIRE(B >= 1 && B <= g->Bmax, "decode_step: batch %lld outside 1..%d", (long long)B, g->Bmax);
    WMAR_REQUIRE(pos >= 0 && pos < g->Tmax, "decode_step: position %d outside the block size %d", pos, g->Tmax);
    hipStream_t st = (hipStream_t)stream;
    StepIO io{(const long long*)tok_dev, 1, 0, logits_dev};
    g->att_nw = wmar_gpt::phase_waves(g->att_phase(pos + 1, B));
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (int rc = gpt_inject(g, st)) return rc;
        hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, g->pos_dev, (int)pos);
        if (int rc = enqueue_step(g, B, io, st)) return rc;
        // on the fused path the call waits for the step and checks the in-launch barrier; a failed step is re-run on the two-launch
        // path (the step is a pure function of the token, the position and the cache rows below it)
        const int f = gpt_sync_failed(g, st);
        if (f < 0) return f;
        if (f == 0) return WMAR_OK;
    }
    set_error("decode_step: the in-launch barrier flag is up on the two-launch path");
    return WMAR_EHIP;
}

int wmar_gpt_profile_role(wmar_gpt* g, int32_t role, int64_t B, int32_t kv_len, int32_t iters, void* stream,
                          double* avg_us) {
    WMAR_REQUIRE(g && avg_us && iters > 0, "profile_role: bad argument");
    WMAR_REQUIRE(B >= 1 && B <= g->Bmax, "profile_role: batch outside 1..%d", g->Bmax);
    WMAR_REQUIRE(kv_len >= 1 && kv_len <= g->Tmax, "profile_role: kv_len outside 1..%d", g->Tmax);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, g->pos_dev, (int)kv_len - 1);
    StepIO io{g->past, (long long)g->Tmax + 1, 0, g->logits};
    g->att_nw = wmar_gpt::phase_waves(g->att_phase(kv_len, B));
    StepPlan p(g, B, io, st);
    // dev knob: WMAR_PROFILE_LAYERS=n cycles through the first n layers only (n = 1: weights stay in the memory-side cache)
    int ncycle = g->L;
#ifdef WMAR_DEV_KNOBS
    { const char* pl = getenv("WMAR_PROFILE_LAYERS"); if (pl && atoi(pl) > 0) ncycle = std::min(atoi(pl), g->L); }
#endif
    auto one = [&](int it) -> int {
        const int l = it % ncycle;
        switch (role) {
            case WMAR_T_EMBED: return p.embed();
            case WMAR_T_QKV: if (p.small) return p.qkv_small(l); p.xcur = g->x; return p.S_qx > 0 ? p.qkvx(l, g->layers[l].bfc2) : p.qkv(l);
            case WMAR_T_ATTN: return p.attn(l);
            case WMAR_T_PROJ: return p.proj(l);
            case WMAR_T_RESID: if (p.small) { set_error("profile_role: the small-batch path has no residual launch"); return WMAR_EINVAL; } return p.resid(g->layers[l].bproj, p.S_proj);
            case WMAR_T_FC1: return p.fc1(l);
            case WMAR_T_FC2: return p.fc2(l);
            case WMAR_T_HEAD: return p.head();
            default: set_error("profile_role: unknown role %d", role); return WMAR_EINVAL;
        }
    };
    const bool was = g->span_on;
    g->span_on = false;
    int rc = WMAR_OK;
    for (int i = 0; i < 3 && rc == WMAR_OK; ++i) rc = one(i);           // warm-up
    if (rc == WMAR_OK && hipEventRecord(g->ev0, st) != hipSuccess) rc = WMAR_EHIP;
    for (int i = 0; i < iters && rc == WMAR_OK; ++i) rc = one(i + 3);
    if (rc == WMAR_OK && hipEventRecord(g->ev1, st) != hipSuccess) rc = WMAR_EHIP;
    g->span_on = was;
    if (rc) return rc;
    WMAR_HIP_CHECK(hipEventSynchronize(g->ev1));
    float ms = 0.f;
    WMAR_HIP_CHECK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    *avg_us = (double)ms * 1000.0 / iters;
    return WMAR_OK;
}

int wmar_gpt_plan_info(wmar_gpt* g, int64_t B, char* buf, int64_t buf_len) {
    WMAR_REQUIRE(g && buf && buf_len > 0, "plan_info: null argument");
    WMAR_REQUIRE(B >= 1 && B <= g->Bmax, "plan_info: batch outside 1..%d", g->Bmax);
    StepIO io{g->past, (long long)g->Tmax + 1, 0, g->logits};
    StepPlan p(g, B, io, nullptr);
    if (p.persist) {
        const int n = snprintf(buf, (size_t)buf_len,
                               "qkv=k_sstep phase (the arithmetic of k_sgemv<QKV>, weights streamed across the barriers);attn=k_sstep phase;proj=k_sstep phase;resid=none;"
                               "resid_launches_per_step=0;fc1=k_sstep phase;fc2=k_sstep phase;head=k_sstep phase;barrier_fallbacks=%d;"
                               "path=persistent step: one launch, %d device-wide barriers (decode_persist.h)", g->fallbacks, 5 * g->L + 1);
        WMAR_REQUIRE(n > 0 && n < buf_len, "plan_info: buffer of %lld bytes too small", (long long)buf_len);
        return WMAR_OK;
    }
    if (p.small) {
        const int n = snprintf(buf, (size_t)buf_len,
                               "qkv=k_sgemv<QKV> (row-major weight stream, LN1 + bias + cache append inside, %d columns per workgroup);"
                               "attn=k_sattn (4 waves per (sequence, head));proj=k_sgemv<PROJ> (residual added in place);resid=none;"
                               "resid_launches_per_step=0;fc1=k_sgemv<FC1> (LN2 + bias + GELU inside);fc2=k_sgemv<FC2> (8 K segments, residual added in place);"
                               "head=k_sgemv<HEAD> (ln_f inside);barrier_fallbacks=%d;path=small-batch streaming (decode_small.h)",
                               sg_cols_wg<3, 3, SG_QKV>(), g->fallbacks);
        WMAR_REQUIRE(n > 0 && n < buf_len, "plan_info: buffer of %lld bytes too small", (long long)buf_len);
        return WMAR_OK;
    }
    const LayerW& w = g->layers[0];
    char qkv[96], proj[160], fc1[96], fc2[96];
    const int S_in = p.S_fc2 + (p.fc2_hi > 0 ? 1 : 0);
    (void)w;
    if (p.qkv_bx()) snprintf(qkv, sizeof qkv, "k_qkvx_bx<%d> (bf16 pipe, %d K slices)", S_in, p.S_qx);
    else if (p.S_qx > 0) snprintf(qkv, sizeof qkv, "k_qkvx<%d,%d> (fp32 MFMA, %d K slices)", p.MT, S_in, p.S_qx);
    else snprintf(qkv, sizeof qkv, "k_gemm<EPI_PACKED> (fp32 MFMA, %d K slices) after k_resid_stats", p.S_qkv);
    if (p.proj_xr) snprintf(proj, sizeof proj, "k_bx_xr<%d,4> (bf16 pipe, 4 K slices + XCD-local reduction: residual fold and LN2 statistics inside)", BX_PER);   // (eight waves x BX_PER / 2 steps since round 5)
    else if (p.proj_bx) snprintf(proj, sizeof proj, "k_bx<1,%d> (bf16 pipe, %d K slices)", BX_PER, p.S_proj);
    else snprintf(proj, sizeof proj, "k_gemm<EPI_PACKED> (fp32 MFMA, %d K slices)", p.S_proj);
    if (p.fc1_x()) snprintf(fc1, sizeof fc1, "k_fc1x (fp32 MFMA, 24-column tiles, whole K)");
    else snprintf(fc1, sizeof fc1, "k_gemm<EPI_GELU,LN> (fp32 MFMA, whole K)");
    snprintf(fc2, sizeof fc2, "k_gemm<EPI_PACKED> (fp32 MFMA, %d%s K slices)", p.S_fc2, p.fc2_hi > 0 ? "/+1" : "");
    // attention: the waves per (sequence, head) follow the cache length (att_phase); reported at 1, block_size / 2 and block_size rows
    char attn[96];
    const int w0 = wmar_gpt::phase_waves(g->att_phase(1, B)), w1 = wmar_gpt::phase_waves(g->att_phase(g->Tmax / 2, B)),
              w2 = wmar_gpt::phase_waves(g->att_phase(g->Tmax, B));
    if (w0 == w1 && w1 == w2) snprintf(attn, sizeof attn, "k_attn_decode<%d,%d>", g->hd, w1);
    else snprintf(attn, sizeof attn, "k_attn_decode<%d,%d> (%d / %d / %d waves at 1 / %d / %d cached rows)", g->hd, w1, w0, w1, w2, g->Tmax / 2, g->Tmax);
    const int n = snprintf(buf, (size_t)buf_len, "qkv=%s;attn=%s;proj=%s;resid=k_resid_stats;resid_launches_per_step=%d;fc1=%s;fc2=%s;head=k_gemm<EPI_LOGITS,LN> (fp32 MFMA);barrier_fallbacks=%d",
                           qkv, attn, proj, p.proj_xr ? 1 : g->L + 1, fc1, fc2, g->fallbacks);
    WMAR_REQUIRE(n > 0 && n < buf_len, "plan_info: buffer of %lld bytes too small", (long long)buf_len);
    return WMAR_OK;
}

#ifdef WMAR_STAMPS
// dev only: the timeline stamps of the LAST decode step that ran, [L][5][WMAR_STAMP_UNITS][8] u64 copied to the host
int wmar_gpt_debug_stamps(wmar_gpt* g, unsigned long long* out, long long n_u64) {
    WMAR_REQUIRE(g && g->stamps && out, "debug_stamps: not enabled (WMAR_STAMPS=1 at creation of a -DWMAR_STAMPS build)");
    const long long n = (long long)g->L * 5 * WMAR_STAMP_UNITS * 8;
    WMAR_HIP_CHECK(hipDeviceSynchronize());
    WMAR_HIP_CHECK(hipMemcpy(out, g->stamps, (size_t)(n_u64 < n ? n_u64 : n) * 8, hipMemcpyDeviceToHost));
    return WMAR_OK;
}
#endif

#ifdef WMAR_DEV_KNOBS
// dev only: the checksums of the last decode step (k_dbg_sum), n_out slots copied to the host
int wmar_gpt_debug_sums(wmar_gpt* g, unsigned long long* out, int n_out, int* n_used) {
    WMAR_REQUIRE(g && g->dbg_sums && out, "debug_sums: not enabled (WMAR_DBG_SUMS=1 at creation)");
    WMAR_HIP_CHECK(hipDeviceSynchronize());
    WMAR_HIP_CHECK(hipMemcpy(out, g->dbg_sums, (size_t)(n_out < 4096 ? n_out : 4096) * 8, hipMemcpyDeviceToHost));
    if (n_used) *n_used = g->dbg_slot;
    return WMAR_OK;
}
#endif

#ifdef WMAR_DEV_KNOBS
// dev only: the small-batch plan's exchange buffers after the last step: which = 0 xs, 1 qs, 2 ys, 3 hs (4 D wide) -> out[8][width]
int wmar_gpt_debug_small(wmar_gpt* g, int which, float* out) {
    WMAR_REQUIRE(g && g->small_ok && out && which >= 0 && which <= 3, "debug_small: bad argument");
    WMAR_HIP_CHECK(hipDeviceSynchronize());
    const float* src = which == 0 ? g->xs : which == 1 ? g->qs : which == 2 ? g->ys : g->hs;
    WMAR_HIP_CHECK(hipMemcpy(out, src, (size_t)SG_MAX_ROWS * (which == 3 ? 4 : 1) * g->D * 4, hipMemcpyDeviceToHost));
    return WMAR_OK;
}
#endif

#ifdef WMAR_DEV_KNOBS
// dev only: row `pos` of every layer's K (which = 0) or V (1) cache -> out[L][Bmax][H][hd] on the host: the per-layer fingerprint of a
// step that costs the step itself nothing (scripts/stress_kv.py)
int wmar_gpt_debug_kv_row(wmar_gpt* g, int which, int pos, float* out) {
    WMAR_REQUIRE(g && out && pos >= 0 && pos < g->Tmax, "debug_kv_row: bad argument");
    WMAR_HIP_CHECK(hipDeviceSynchronize());
    const float* src = (which ? g->vcache : g->kcache) + (long long)pos * g->hd;
    WMAR_HIP_CHECK(hipMemcpy2D(out, (size_t)g->hd * 4, src, (size_t)g->Tmax * g->hd * 4, (size_t)g->hd * 4,
                               (size_t)g->L * g->Bmax * g->H, hipMemcpyDeviceToHost));
    return WMAR_OK;
}
#endif


int wmar_gpt_set_attention_phases(wmar_gpt* g, int32_t one_wave_upto, int32_t two_waves_upto) {
    WMAR_REQUIRE(g, "set_attention_phases: null argument");
    if (one_wave_upto < 0 && two_waves_upto < 0) {          // (-1, -1): back to the automatic schedule (batch-dependent, att_phase)
        if (g->att_user) g->drop_graph();
        g->att_t1 = 1 << 30; g->att_t2 = 1 << 30; g->att_user = false;
        return WMAR_OK;
    }
    WMAR_REQUIRE(one_wave_upto >= 0 && two_waves_upto >= one_wave_upto, "set_attention_phases: bad thresholds");
    if (!g->att_user || g->att_t1 != one_wave_upto || g->att_t2 != two_waves_upto) g->drop_graph();
    g->att_t1 = one_wave_upto; g->att_t2 = two_waves_upto; g->att_user = true;
    return WMAR_OK;
}

int wmar_gpt_set_timing(wmar_gpt* g, int32_t enabled) { if (!g) return WMAR_EINVAL; g->timing = enabled; return WMAR_OK; }
int wmar_gpt_get_timing(wmar_gpt* g, double* total_us, int64_t* calls, double* step_ms) {
    if (!g) return WMAR_EINVAL;
    for (int i = 0; i < WMAR_T_NCLASS; ++i) {
        if (total_us) total_us[i] = g->cls_us[i];
        if (calls) calls[i] = g->cls_calls[i];
    }
    if (step_ms) *step_ms = g->step_ms;
    return WMAR_OK;
}

#ifndef WMAR_GRAPH_STEPS_DEFAULT
#define WMAR_GRAPH_STEPS_DEFAULT 4       // 2 / 4 / 8 / 16 measured alike (3.723 -> 3.712 ms per step: the seam between two replays is ~10 us)
#endif
static int gpt_generate_once(wmar_gpt* g, const wmar_wm_ctx* wm, const wmar_sample_params* sp, const int64_t* cond_dev,
                             int64_t B, int32_t steps, const float* q_dev, int64_t* tokens_out_dev,
                             float* logits_trace_dev, void* stream) {
    WMAR_REQUIRE(g && sp && cond_dev && q_dev && tokens_out_dev, "generate: null argument");
    WMAR_REQUIRE(B >= 1 && B <= g->Bmax, "generate: batch %lld outside 1..%d", (long long)B, g->Bmax);
    WMAR_REQUIRE(steps >= 1 && steps <= g->Tmax, "generate: steps %d outside 1..block_size %d", steps, g->Tmax);
    WMAR_REQUIRE(!(sp->top_p >= 0) || sp->top_p <= 1.0, "`top_p` has to be a float > 0 and < 1, but is %f", sp->top_p);
    if (wm) {
        WMAR_REQUIRE(wm->table_dev && wm->vocab_size == g->V, "generate: watermark vocab mismatch");
        WMAR_REQUIRE(wm->context_size >= 0 && wm->context_size <= WMAR_MAX_CONTEXT, "context size unsupported");
        WMAR_REQUIRE(wm->seed_strategy != WMAR_SEED_SPATIAL || wm->context_size == 1 || wm->context_size == 3,
                     "Spatial seeding only implemented for context size in [1,3]");
    }
    hipStream_t st = (hipStream_t)stream;
    const long long pstride = g->Tmax + 1;
    if (int rc = gpt_inject(g, st)) return rc;
    // past[b][0] = conditioning token; position and step counters to 0
    WMAR_HIP_CHECK(hipMemcpy2DAsync(g->past, pstride * 8, cond_dev, 8, 8, (size_t)B, hipMemcpyDeviceToDevice, st));
    WMAR_HIP_CHECK(hipMemsetAsync(g->pos_dev, 0, 8, st));

    SampArgs a{};
    if (wm) {
        a.wm.table = wm->table_dev; a.wm.n_rows = wm->n_rows; a.wm.row_words = (wm->vocab_size + 31) / 32;
        a.wm.seed_mode = wm->seed_strategy; a.wm.h = wm->context_size; a.wm.S = wm->spatial_dim;
        a.wm.delta = wm->delta; a.wm.enabled = 1;
    }
    a.logits = g->logits; a.V = g->V; a.past = g->past; a.past_stride = pstride;
    a.t_dev = g->step_dev;      // filled below: current length = step + 1
    a.temperature = sp->temperature; a.top_k = sp->top_k; a.use_top_p = sp->top_p >= 0;
    a.top_p_thr = (float)(1.0 - sp->top_p);
    a.q = q_dev; a.q_step_stride = (long long)B * g->V; a.step_dev = g->step_dev;
    a.scratch = a.V > 65536 ? g->scratch : nullptr;   /* rows up to 65536 entries live in the sampler's registers */
    a.tok_out = (long long*)tokens_out_dev; a.tok_out_stride = steps;
    a.past_append = g->past; a.trace = logits_trace_dev; a.B = B;
    // sampler reads its length from pos_dev+... : length of past at step n is n+1 == pos+1.
    // k_sample_fused takes t from *t_dev, so point it at a dedicated counter kept at pos+1.
    int* len_dev = g->pos_dev + 2;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, len_dev, 1);
    a.t_dev = len_dev;

    StepIO io{g->past, pstride, 1, g->logits};
    auto one_step = [&](hipStream_t s, int phase) -> int {
        g->att_nw = wmar_gpt::phase_waves(phase);
        int rc = enqueue_step(g, B, io, s);
        if (rc) return rc;
        g->span_begin(WMAR_T_SAMPLE, s);
        rc = launch_sample_fused(a, s);
        g->span_end(s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_advance3, dim3(1), dim3(1), 0, s, g->pos_dev);
        return launch_status("k_advance3");
    };

    if (g->timing) WMAR_HIP_CHECK(hipEventRecord(g->ev0, st));
    if (sp->use_graph) {
        // The captured step only depends on pointers and scalars: reuse the instantiated graph
        // when a call repeats them (bench / harness loops), re-capture otherwise.
        unsigned long long key[12] = {(unsigned long long)B, (unsigned long long)steps, (unsigned long long)(uintptr_t)q_dev,
                                      (unsigned long long)(uintptr_t)tokens_out_dev, (unsigned long long)(uintptr_t)logits_trace_dev,
                                      wm ? (unsigned long long)(uintptr_t)wm->table_dev : 0ull,
                                      wm ? (unsigned long long)wm->n_rows : 0ull,
                                      wm ? ((unsigned long long)wm->seed_strategy << 40) | ((unsigned long long)wm->context_size << 20) | (unsigned long long)wm->spatial_dim : 0ull,
                                      0ull, 0ull, (unsigned long long)sp->top_k, 0ull};
        float fd = wm ? wm->delta : 0.f, ft = sp->temperature;
        memcpy(&key[8], &fd, 4); memcpy(&key[9], &ft, 4); memcpy(&key[11], &sp->top_p, 8);
        bool need[wmar_gpt::N_PHASE] = {false, false, false};
        for (int n = 0; n < steps; ++n) need[g->att_phase(n + 1, B)] = true;
        // Steps per captured graph: a run that stays in ONE attention phase (the default schedule) replays groups of `gs` steps, so
        // the seam between two graph launches is paid once per group.
        int gs = 1;
        {
            static int env_gs = -1;
            if (env_gs < 0) { const char* e = getenv("WMAR_GRAPH_STEPS"); env_gs = e ? atoi(e) : 0; }
            const int want = env_gs > 0 ? env_gs : WMAR_GRAPH_STEPS_DEFAULT;
            const int n_need = (int)need[0] + (int)need[1] + (int)need[2];
            if (n_need == 1 && want > 1 && steps % want == 0) gs = want;
        }
        key[1] |= (unsigned long long)gs << 32;
        if (memcmp(key, g->graph_key, sizeof(key)) != 0) g->drop_graph();
        bool have = true;
        for (int ph = 0; ph < wmar_gpt::N_PHASE; ++ph) have = have && (!need[ph] || g->exec[ph]);
        if (!have) {
            g->drop_graph();
            for (int ph = 0; ph < wmar_gpt::N_PHASE; ++ph) {
                if (!need[ph]) continue;          // only the phases this run passes through are captured
                WMAR_HIP_CHECK(hipStreamBeginCapture(g->cap_stream, hipStreamCaptureModeThreadLocal));
                int rc = WMAR_OK;
                for (int k = 0; k < gs && rc == WMAR_OK; ++k) rc = one_step(g->cap_stream, ph);
                hipError_t e = hipStreamEndCapture(g->cap_stream, &g->graph[ph]);
                if (rc) { g->drop_graph(); return rc; }
                if (e != hipSuccess) { g->drop_graph(); set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return WMAR_EHIP; }
                e = hipGraphInstantiate(&g->exec[ph], g->graph[ph], nullptr, nullptr, 0);
                if (e != hipSuccess) { g->drop_graph(); set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return WMAR_EHIP; }
            }
            memcpy(g->graph_key, key, sizeof(key));
        }
        hipError_t e = hipSuccess;
        for (int n = 0; n < steps && e == hipSuccess; n += gs) e = hipGraphLaunch(g->exec[g->att_phase(n + 1, B)], st);
        if (e == hipSuccess) e = hipEventRecord(g->ev1, st);   // also marks "replays finished" for drop_graph()
        if (e == hipSuccess) { g->pending = true; }
        if (e != hipSuccess) { set_error("graph replay failed: %s", hipGetErrorString(e)); return WMAR_EHIP; }
        { StepPlan pl(g, B, io, nullptr); if (!pl.small && pl.proj_xr) g->xr_enqueued = true; if (pl.persist) g->ps_enqueued = true; }   // a replay runs the launches of its capture
        // asynchronous: the caller's stream orders everything after the replays
        if (g->timing) WMAR_HIP_CHECK(hipEventSynchronize(g->ev1));
    } else {
        g->span_on = g->timing != 0;
        int rc = WMAR_OK;
        for (int n = 0; n < steps && rc == WMAR_OK; ++n) rc = one_step(st, g->att_phase(n + 1, B));
        g->span_on = false;
        if (rc) return rc;
        if (g->timing) {
            WMAR_HIP_CHECK(hipEventRecord(g->ev1, st));
            WMAR_HIP_CHECK(hipStreamSynchronize(st));
            g->span_collect();
        }
    }
    if (g->timing) {
        float ms = 0;
        WMAR_HIP_CHECK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
        g->step_ms = ms / steps;
    }
    return WMAR_OK;
}

int wmar_gpt_generate(wmar_gpt* g, const wmar_wm_ctx* wm, const wmar_sample_params* sp, const int64_t* cond_dev,
                      int64_t B, int32_t steps, const float* q_dev, int64_t* tokens_out_dev,
                      float* logits_trace_dev, void* stream) {
    WMAR_REQUIRE(g, "generate: null argument");
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (int rc = gpt_generate_once(g, wm, sp, cond_dev, B, steps, q_dev, tokens_out_dev, logits_trace_dev, stream)) return rc;
        // While the engine runs the fused projection launch the call waits for its replays and reads the barrier flags; a run that
        // raised one is repeated on the two-launch path (same inputs, same noise: the same tokens).  On the two-launch path the call
        // stays asynchronous.
        const int f = gpt_sync_failed(g, (hipStream_t)stream);
        if (f < 0) return f;
        if (f == 0) return WMAR_OK;
    }
    set_error("generate: the in-launch barrier flag is up on the two-launch path");
    return WMAR_EHIP;
}

}  // extern "C"
