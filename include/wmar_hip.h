/*
 * wmar_hip.h -- C ABI of libwmar_hip.so, the MI355X (gfx950) implementation of
 * wmar's watermarked autoregressive image generation + detection hot path.
 *
 * Conventions
 *   - plain C types only; every `*_dev` / "device" pointer is a HIP device pointer
 *     (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  Tensors stay owned by the caller.
 *   - every function returns 0 on success or a negative WMAR_E* code;
 *     wmar_last_error() returns a human-readable message for the calling thread.
 *   - engines allocate in *_create; afterwards only small per-call scratch is allocated (prompt tables of
 *     wmar_cham_generate_image, the one-off unconditional adaLN table of wmar_rar_generate, a status word of
 *     wmar_gumbel_score).
 *   - an engine handle (wmar_gpt / wmar_rar / wmar_cham / wmar_vq / wmar_mvq) is NOT thread-safe: it owns
 *     its workspaces, KV cache and captured graphs; use it from one thread and one stream at a time
 *     (the reference is single-threaded per model too).
 *
 * Each entry point cites the reference interface (facebookresearch/wmar) it replaces.
 */
#ifndef WMAR_HIP_H
#define WMAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WMAR_OK 0
#define WMAR_EINVAL (-1)   /* bad argument / unsupported shape            */
#define WMAR_EHIP (-2)     /* HIP runtime error                           */
#define WMAR_ESHORT (-3)   /* detect: len(codes) - context_size < 1       */
/* LINEAR / FIXED seeding take any context size up to this (gentime_watermark.py:236-241 has no limit; the key table has
 * context_size * (vocab - 1) + 1 rows of vocab / 8 bytes, so memory is the practical bound); SPATIAL: 1 or 3 as in the
 * reference (:243); the Chameleon generation loop keeps 3 prompt tokens in front of the image tokens: context_size <= 3 there. */
#define WMAR_MAX_CONTEXT 16
#define WMAR_ENOMEM (-4)
#define WMAR_EMISSING (-5) /* a required checkpoint tensor is missing      */

/* enum values of wmar.watermarking.gentime_watermark.SeedStrategy / SplitStrategy (:95-106) */
#define WMAR_SEED_FIXED 0
#define WMAR_SEED_LINEAR 1
#define WMAR_SEED_SPATIAL 2
#define WMAR_SPLIT_RAND 0
#define WMAR_SPLIT_STRATIFIED 1

const char* wmar_last_error(void);
int wmar_version(void);

/* ---------------------------------------------------------------- watermark key
 * The greenlist of a context is a pure function of seed32 = (salt*sum(ctx)) mod 2^32
 * (GentimeWatermark._get_greenlist_ids_for_context, gentime_watermark.py:219-226, and
 * _split_with_seed :161-174).  The whole key is therefore a table of bitmaps:
 * row r = greenlist of context sum r, `vocab` bits packed LSB-first into uint32
 * words, row stride = wmar_key_row_words(vocab). */
typedef struct wmar_key_params {
    uint64_t salt_key;        /* GentimeWatermark.salt_key (default 15485863)            */
    const int64_t* alive_ids; /* host, in the order init_alivecodes produced them        */
    int64_t n_alive;
    const int64_t* dead_ids;  /* host                                                    */
    int64_t n_dead;
    int64_t vocab_size;
    double gamma;
    int32_t split_strategy;   /* WMAR_SPLIT_*  (clustering is out of scope)              */
    int32_t seed_strategy;    /* WMAR_SEED_*                                             */
} wmar_key_params;

int64_t wmar_key_row_words(int64_t vocab_size);
/* Number of table rows needed so that every reachable context sum has a row:
 * FIXED -> 1; LINEAR / SPATIAL with context size h and token ids < max_token -> h*(max_token-1)+1. */
int64_t wmar_key_table_rows(int32_t seed_strategy, int32_t context_size, int64_t max_token);
/* Host builder (MT19937 + Fisher-Yates, `n_threads` host threads; 0 = all cores).
 * Writes rows [row0, row0+n_rows) to `out_host`.  Replaces the per-row, per-step
 * _split_with_seed calls of gentime_watermark.py:266 and :281. */
int wmar_key_table_build(const wmar_key_params* key, int64_t row0, int64_t n_rows, uint32_t* out_host,
                         int32_t n_threads);
/* One greenlist in the reference's order (ids as _split_with_seed returns them). Host. */
int64_t wmar_key_greenlist(const wmar_key_params* key, uint64_t seed, int64_t* out_ids_host);

/* What every device-side watermark call needs to find a row's bitmap. */
typedef struct wmar_wm_ctx {
    const uint32_t* table_dev; /* [n_rows, row_words]                                     */
    int64_t n_rows;
    int64_t vocab_size;
    int32_t seed_strategy;
    int32_t context_size;      /* h                                                       */
    int32_t spatial_dim;       /* 16 Taming/RAR, 32 Chameleon (gentime_watermark.py:153)  */
    float delta;
} wmar_wm_ctx;

/* GentimeWatermark._process_logits (gentime_watermark.py:229-271): logits[b, G(ctx_b)] += delta,
 * in place.  past_ids_dev: int64 [B, t] with row stride `past_stride`.  Rows whose context is
 * too short are left untouched (the reference swallows the ValueError). */
int wmar_wm_process_logits(const wmar_wm_ctx* wm, float* logits_dev, int64_t B, const int64_t* past_ids_dev,
                           int64_t t, int64_t past_stride, void* stream);

/* The sampling stage of sample_with_past (mingpt.py:348-363) fused into one kernel per row:
 * watermark bias -> /temperature -> top-k -> top-p -> softmax -> argmax(p / q).
 * wm == NULL: no watermark.  top_k <= 0: off.  top_p < 0: off.  q_dev: the Exp(1) noise
 * torch.multinomial would draw, [B, V].  scratch_dev: float [B, V].  tok_out_dev: int64 [B].
 * The arithmetic is the one pinned in include/wmar_math.h. */
int wmar_sample_fused(const wmar_wm_ctx* wm, const float* logits_dev, int64_t B, int64_t V,
                      const int64_t* past_ids_dev, int64_t t, int64_t past_stride, float temperature,
                      int32_t top_k, double top_p, const float* q_dev, float* scratch_dev,
                      int64_t* tok_out_dev, void* stream);

/* GentimeWatermark.detect (gentime_watermark.py:285-344): per row of codes_dev int64 [B, L]:
 * unique (h+1)-grams -> n_scored, n_green -> p = I_gamma(n_green, 1+n_scored-n_green) in fp64
 * (NaN when n_green == 0, as scipy.special.betainc).  mask_dev (nullable): int8 [B, mask_stride],
 * entry i: -1 for the first h positions and for repeated n-grams, else the green bit.
 * Returns WMAR_ESHORT when L - h < 1 (the reference raises ValueError). */
int wmar_detect(const wmar_wm_ctx* wm, double gamma, const int64_t* codes_dev, int64_t B, int64_t L,
                int32_t* n_scored_dev, int32_t* n_green_dev, double* pval_dev, int8_t* mask_dev,
                int64_t mask_stride, void* stream);
/* number of n-grams the detector enumerates for a passage of length L (mask length = h + this) */
int64_t wmar_detect_num_ngrams(int32_t seed_strategy, int32_t context_size, int64_t L);

/* ------------------------------------------------------------------------ GPT
 * minGPT decoder with a static KV cache (deps/taming/modules/transformer/mingpt.py:125-214).
 * Tensors are looked up by their checkpoint key names relative to `transformer.`
 * (tok_emb.weight, pos_emb, blocks.N.{ln1,ln2}.{weight,bias}, blocks.N.attn.{key,query,value,proj}.{weight,bias},
 *  blocks.N.mlp.{0,2}.{weight,bias}, ln_f.{weight,bias}, head.weight), all fp32 device pointers in
 * torch's nn.Linear layout; they are repacked into MFMA-fragment order at create time and
 * need not outlive the call. */
typedef struct wmar_gpt_config {
    int32_t vocab_size, block_size, n_layer, n_head, n_embd;
    int32_t max_batch; /* KV cache and workspaces are sized for this */
} wmar_gpt_config;

typedef struct wmar_gpt wmar_gpt;

int wmar_gpt_create(const wmar_gpt_config* cfg, const char* const* names, const void* const* tensors_dev,
                    int32_t n_tensors, void* stream, wmar_gpt** out);
void wmar_gpt_destroy(wmar_gpt* g);
int64_t wmar_gpt_device_bytes(const wmar_gpt* g);

/* GPT.forward_with_past for one new token per sequence (mingpt.py:183-214): consumes
 * tok_dev int64 [B] at position `pos` (= past_length), appends K/V at `pos`, writes
 * logits_dev float [B, vocab].  Positions must be fed in order 0,1,2,... */
int wmar_gpt_decode_step(wmar_gpt* g, const int64_t* tok_dev, int64_t B, int32_t pos, float* logits_dev,
                         void* stream);

/* sample_with_past (mingpt.py:326-368) as `steps` replays of ONE captured hipGraph
 * (decode step + fused watermark/sampling + position advance; the position lives in
 * device memory).  cond_dev int64 [B] (the class token), q_dev float [steps, B, V] (noise for
 * every step, see wmar_sample_fused), tokens_out_dev int64 [B, steps].
 * logits_trace_dev (nullable): float [steps, B, V], raw model logits per step.
 * Everything is ordered on `stream`; one generate per engine at a time.  Asynchronous on the two-launch path; while the fused
 * projection launch is in use the call waits for its replays and verifies the in-launch barrier (see wmar_gpt_check). */
typedef struct wmar_sample_params {
    float temperature;
    int32_t top_k;  /* <= 0: off */
    double top_p;   /* < 0: off  */
    int32_t use_graph; /* 1: hipGraph replay, 0: eager launches */
} wmar_sample_params;

int wmar_gpt_generate(wmar_gpt* g, const wmar_wm_ctx* wm, const wmar_sample_params* sp, const int64_t* cond_dev,
                      int64_t B, int32_t steps, const float* q_dev, int64_t* tokens_out_dev,
                      float* logits_trace_dev, void* stream);

/* Per-kernel-class device times of the last wmar_gpt_generate call, measured with HIP events
 * recorded on the caller's stream around every launch.  Only available for eager runs
 * (use_graph = 0) after wmar_gpt_set_timing(g, 1); used by bench.py for the roofline line.
 * Classes: */
#define WMAR_T_EMBED 0   /* token+position embedding, LN statistics            */
#define WMAR_T_QKV 1     /* LN1 -> QKV GEMM -> KV cache                         */
#define WMAR_T_ATTN 2    /* decode attention                                   */
#define WMAR_T_PROJ 3    /* attention output projection (split-K slabs)        */
#define WMAR_T_RESID 4   /* residual fold + LN statistics                      */
#define WMAR_T_FC1 5     /* LN2 -> FC1 GEMM -> GELU                             */
#define WMAR_T_FC2 6     /* FC2 GEMM (split-K slabs)                           */
#define WMAR_T_HEAD 7    /* ln_f -> vocabulary head GEMM                       */
#define WMAR_T_SAMPLE 8  /* fused watermark + sampling                         */
#define WMAR_T_NCLASS 9
/* In-launch synchronisation (no reference counterpart).  At batches of 33..64 rows the output projection, its split-K reduction,
 * the residual fold and the LayerNorm statistics run as ONE launch (k_bx_xr) behind an XCD-local barrier.  wmar_gpt_create enables it
 * only when blocks with equal blockIdx % 8 share an XCD and the device holds the whole grid at once (WMAR_NO_XR=1 at creation keeps
 * the two-launch path: k_bx + k_resid_stats).  Every wait is bounded; a wait that gives up, or a block on a foreign XCD, raises a
 * device flag, later waits leave at once, and wmar_gpt_generate / wmar_gpt_decode_step -- which wait for the stream and read the
 * flags while the fused launch is in use -- switch the engine to the two-launch path and RE-RUN the call (same inputs: same
 * results; wmar_gpt_plan_info reports `barrier_fallbacks`).  On the two-launch path both calls are asynchronous again.
 * wmar_gpt_check does the same flag test for callers that enqueue single roles (wmar_gpt_profile_role): WMAR_EHIP = the launches
 * since the last check are invalid, the engine continues on the two-launch path.  WMAR_INJECT_SYNC_FAIL=1 at creation (tests)
 * raises the flag in front of the first fused call. */
int wmar_gpt_check(wmar_gpt* g, void* stream);
int wmar_gpt_set_timing(wmar_gpt* g, int32_t enabled);
/* Decode attention runs 1 / 2 / 4 waves per (sequence, head) while the cache holds <= one_wave_upto / <= two_waves_upto /
 * more rows (one captured step graph per phase).  Defaults were measured at batch 64 on MI355X; a tuning entry point only --
 * results do not depend on it beyond fp32 summation order across waves.  (-1, -1) returns to the automatic schedule (one wave at
 * every length for batch x heads >= 512, 2 / 4 waves up to / beyond 128 cached rows below that). */
int wmar_gpt_set_attention_phases(wmar_gpt* g, int32_t one_wave_upto, int32_t two_waves_upto);
/* Replays ONE role's kernel `iters` times back to back on `stream` (cycling through the layers, so
 * weights stream from HBM as in a real step) between two HIP events: *avg_us = average per launch,
 * launch boundary included.  kv_len = cached rows the attention role reads. */
int wmar_gpt_profile_role(wmar_gpt* g, int32_t role, int64_t B, int32_t kv_len, int32_t iters, void* stream,
                          double* avg_us);
/* The kernels a decode step of B rows launches per role, as `role=kernel;...` text (bench.py names what it measured from this, not
 * from a table of its own).  Returns WMAR_EINVAL if buf is too small. */
int wmar_gpt_plan_info(wmar_gpt* g, int64_t B, char* buf, int64_t buf_len);
/* total_us[c], calls[c] for each class; *step_ms = average wall time of one decode step (any mode). */
int wmar_gpt_get_timing(wmar_gpt* g, double* total_us, int64_t* calls, double* step_ms);

/* ------------------------------------------------------------------------ RAR
 * RAR generator (deps/rar/modeling/rar.py): adaLN blocks, per-head q/k LayerNorm, KV cache,
 * classifier-free guidance.  Tensors by the checkpoint's key names (cls_token, embeddings.weight,
 * pos_embed, target_aware_pos_embed, timesteps_embeddings, blocks.N.{norm1,norm2}.{weight,bias},
 * blocks.N.attn.{qkv,proj}.{weight,bias}, blocks.N.attn.{q_norm,k_norm}.{weight,bias},
 * blocks.N.mlp.{fc1,fc2}.{weight,bias}, blocks.N.adaLN_modulation.1.{weight,bias},
 * adaln_before_head.adaLN_modulation.1.{weight,bias}, lm_head.{weight,bias}). */
typedef struct wmar_rar_config {
    int32_t hidden_size, num_hidden_layers, num_attention_heads, intermediate_size;
    int32_t image_seq_len, codebook_size, condition_num_classes;
    int32_t max_batch;   /* images per call; rows double under guidance */
} wmar_rar_config;

typedef struct wmar_rar wmar_rar;

int wmar_rar_create(const wmar_rar_config* cfg, const char* const* names, const void* const* tensors_dev,
                    int32_t n_tensors, void* stream, wmar_rar** out);
void wmar_rar_destroy(wmar_rar* g);
int64_t wmar_rar_device_bytes(const wmar_rar* g);

/* RAR.forward_fn for ONE sequence position with a warm KV cache (rar.py:319-405): row m consumes
 * tok_dev[m] (-1 = the cls token) at position `pos` under condition id cond_ids_dev[m] (already offset:
 * class + codebook_size + 1, or the "none" id) and yields logits_dev float [M, codebook].  Positions in
 * order 0,1,2,...  Used by the parity tests. */
int wmar_rar_forward_position(wmar_rar* g, const int64_t* tok_dev, const int64_t* cond_ids_dev, int64_t M, int32_t pos,
                              float* logits_dev, void* stream);

/* RAR.generate (rar.py:408-459): class_ids_dev int64 [B] in [0, classes); cfg_scale_host float
 * [image_seq_len] = the reference's per-step cfg_scale (host; ignored when use_guidance == 0);
 * q_dev float [image_seq_len, B, V] Exp(1) noise; tokens_out_dev int64 [B, image_seq_len].
 * The watermark context is the generated ids only, so the first token is never biased. */
int wmar_rar_generate(wmar_rar* g, const wmar_wm_ctx* wm, const int64_t* class_ids_dev, int64_t B,
                      const float* cfg_scale_host, int32_t use_guidance, float temperature, const float* q_dev,
                      int64_t* tokens_out_dev, int32_t use_graph, void* stream);

/* RAR generation with the Gumbel-key sampler below instead of multinomial + greenlist (BASELINE
 * config "RAR-XL ... Gumbel-key watermark"; an EXTENSION -- the reference image code has no such
 * path).  log_rs_dev float [V]: the fixed key (wmar_gumbel_key_build). */
int wmar_rar_generate_gumbel(wmar_rar* g, const int64_t* class_ids_dev, int64_t B, const float* cfg_scale_host,
                             int32_t use_guidance, float temperature, float top_p, int32_t top_k,
                             const float* log_rs_dev, int64_t* tokens_out_dev, int32_t use_graph, void* stream);

/* In-launch synchronisation of the RAR engine: the fused residual + adaLN modulation launch of a block (k_resid_mod) lets its
 * workgroups wait for each other's partial sums.  wmar_rar_create enables it when the device holds the whole grid at once
 * (occupancy x compute units; WMAR_NO_XR=1 disables it), otherwise the same work runs as a two-launch pair (bit-identical results).
 * A wait that gives up raises a device flag; wmar_rar_generate* / wmar_rar_forward_position wait for the stream while the fused
 * launch is in use, and on a raised flag switch to the two-launch pair and RE-RUN the call.  wmar_rar_check: the same flag test
 * for other callers (WMAR_EHIP = the launches since the last check are invalid).  wmar_rar_launch_status: whether the fused
 * launch is still in use, and how many calls were re-run. */
int wmar_rar_launch_status(const wmar_rar* g, int32_t* fused, int32_t* fallbacks);
int wmar_rar_check(wmar_rar* g, void* stream);

/* ----------------------------------------------------------------- Gumbel key (row G1)
 * Aaronson-style sampling of wmar_audio/watermark/engine.py:29-75 (`gumbel_sample`) and its
 * detector :123-134 (`gumbel_score_tok`).  The key of a row is rs = torch.rand(V, generator =
 * CPU MT19937 seeded with the row's window hash) (engine.py:64-66); with ngram = 0 the hash is
 * the seed itself (:17-18), so one key serves every row.
 *
 * wmar_gumbel_key_build (host): rs_host[v] = (mt.next() & 0xffffff) * 2^-24 (torch's fp32
 * uniform), log_rs_host[v] = (float)log(rs) and score_host[v] = (float)-log(1 - rs).  Any output
 * may be null. */
int wmar_gumbel_key_build(uint64_t seed, int64_t vocab_size, float* rs_host, float* log_rs_host, float* score_host);

/* next_token[b] = argmax_v rs[v]^(1/p[v]) with p = softmax(logits/temp) after the optional top-p
 * (descending sort, drop where cumsum - p > top_p, renormalise) or else top-k (others := 1e-6,
 * renormalise) -- engine.py:41-75.  The race is run as argmax log(rs[v]) * (1/p[v]) (monotone
 * image; scores below log(2^-150) collapse to one class like the reference's fp32 underflow).
 * use_sampling == 0 or temp <= 0: plain argmax(logits).  key_row_stride: elements between the
 * keys of consecutive rows (0: one shared key).  V <= 16384. */
int wmar_gumbel_sample(const float* logits_dev, int64_t B, int64_t V, const float* log_rs_dev, int64_t key_row_stride,
                       int32_t use_sampling, float temp, float top_p, int32_t top_k, int64_t* tok_out_dev, void* stream);

/* gumbel_score_tok for tokens int64 [B, L]: scores_f32_dev[b,l] = -log(1 - rs)[token] and
 * scores_i64_dev[b,l] = that value truncated toward zero (what the reference returns, because it
 * accumulates into zeros_like(tokens)).  Either output may be null. */
int wmar_gumbel_score(const int64_t* tokens_dev, int64_t B, int64_t L, int64_t V, const float* score_key_dev,
                      int64_t key_row_stride, int64_t* scores_i64_dev, float* scores_f32_dev, void* stream);

/* ------------------------------------------------------------------ Chameleon (row C1)
 * Chameleon / Anole text->image decode: deps/chameleon/inference/transformer.py:288-353
 * (Transformer), chameleon.py:299-389 (ImageDecoder), logits_processor.py:135-156, 312-336,
 * token_selector.py:26-47.  bf16 weights and activations, fp32 accumulation.  Tensors by the
 * checkpoint's key names (tok_embeddings.weight, layers.N.attention.{wqkv,wo}.weight,
 * layers.N.attention.{q,k}_normalization.{weight,bias}, layers.N.feed_forward.{w13,w2}.weight,
 * layers.N.{attention,ffn}_norm.weight, norm.weight, output.weight); all bf16 (tensors_bf16 = 1,
 * the reference's storage) or all fp32 (rounded to bf16 at pack time). */
typedef struct wmar_cham_config {
    int32_t dim, n_layers, n_heads, n_kv_heads, vocab_size;
    int32_t ffn_hidden;        /* FeedForward hidden size after the multiple_of rounding (transformer.py:175-179) */
    float norm_eps, rope_theta;
    int32_t qk_normalization;  /* LayerNorm(head_dim) on q and k (transformer.py:74-77) */
    int32_t swin_norm;         /* must be 0 (the 30B block order is not built) */
    int32_t max_rows;          /* sequences per step = 3 * images per call */
    int32_t max_seq_len;       /* prompt + generated tokens per sequence */
    int32_t tensors_bf16;
} wmar_cham_config;

typedef struct wmar_cham_sample_params {
    float temperature;
    double top_p;                                     /* < 0: off */
    float guidance_scale_text, guidance_scale_image;  /* Options.Image.cfg: 3.0 / 1.2 */
    int32_t use_graph;
    int32_t pad_id;   /* vocab.pad_id: what AlignPromptRight pads short prompts with on the left (alignment.py:27-41); the
                       * watermark context of the first image tokens can reach into it when context_size >= 2 */
} wmar_cham_sample_params;

typedef struct wmar_cham wmar_cham;

int wmar_cham_create(const wmar_cham_config* cfg, const char* const* names, const void* const* tensors_dev,
                     int32_t n_tensors, void* stream, wmar_cham** out);
void wmar_cham_destroy(wmar_cham* g);
int64_t wmar_cham_device_bytes(const wmar_cham* g);

/* One token per sequence: row m consumes tok_dev[m] at position pos_dev[m] (its KV cache must hold
 * positions 0..pos-1) -> logits_dev float [M, vocab] (nullable: cache update only).  This is
 * Transformer.forward_with_attn_bias with q_seqlen = 1 per sequence (model_adapter.py:106-113);
 * a prompt is prefilled by calling it once per position. */
int wmar_cham_forward_tokens(wmar_cham* g, const int64_t* tok_dev, const int32_t* pos_dev, int64_t M, float* logits_dev,
                             void* stream);

/* ImageDecoder (chameleon.py:299-389) for B prompts: prompt_tokens_host = the 3B token lists of
 * _split_inputs_for_cfg (:351-372; full-conditioned, image-conditioned, unconditioned) laid end to
 * end, prompt_lens_host int32 [3B].  Per generated token: guidance mix -> watermark bias ->
 * allow-only (allow_dev: bit per vocabulary entry, nullable; allow_ids_dev: the same set as ascending int32 ids
 * [n_allow], nullable -- with it the sampler works on the compacted row, which is exact) -> temperature -> top-p -> softmax ->
 * multinomial on the first stream (q_dev float [n_tokens, B, vocab] Exp(1) noise, one [B, vocab]
 * draw per token as probs.multinomial makes).  tokens_out_dev int64 [B, n_tokens] (vocabulary ids).
 * The watermark context is the whole right-aligned input row, as the reference's processors see it (generation.py:86): its
 * last 3 entries (prompt tokens, left-padded with pad_id) are kept in front of the generated tokens, enough for every linear
 * context size the library supports (<= 3).  SPATIAL seeding is rejected: the reference does not offer it for Chameleon. */
int wmar_cham_generate_image(wmar_cham* g, const wmar_wm_ctx* wm, const int64_t* prompt_tokens_host,
                             const int32_t* prompt_lens_host, int64_t B, const wmar_cham_sample_params* sp,
                             const uint32_t* allow_dev, const int32_t* allow_ids_dev, int32_t n_allow, const float* q_dev,
                             int32_t n_tokens, int64_t* tokens_out_dev, void* stream);

/* The ImageDecoder logits pipeline of one step as ONE launch (chameleon.py:313-327, generation.py:84-93):
 * logits3_dev float [3B, V] = [full | image-conditioned | unconditioned] rows; guidance mix
 * (logits_processor.py:312-336) -> watermark bias (called positionally on the input rows,
 * past_ids_dev int64 [B, past_stride] with t valid entries; nullable for FIXED keys) ->
 * allow-only bitmap (nullable; allow_ids_dev / n_allow: the same set as ascending ids, enables row compaction) ->
 * /temperature -> top-p (< 0: off) -> softmax ->
 * argmax(p / q) on the first stream (token_selector.py:36-47).  tok_out_dev int64 [B]. */
int wmar_cham_sample(const wmar_wm_ctx* wm, const float* logits3_dev, int64_t B, int64_t V, const int64_t* past_ids_dev,
                     int64_t t, int64_t past_stride, float temperature, double top_p, float guidance_scale_text,
                     float guidance_scale_image, const uint32_t* allow_dev, const int32_t* allow_ids_dev, int32_t n_allow,
                     const float* q_dev, float* scratch_dev, int64_t* tok_out_dev, void* stream);

/* ---------------------------------------------------------------------- VQGAN
 * Taming VQGAN (deps/taming/models/vqgan.py:30-73, modules/diffusionmodules/model.py:343-538,
 * modules/vqvae/quantize.py:272-331).  Tensors by key name relative to `first_stage_model.`
 * (encoder.*, decoder.*, quantize.embedding.weight, quant_conv.*, post_quant_conv.*). */
typedef struct wmar_vq_config {
    int32_t ch, num_res_blocks, resolution, in_channels, out_ch, z_channels, embed_dim, n_embed;
    int32_t n_levels;
    int32_t ch_mult[8];
    int32_t n_attn_res;
    int32_t attn_resolutions[8];
    int32_t max_batch;
} wmar_vq_config;

typedef struct wmar_vq wmar_vq;

int wmar_vq_create(const wmar_vq_config* cfg, const char* const* names, const void* const* tensors_dev,
                   int32_t n_tensors, void* stream, wmar_vq** out);
void wmar_vq_destroy(wmar_vq* v);
int64_t wmar_vq_device_bytes(const wmar_vq* v);

/* TamingARMMWrapper.codes_to_images (wmar/models/taming_wrapper.py:79-84): codes int64 [B, S*S]
 * -> images float [B, 3, R, R] (NCHW) clamped to [-1, 1]. */
int wmar_vq_decode(wmar_vq* v, const int64_t* codes_dev, int64_t B, float* images_dev, void* stream);
/* TamingARMMWrapper.images_to_codes (taming_wrapper.py:88-92): images float [B, 3, R, R] -> codes int64 [B, S*S].
 * prequant_dev (nullable): float [B*S*S, embed_dim], the vectors handed to the quantizer. */
int wmar_vq_encode(wmar_vq* v, const float* images_dev, int64_t B, int64_t* codes_dev, float* prequant_dev,
                   void* stream);

/* --------------------------------------------------------------- MaskGIT-VQGAN (RAR's tokenizer)
 * deps/rar/modeling/modules/maskgit_vqgan.py + PretrainedTokenizer (deps/rar/modeling/titok.py:41-89).
 * Tensors by key name (encoder.*, decoder.*, quantize.embedding.weight).  Images cross this API in the
 * WRAPPER's convention, [-1, 1] (RarARMMWrapper.codes_to_images / images_to_codes, rar_wrapper.py:108-128). */
typedef struct wmar_mvq_config {
    int32_t hidden_channels, num_res_blocks, resolution, num_channels, z_channels, num_embeddings;
    int32_t n_levels;
    int32_t channel_mult[8];
    int32_t max_batch;
} wmar_mvq_config;

typedef struct wmar_mvq wmar_mvq;

int wmar_mvq_create(const wmar_mvq_config* cfg, const char* const* names, const void* const* tensors_dev,
                    int32_t n_tensors, void* stream, wmar_mvq** out);
void wmar_mvq_destroy(wmar_mvq* v);
int64_t wmar_mvq_device_bytes(const wmar_mvq* v);
int wmar_mvq_decode(wmar_mvq* v, const int64_t* codes_dev, int64_t B, float* images_dev, void* stream);
int wmar_mvq_encode(wmar_mvq* v, const float* images_dev, int64_t B, int64_t* codes_dev, float* prequant_dev,
                    void* stream);

/* ------------------------------------------------------------------------ evaluation transforms (harness)
 * wmar/augmentations/valuemetric.py:41-140, geometric.py:22-117 as applied by generate.py:142-164: one launch over the whole batch
 * [B, C, H, W] (fp32, contiguous).  op: 0 identity, 1 Gaussian blur (p0 = odd kernel size), 2 Gaussian noise (p0 = standard deviation,
 * noise_dev = standard normal draws of the image shape), 3 brightness (p0 = factor), 4 rotation (p0 = counter-clockwise quarter turns,
 * p1 = remainder in degrees, [0, 90)), 5 horizontal flip, 6 upper-left crop of p0 x p1 pixels resized back (antialiased bilinear),
 * 7 the same crop padded back with zeros.  pm1 != 0: pixels cross this call in [-1, 1] (the decoder's range) and are transformed in
 * [0, 1] and clamped, as generate.py:146-150 does around every transform.  JPEG stays on the host (PIL), as in the reference. */
int wmar_augment(int32_t op, const float* in_dev, float* out_dev, const float* noise_dev, int64_t B, int32_t C, int32_t H,
                 int32_t W, int32_t pm1, double p0, double p1, void* stream);

/* ------------------------------------------------------------------------ exchange step (RCCL over xGMI)
 * The sharded job (one process per GPU; rank r == the reference's `--chunk_id r --num_chunks world`, generate.py:204, :304) has no
 * data-path collective.  Its two exchanges -- the finished key table from rank 0 once, the per-image records once per step
 *  -- as thin RCCL wrappers; RCCL is resolved at run time so that a process that already carries one
 * (PyTorch-ROCm) keeps exactly that copy.  `id` is RCCL's 128-byte unique id: made on one rank, handed to the others by the host
 * (file, store, environment).  Buffers are device pointers, sizes in bytes; the calls are asynchronous on `stream`. */
typedef struct wmar_comm wmar_comm;
#define WMAR_COMM_ID_BYTES 128
int wmar_comm_unique_id(void* id_out, int64_t id_bytes);
int wmar_comm_init(const void* id, int64_t id_bytes, int32_t rank, int32_t world, wmar_comm** out);
int wmar_comm_bcast(wmar_comm* c, void* buf_dev, int64_t bytes, int32_t root, void* stream);
/* recv_dev holds world x bytes_per_rank, rank order */
int wmar_comm_allgather(wmar_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream);
int32_t wmar_comm_rank(const wmar_comm* c);
int32_t wmar_comm_world(const wmar_comm* c);
void wmar_comm_destroy(wmar_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* WMAR_HIP_H */
