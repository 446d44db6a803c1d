// Argument blocks shared by watermark.hip (kernels) and gpt.hip (generation graph).
#pragma once
#include "common.h"

namespace wmar {

struct WmDev {
    const uint32_t* table;
    long long n_rows;
    long long row_words;
    int seed_mode, h, S;
    float delta;
    int enabled;
};

inline WmDev make_wm(const wmar_wm_ctx* wm) {
    WmDev d{};
    if (wm) {
        d.table = wm->table_dev;
        d.n_rows = wm->n_rows;
        d.row_words = (wm->vocab_size + 31) / 32;
        d.seed_mode = wm->seed_strategy;
        d.h = wm->context_size;
        d.S = wm->spatial_dim;
        d.delta = wm->delta;
        d.enabled = 1;
    }
    return d;
}

struct SampArgs {
    WmDev wm;
    const float* logits;
    long long V;
    const long long* past;       // [B, past_stride]
    long long past_stride;
    long long t_host;            // used when t_dev == nullptr
    const int* t_dev;            // device-resident current length (generation graph)
    float temperature;
    int top_k;
    int use_top_p;
    float top_p_thr;             // (float)(1 - top_p)
    const float* q;              // [B, V]
    long long q_step_stride;     // elements between steps when q is indexed by *step_dev
    const int* step_dev;         // nullable
    float* scratch;              // [B, V]
    long long* tok_out;          // [B]
    long long tok_out_stride;    // tok_out[b*stride + step]
    long long* past_append;      // nullable: past[b*past_stride + t] = token
    float* trace;                // nullable: raw logits copy [steps][B][V]
    long long B;
    // classifier-free guidance (RAR.generate, rar.py:437-442): logits = uncond + (cond - uncond) * scale[step]
    const float* logits_uncond;  // nullable [B, V]
    const float* cfg_scale;      // device float [steps]
    // InBatchInstructCFG (deps/chameleon/inference/logits_processor.py:312-336): with logits_img set,
    // logits = uncond + g_image * (img - uncond) + g_text * (full - img); `logits` holds the fully conditioned rows
    const float* logits_img;     // nullable [B, V]
    float g_text, g_image;
    // AllowOnlyTokensLogitsProcessor (logits_processor.py:135-156): bit v clear -> logit v = -inf (after the watermark bias)
    const uint32_t* allow;       // nullable [V/32]
    // Row compaction: with `gather` (ascending source ids, int32 [V]) the row worked on is logits[gather[0..V)] -- exact
    // when every other entry would be -inf anyway (allow-only list); V is then the compact length, Vsrc the logits' width.
    // The token written is gather[argmax].
    const int* gather;           // nullable
    long long Vsrc;
};

int launch_sample_fused(const SampArgs& a, hipStream_t st);

// Gumbel-key sampler (gumbel.hip): same step / append plumbing as SampArgs.
struct GumbelArgs {
    const float* logits;
    const float* logits_uncond;  // nullable (guidance, as above)
    const float* cfg_scale;
    const int* step_dev;         // nullable
    const int* t_dev;            // nullable: current length for past_append
    long long V, B;
    const float* log_rs;         // key rows: log_rs + b * key_row_stride
    long long key_row_stride;
    int use_sampling;
    float temp, top_p;
    int top_k;
    long long* tok_out;          // tok_out[b*tok_out_stride + step]
    long long tok_out_stride;
    long long* past_append;      // nullable: past_append[b*past_stride + t] = token
    long long past_stride;
};

int launch_gumbel_sample(const GumbelArgs& a, hipStream_t st);

}  // namespace wmar
