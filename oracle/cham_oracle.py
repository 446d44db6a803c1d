"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch) of the Chameleon transformer decode path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (wmar_amd) never does.

Reference followed: deps/chameleon/inference/transformer.py:37-160 (Attention), :163-217
(FeedForward), :220-285 (TransformerBlock, swin_norm = False), :288-337 (Transformer); third-party
semantics restated from their published behaviour: xformers ``RMSNorm`` (x * rsqrt(mean(x^2) + eps)
* weight, computed in fp32), ``rope_padded`` (adjacent pairs, angle = position * theta^(-2i/head_dim),
K written rotated into the cache) and ``memory_efficient_attention_forward`` with a causal
block-diagonal mask (softmax(q k^T / sqrt(head_dim)) v per sequence, fp32 accumulation).

**Parity status: unpinned against the reference itself** -- xformers is not installed in the build
container and no checkpoint exists, so the reference transformer cannot be run here (SURVEY.md
section 8c).  Its STRUCTURE (norm placement, qk LayerNorm, rotary pairing convention, SwiGLU,
residuals, head) is pinned against an independent public implementation of the same model: the
Hugging Face ``transformers`` port (``modeling_chameleon.ChameleonDecoderLayer``), run on the same
random weights after the port's own weight-layout conversion (rotary-halves permutation of wq / wk
and of the qk-norm parameters); see tests/test_oracle_chameleon_hf.py.  The logits processors, token selector and
vocabulary translation around it ARE importable and pinned by fixtures (tests/golden).

Numerics: the reference runs in bf16 (weights, activations, KV cache) with fp32 accumulation; every
module output is rounded to bf16.  ``fold=True`` evaluates the algebraically identical form the HIP
engine uses (RMSNorm weight folded into the next Linear, 1/rms applied to its output) so that the
engine can be checked tightly; ``fold=False`` keeps the reference's rounding points.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

T = torch.Tensor


ROUND_BF16 = True   # False: plain fp32 everywhere (used to cross-check the STRUCTURE against the HF port of the model)


def bf(x: T) -> T:
    return x.to(torch.bfloat16).to(torch.float32) if ROUND_BF16 else x


def _linear(x: T, w: T) -> T:
    """bf16 Linear: exact products, fp32 accumulation, bf16 result (returned as fp32 values)."""
    return bf(x @ w.float().t())


def _rstd(x: T, eps: float) -> T:
    return torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)


def _norm_linear(x: T, gamma: T, w: T, eps: float, fold: bool) -> T:
    if fold:
        wf = bf(w.float() * gamma.float()[None, :])
        return bf(_rstd(x, eps) * (x @ wf.t()))
    n = bf(x * _rstd(x, eps) * gamma.float())                 # RMSNorm output in bf16
    return _linear(n, w)


def _rope(x: T, pos: T, theta: float) -> T:
    """x [M, H, hd] (fp32 values), pos int [M]: adjacent pairs (2i, 2i+1), fp32 math, bf16 result."""
    hd = x.shape[-1]
    i = torch.arange(hd // 2, dtype=torch.float32)
    freq = torch.pow(torch.tensor(theta, dtype=torch.float32), -2.0 * i / hd)
    ang = pos.to(torch.float32)[:, None] * freq[None, :]          # [M, hd/2]
    cs, sn = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
    x0, x1 = x[..., 0::2], x[..., 1::2]
    out = torch.empty_like(x)
    out[..., 0::2] = x0 * cs - x1 * sn
    out[..., 1::2] = x0 * sn + x1 * cs
    return bf(out)


class Cache:
    """Per layer, per row: K and V of positions 0..len-1 ([t, Hkv, hd] fp32 tensors holding bf16 values)."""

    def __init__(self, n_layers: int, rows: int):
        self.k: List[List[Optional[T]]] = [[None] * rows for _ in range(n_layers)]
        self.v: List[List[Optional[T]]] = [[None] * rows for _ in range(n_layers)]


def forward_tokens(sd: Dict[str, T], cfg, tok: T, pos: T, cache: Cache, fold: bool = False, want_logits: bool = True) -> Optional[T]:
    """One token per row (transformer.py:316-337 with q_seqlen = 1): tok int64 [M], pos int [M]; row m's cache must hold
    positions 0..pos[m]-1 (entries at >= pos[m] are overwritten, as rope_padded does).  Returns fp32 logits [M, V]."""
    M = tok.shape[0]
    H, Hkv, hd, D = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.dim
    x = sd["tok_embeddings.weight"][tok].float()
    for l in range(cfg.n_layers):
        p = f"layers.{l}."
        qkv = _norm_linear(x, sd[p + "attention_norm.weight"], sd[p + "attention.wqkv.weight"], cfg.norm_eps, fold)
        q = qkv[:, : H * hd].view(M, H, hd)
        k, v = qkv[:, H * hd:].chunk(2, 1)
        k, v = k.reshape(M, Hkv, hd), v.reshape(M, Hkv, hd)
        if cfg.qk_normalization:
            q = bf(F.layer_norm(q, (hd,), sd[p + "attention.q_normalization.weight"].float(),
                                sd[p + "attention.q_normalization.bias"].float(), 1e-5))
            k = bf(F.layer_norm(k, (hd,), sd[p + "attention.k_normalization.weight"].float(),
                                sd[p + "attention.k_normalization.bias"].float(), 1e-5))
        q, k = _rope(q, pos, cfg.rope_theta), _rope(k, pos, cfg.rope_theta)
        out = torch.empty(M, H, hd)
        for m in range(M):
            P = int(pos[m])
            kc, vc = cache.k[l][m], cache.v[l][m]
            kc = k[m][None] if kc is None or P == 0 else torch.cat([kc[:P], k[m][None]])
            vc = v[m][None] if vc is None or P == 0 else torch.cat([vc[:P], v[m][None]])
            cache.k[l][m], cache.v[l][m] = kc, vc
            kk = kc.repeat_interleave(H // Hkv, dim=1)           # [t, H, hd]
            vv = vc.repeat_interleave(H // Hkv, dim=1)
            s = torch.einsum("hd,thd->ht", q[m], kk) / hd ** 0.5
            out[m] = torch.einsum("ht,thd->hd", torch.softmax(s, dim=-1), vv)
        attn = _linear(bf(out).view(M, H * hd), sd[p + "attention.wo.weight"])
        h = bf(x + attn)
        x13 = _norm_linear(h, sd[p + "ffn_norm.weight"], sd[p + "feed_forward.w13.weight"], cfg.norm_eps, fold)
        x1, x3 = x13.chunk(2, -1)
        ff = _linear(bf(bf(F.silu(x1)) * x3), sd[p + "feed_forward.w2.weight"])
        x = bf(h + ff)
    if not want_logits:
        return None
    return _norm_linear(x, sd["norm.weight"], sd["output.weight"], cfg.norm_eps, fold)


def prefill_right_aligned(sd, cfg, prompts: List[List[int]], cache: Cache, fold: bool = False) -> Tuple[T, T]:
    """Feed right-aligned prompts one position at a time (what the engine does; equivalent to the reference's ragged
    first pass, model_adapter.py:76-100).  Returns (logits of each row's last prompt token [M, V], next positions [M])."""
    M = len(prompts)
    maxlen = max(len(p) for p in prompts)
    last = None
    for j in range(maxlen):
        rows = [m for m in range(M) if j - (maxlen - len(prompts[m])) >= 0]
        tok = torch.tensor([prompts[m][j - (maxlen - len(prompts[m]))] for m in rows])
        pos = torch.tensor([j - (maxlen - len(prompts[m])) for m in rows])
        sub = Cache(cfg.n_layers, len(rows))
        for l in range(cfg.n_layers):
            for i, m in enumerate(rows):
                sub.k[l][i], sub.v[l][i] = cache.k[l][m], cache.v[l][m]
        lg = forward_tokens(sd, cfg, tok, pos, sub, fold, want_logits=(j == maxlen - 1))
        for l in range(cfg.n_layers):
            for i, m in enumerate(rows):
                cache.k[l][m], cache.v[l][m] = sub.k[l][i], sub.v[l][i]
        if j == maxlen - 1:
            last = lg
    return last, torch.tensor([len(p) for p in prompts])


def instruct_cfg(logits: T, g_text: float, g_image: float) -> T:
    """InBatchInstructCFGLogitsProcessor (logits_processor.py:312-336) on [3B, V] -> mixed [B, V]."""
    full, img, unc = logits.chunk(3)
    return unc + g_image * (img - unc) + g_text * (full - img)


def sample_step(logits3: T, q: T, temperature: float, top_p: Optional[float], g_text: float, g_image: float,
                allow_ids=None, key=None, past_ids=None, delta: float = 0.0):
    """One ImageDecoder step after the model (chameleon.py:313-327, generation.py:84-93): guidance mix -> watermark (called
    positionally on the first stream's input rows) -> allow-only -> temperature -> top-p -> softmax -> multinomial on the
    first stream.  Returns (tokens int64 [B], processed logits [B, V] before temperature)."""
    import numpy as np

    from oracle import wm_oracle as W
    B = logits3.shape[0] // 3
    lg = instruct_cfg(logits3.float(), g_text, g_image).numpy().copy()
    if key is not None:
        lg = W.process_logits(key, np.asarray(past_ids)[:B], lg, delta)
    if allow_ids is not None:
        keep = np.zeros(lg.shape[1], dtype=bool)
        keep[np.asarray(allow_ids)] = True
        lg[:, ~keep] = -np.inf
    tok = W.sample_rows(lg, np.asarray(q, dtype=np.float32), temperature, None, top_p)
    return tok, lg
