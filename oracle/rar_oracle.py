"""TEST INFRASTRUCTURE ONLY -- torch-fp32 CPU restatement of the RAR generator and the
MaskGIT-VQGAN tokenizer.  Only tests/, smoke() and bench.py's
cpu_baseline may import it.  Parity status: pinned -- tests/golden/make_golden.py runs the
reference's RAR / maskgit_vqgan modules on the same seeded weights (timm's `Mlp` is a
third-party class absent here; a 10-line stand-in with its published semantics
fc1 -> GELU -> fc2 is used while generating the vectors, so the MLP is pinned up to that
restatement).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import wm_oracle as W

T = torch.Tensor


def _ln(x, w, b, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def rar_position(sd: Dict[str, T], cfg, tok_emb: T, cond_emb: T, p: int, kc, vc):
    """One sequence position through all blocks (RAR.forward_fn with a warm KV cache,
    deps/rar/modeling/rar.py:319-405; Block :138-183; Attention :56-118; FinalLayer :123-134).
    tok_emb [M,d]: embedding of the token AT position p (cls_token for p = 0);
    cond_emb [M,d]: embedding of the row's condition id.  kc/vc: per-layer caches [M,H,t,hd] or None.
    Returns logits [M,V] and the updated caches."""
    d, H, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
    M = tok_emb.shape[0]
    x = tok_emb + sd["pos_embed"][0, p]                                     # :352-372
    if p >= 1:
        x = x + sd["target_aware_pos_embed"][0, p + 1]                      # :374-378 (prefix = 2)
    c = cond_emb + sd["timesteps_embeddings"][0, p]                         # :384
    sc = F.silu(c)
    nk, nv = [], []
    for i in range(cfg.num_hidden_layers):
        pre = f"blocks.{i}."
        mod = F.linear(sc, sd[pre + "adaLN_modulation.1.weight"], sd[pre + "adaLN_modulation.1.bias"])
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=-1)
        h = _ln(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]) * (1 + scale_msa) + shift_msa
        qkv = F.linear(h, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"]).view(M, 3, H, hd)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = _ln(q, sd[pre + "attn.q_norm.weight"], sd[pre + "attn.q_norm.bias"])
        k = _ln(k, sd[pre + "attn.k_norm.weight"], sd[pre + "attn.k_norm.bias"])
        k = k.unsqueeze(2); v = v.unsqueeze(2)                              # [M,H,1,hd]
        kk = k if kc is None else torch.cat((kc[i], k), dim=2)
        vv = v if vc is None else torch.cat((vc[i], v), dim=2)
        nk.append(kk); nv.append(vv)
        att = torch.softmax((q.unsqueeze(2) @ kk.transpose(-2, -1)) * (hd ** -0.5), dim=-1)   # SDPA, one query
        y = (att @ vv).reshape(M, d)
        x = x + gate_msa * F.linear(y, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
        h2 = _ln(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]) * (1 + scale_mlp) + shift_mlp
        h2 = F.gelu(F.linear(h2, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
        x = x + gate_mlp * F.linear(h2, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    fmod = F.linear(sc, sd["adaln_before_head.adaLN_modulation.1.weight"], sd["adaln_before_head.adaLN_modulation.1.bias"])
    scale, shift = fmod.chunk(2, dim=-1)                                    # FinalLayer: scale first
    x = F.layer_norm(x, (d,), None, None, 1e-6) * (1 + scale) + shift
    logits = F.linear(x, sd["lm_head.weight"], sd["lm_head.bias"])
    return logits, nk, nv


def cfg_scales(steps: int, guidance_scale: float, guidance_scale_pow: float) -> T:
    """rar.py:430-434, evaluated exactly as the reference does (fp32 tensors)."""
    out = []
    for step in range(steps):
        scale_pow = torch.ones((1)) * guidance_scale_pow
        scale_step = (1 - torch.cos(((step / steps) ** scale_pow) * torch.pi)) * 1 / 2
        out.append((guidance_scale - 1) * scale_step + 1)
    return torch.cat(out)


def default_q_source(step, B, V):
    return torch.empty(B, V, dtype=torch.float32).exponential_(1)


@torch.no_grad()
def generate(sd: Dict[str, T], cfg, condition: T, guidance_scale=4.0, guidance_scale_pow=0.0,
             randomize_temperature=1.0, key: Optional[W.KeyParams] = None, delta: float = 0.0,
             q_source: Callable = default_q_source, record=None, draw_drop_mask: bool = True,
             sampler: Optional[Callable] = None, max_steps: Optional[int] = None) -> T:
    """RAR.generate (rar.py:408-459) with classifier-free guidance and the logit processor.
    condition int64 [B] class ids.  Returns int64 [B, image_seq_len].
    max_steps (tests at full depth): stop after that many tokens -- the guidance schedule still spans image_seq_len."""
    B = condition.shape[0]
    V, L = cfg.codebook_size, cfg.image_seq_len
    if draw_drop_mask:
        # preprocess_condition draws a label-drop mask (cond_drop_prob = 0.0: never set) -- the draw
        # still advances the generator that torch.multinomial uses afterwards (rar.py:305)
        torch.rand(B, 1, dtype=torch.float)
    cond = condition.view(-1) + V + 1                                       # preprocess_condition :303-308
    none = torch.full_like(cond, cfg.none_condition_id)
    both = torch.cat([cond, none])
    cond_emb = sd["embeddings.weight"][both]
    scales = cfg_scales(L, guidance_scale, guidance_scale_pow)
    # position 0: the cls token (its logits are discarded)
    cls = sd["cls_token"][0, 0].expand(2 * B, -1)
    _, kc, vc = rar_position(sd, cfg, cls, cond_emb, 0, None, None)
    ids = torch.zeros(B, 0, dtype=torch.long)
    tok_emb = cond_emb                                                      # position 1 holds the condition token
    for step in range(L if max_steps is None else min(L, max_steps)):
        logits, kc, vc = rar_position(sd, cfg, tok_emb, cond_emb, step + 1, kc, vc)
        cl, ul = logits[:B], logits[B:]
        if guidance_scale != 0:
            mixed = ul + (cl - ul) * scales[step]
        else:
            mixed = cl
        if sampler is not None:
            # replacement sampler (Gumbel key, row G1): (mixed logits [B, V], step) -> int64 [B]
            t = sampler(mixed, step).to(torch.long)
            ids = torch.cat([ids, t.view(-1, 1)], dim=1)
            tok_emb = sd["embeddings.weight"][torch.cat([t, t])]
            continue
        lg = mixed.numpy()
        if key is not None:
            lg = W.process_logits(key, ids.numpy(), lg, delta)
        q = q_source(step, B, V)
        tok = W.sample_rows(lg, q.numpy(), randomize_temperature, None, None)
        if record is not None:
            record.append(dict(cond_logits=cl.numpy().copy(), uncond_logits=ul.numpy().copy(), biased=np.array(lg), q=q.numpy().copy(), tok=tok.copy()))
        t = torch.from_numpy(tok)
        ids = torch.cat([ids, t.view(-1, 1)], dim=1)
        tok_emb = sd["embeddings.weight"][torch.cat([t, t])]
    return ids


# ------------------------------------------------------------------ MaskGIT-VQGAN tokenizer

def _mres(sd, p, x):
    """ResnetBlock.forward, maskgit_vqgan.py:69-87 (the 1x1 shortcut acts on the block OUTPUT)."""
    h = F.silu(F.group_norm(x, 32, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6))
    h = F.conv2d(h, sd[p + "conv1.weight"], None, padding=1)
    h = F.silu(F.group_norm(h, 32, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6))
    h = F.conv2d(h, sd[p + "conv2.weight"], None, padding=1)
    res = x
    if p + "nin_shortcut.weight" in sd:
        res = F.conv2d(h, sd[p + "nin_shortcut.weight"], None)
    return h + res


@torch.no_grad()
def maskgit_decode(sd, cfg, codes: T) -> T:
    """RarARMMWrapper.codes_to_images (rar_wrapper.py:108-118) -> PretrainedTokenizer.decode
    (titok.py:81-85) -> Decoder.forward (maskgit_vqgan.py:222-237).  Returns pixels in [-1, 1]."""
    B = codes.shape[0]
    S = cfg.codes_size
    z = sd["quantize.embedding.weight"][codes.reshape(-1)].view(B, S, S, -1).permute(0, 3, 1, 2).contiguous()
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    for b in range(cfg.num_res_blocks):
        h = _mres(sd, f"decoder.mid.{b}.", h)
    for lvl in reversed(range(cfg.num_resolutions)):
        for b in range(cfg.num_res_blocks):
            h = _mres(sd, f"decoder.up.{lvl}.block.{b}.", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"decoder.up.{lvl}.upsample_conv.weight"], sd[f"decoder.up.{lvl}.upsample_conv.bias"], padding=1)
    h = F.silu(F.group_norm(h, 32, sd["decoder.norm_out.weight"], sd["decoder.norm_out.bias"], 1e-6))
    img = F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
    img = torch.clamp(img, 0.0, 1.0)
    return torch.clamp(img * 2.0 - 1.0, -1.0, 1.0)


@torch.no_grad()
def maskgit_prequant(sd, cfg, images: T) -> T:
    """images in [-1,1] -> encoder output rows [B*S*S, z] (rar_wrapper.py:120-128, maskgit_vqgan.py:173-185)."""
    x = (images + 1.0) / 2.0
    h = F.conv2d(x, sd["encoder.conv_in.weight"], None, padding=1)
    for lvl in range(cfg.num_resolutions):
        for b in range(cfg.num_res_blocks):
            h = _mres(sd, f"encoder.down.{lvl}.block.{b}.", h)
        if lvl != cfg.num_resolutions - 1:
            h = F.avg_pool2d(h, kernel_size=2, stride=2)
    for b in range(cfg.num_res_blocks):
        h = _mres(sd, f"encoder.mid.{b}.", h)
    h = F.silu(F.group_norm(h, 32, sd["encoder.norm_out.weight"], sd["encoder.norm_out.bias"], 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"])
    return h.permute(0, 2, 3, 1).contiguous().view(-1, h.shape[1])


@torch.no_grad()
def maskgit_argmin(emb: T, z: T) -> T:
    """VectorQuantizer.compute_distances + argmin (maskgit_vqgan.py:286-321)."""
    d = torch.addmm(z.pow(2.0).sum(dim=1, keepdim=True) + emb.t().pow(2.0).sum(dim=0, keepdim=True), z, emb.t(), alpha=-2.0)
    return torch.argmin(d, dim=1)


@torch.no_grad()
def maskgit_encode(sd, cfg, images: T) -> T:
    z = maskgit_prequant(sd, cfg, images)
    return maskgit_argmin(sd["quantize.embedding.weight"], z).view(images.shape[0], -1)
