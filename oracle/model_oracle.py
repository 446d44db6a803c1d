"""TEST INFRASTRUCTURE ONLY -- torch-fp32 CPU restatement of the floating-point
stages of the Taming path (the integer stages live in wm_oracle.c).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module; the product package never does.

Every function works on plain ``{key: tensor}`` dicts that use the reference's
checkpoint key names, and cites the reference code it restates.  Parity status:
pinned -- tests/golden/make_golden.py runs the reference modules themselves
(imported from the reference checkout) on the same seeded weights and commits
their outputs; tests/test_oracle_golden.py compares this file against them.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import wm_oracle as W

T = torch.Tensor


# --------------------------------------------------------------------------- GPT

def gpt_step(sd: Dict[str, T], n_head: int, idx: T, past_k, past_v, past_length: int, cache=None):
    """GPT.forward_with_past for one new token per sequence.
    deps/taming/modules/transformer/mingpt.py:183-214 (+ Block :112-122, attention :69-95).

    idx [B,1] int64; past_k/past_v: lists (per layer) of [B,H,t,hd] or None.
    Returns logits [B,V], new_k, new_v (lists of [B,H,1,hd]).

    cache = (K, V), lists (per layer) of preallocated [B,H,Tmax,hd]: the new row is written at `past_length` and the attention
    reads the view [:, :, :past_length + 1] -- the same rows as the reference's `torch.cat` (mingpt.py:80-81, :192), without
    copying the whole cache twice per step (9.7 GB at 48 layers x 64 rows x 256 positions: 437 s -> ~70 s for a full-size loop);
    tests/test_oracle_golden.py pins the two forms against each other.  past_k / past_v are ignored then."""
    B = idx.shape[0]
    x = sd["tok_emb.weight"][idx[:, 0]] + sd["pos_emb"][0, past_length]  # :186-200
    d = x.shape[-1]
    hd = d // n_head
    n_layer = 0
    while f"blocks.{n_layer}.ln1.weight" in sd:
        n_layer += 1
    new_k, new_v = [], []
    for i in range(n_layer):
        p = f"blocks.{i}."
        h = F.layer_norm(x, (d,), sd[p + "ln1.weight"], sd[p + "ln1.bias"], 1e-5)
        k = F.linear(h, sd[p + "attn.key.weight"], sd[p + "attn.key.bias"]).view(B, 1, n_head, hd).transpose(1, 2)
        q = F.linear(h, sd[p + "attn.query.weight"], sd[p + "attn.query.bias"]).view(B, 1, n_head, hd).transpose(1, 2)
        v = F.linear(h, sd[p + "attn.value.weight"], sd[p + "attn.value.bias"]).view(B, 1, n_head, hd).transpose(1, 2)
        new_k.append(k)
        new_v.append(v)
        if cache is not None:
            cache[0][i][:, :, past_length] = k[:, :, 0]
            cache[1][i][:, :, past_length] = v[:, :, 0]
            kk, vv = cache[0][i][:, :, :past_length + 1], cache[1][i][:, :, :past_length + 1]
        elif past_k is not None:
            kk = torch.cat((past_k[i], k), dim=-2)
            vv = torch.cat((past_v[i], v), dim=-2)
        else:
            kk, vv = k, v
        att = (q @ kk.transpose(-2, -1)) * (1.0 / math.sqrt(hd))  # no mask with one query token
        att = F.softmax(att, dim=-1)
        y = (att @ vv).transpose(1, 2).contiguous().view(B, d)
        x = x + F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h2 = F.layer_norm(x, (d,), sd[p + "ln2.weight"], sd[p + "ln2.bias"], 1e-5)
        h2 = F.gelu(F.linear(h2, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
        x = x + F.linear(h2, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
    x = F.layer_norm(x, (d,), sd["ln_f.weight"], sd["ln_f.bias"], 1e-5)
    logits = F.linear(x, sd["head.weight"])
    return logits, new_k, new_v


@torch.no_grad()
def gpt_prefix(sd: Dict[str, T], n_head: int, idx: T, positions=None) -> T:
    """GPT.forward_with_past on a whole prefix with `past=None`: the causally masked branch of
    CausalSelfAttention.forward (deps/taming/modules/transformer/mingpt.py:69-95, mask :84-85) under
    forward_with_past :183-214 (position embeddings `pos_emb[:, :T]`, :199).  Mathematically the logits the
    incremental loop (`gpt_step`) produces at every position of a teacher-forced sequence, in ONE pass: what
    makes the full 48-layer model checkable in seconds on a CPU.

    idx [B,T] int64 -> logits [B, len(positions), V] (all T positions when `positions` is None)."""
    B, Tn = idx.shape
    x = sd["tok_emb.weight"][idx] + sd["pos_emb"][:, :Tn]                      # :186-199
    d = x.shape[-1]
    hd = d // n_head
    mask = torch.tril(torch.ones(Tn, Tn, dtype=torch.bool))
    i = 0
    while f"blocks.{i}.ln1.weight" in sd:
        p = f"blocks.{i}."
        h = F.layer_norm(x, (d,), sd[p + "ln1.weight"], sd[p + "ln1.bias"], 1e-5)
        k = F.linear(h, sd[p + "attn.key.weight"], sd[p + "attn.key.bias"]).view(B, Tn, n_head, hd).transpose(1, 2)
        q = F.linear(h, sd[p + "attn.query.weight"], sd[p + "attn.query.bias"]).view(B, Tn, n_head, hd).transpose(1, 2)
        v = F.linear(h, sd[p + "attn.value.weight"], sd[p + "attn.value.bias"]).view(B, Tn, n_head, hd).transpose(1, 2)
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
        att = att.masked_fill(~mask, float("-inf"))
        att = F.softmax(att, dim=-1)
        y = (att @ v).transpose(1, 2).contiguous().view(B, Tn, d)
        x = x + F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h2 = F.layer_norm(x, (d,), sd[p + "ln2.weight"], sd[p + "ln2.bias"], 1e-5)
        h2 = F.gelu(F.linear(h2, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
        x = x + F.linear(h2, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
        i += 1
    if positions is not None:
        x = x[:, list(positions)]
    x = F.layer_norm(x, (d,), sd["ln_f.weight"], sd["ln_f.bias"], 1e-5)
    return F.linear(x, sd["head.weight"])


def default_q_source(step: int, B: int, V: int) -> T:
    """The noise torch.multinomial(probs, 1) draws: one [B,V] Exp(1) tensor per step
    from the default CPU generator."""
    return torch.empty(B, V, dtype=torch.float32).exponential_(1)


@torch.no_grad()
def sample_with_past(sd: Dict[str, T], n_head: int, cond: T, steps: int, temperature=1.0, top_k=None,
                     top_p=None, key: Optional[W.KeyParams] = None, delta: float = 0.0,
                     q_source: Callable[[int, int, int], T] = default_q_source, record=None, static_cache: bool = False):
    """sample_with_past, deps/taming/modules/transformer/mingpt.py:326-368, with the
    logit processor of gentime_watermark.py:229-271 when `key` is given.
    cond [B,1] int64.  Returns int64 [B, steps].  static_cache: see gpt_step (full-size loops)."""
    sample = cond.clone()
    cond_len = cond.shape[1]
    assert cond_len == 1
    pk = pv = None
    x = cond
    V = sd["head.weight"].shape[0]
    cache = None
    if static_cache:
        Bc, d = cond.shape[0], sd["tok_emb.weight"].shape[1]
        nl = 0
        while f"blocks.{nl}.ln1.weight" in sd:
            nl += 1
        cache = ([torch.zeros(Bc, n_head, steps, d // n_head) for _ in range(nl)], [torch.zeros(Bc, n_head, steps, d // n_head) for _ in range(nl)])
    for n in range(steps):
        logits, nk, nv = gpt_step(sd, n_head, x, pk, pv, n + cond_len - 1, cache=cache)
        if cache is not None:
            pass
        elif pk is None:
            pk, pv = nk, nv
        else:
            pk = [torch.cat((a, b), dim=-2) for a, b in zip(pk, nk)]
            pv = [torch.cat((a, b), dim=-2) for a, b in zip(pv, nv)]
        lg = logits.numpy()
        if key is not None:
            lg = W.process_logits(key, sample.numpy(), lg, delta)
        q = q_source(n, lg.shape[0], V)
        tok = W.sample_rows(lg, q.numpy(), temperature, top_k, top_p)
        if record is not None:
            record.append(dict(logits=logits.numpy().copy(), biased=np.array(lg), q=q.numpy().copy(), tok=tok.copy()))
        x = torch.from_numpy(tok).view(-1, 1)
        sample = torch.cat((sample, x), dim=1)
    return sample[:, cond_len:]


# ------------------------------------------------------------------------- VQGAN

def _gn_swish(x: T, w: T, b: T, swish: bool = True) -> T:
    """Normalize (GroupNorm 32 groups, eps 1e-6) + nonlinearity; model.py:30-36."""
    h = F.group_norm(x, 32, w, b, 1e-6)
    return h * torch.sigmoid(h) if swish else h


def _resnet(sd, p: str, x: T) -> T:
    """ResnetBlock.forward, model.py:118-138 (temb is None on this path)."""
    h = _gn_swish(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = _gn_swish(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "nin_shortcut.weight"], sd[p + "nin_shortcut.bias"])
    return x + h


def _attn(sd, p: str, x: T) -> T:
    """AttnBlock.forward, model.py:169-193."""
    h = _gn_swish(x, sd[p + "norm.weight"], sd[p + "norm.bias"], swish=False)
    q = F.conv2d(h, sd[p + "q.weight"], sd[p + "q.bias"])
    k = F.conv2d(h, sd[p + "k.weight"], sd[p + "k.bias"])
    v = F.conv2d(h, sd[p + "v.weight"], sd[p + "v.bias"])
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    h = F.conv2d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return x + h


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


@torch.no_grad()
def decoder_forward(sd: Dict[str, T], num_resolutions: int, num_res_blocks: int, z: T) -> T:
    """Decoder.forward, model.py:507-538.  `sd` keys are relative to ``decoder.``."""
    h = F.conv2d(z, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    h = _resnet(sd, "mid.block_1.", h)
    h = _attn(sd, "mid.attn_1.", h)
    h = _resnet(sd, "mid.block_2.", h)
    for lvl in reversed(range(num_resolutions)):
        for b in range(num_res_blocks + 1):
            h = _resnet(sd, f"up.{lvl}.block.{b}.", h)
            if f"up.{lvl}.attn.{b}.norm.weight" in sd:
                h = _attn(sd, f"up.{lvl}.attn.{b}.", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")  # Upsample, model.py:50-54
            h = F.conv2d(h, sd[f"up.{lvl}.upsample.conv.weight"], sd[f"up.{lvl}.upsample.conv.bias"], padding=1)
    h = _gn_swish(h, sd["norm_out.weight"], sd["norm_out.bias"])
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


@torch.no_grad()
def encoder_forward(sd: Dict[str, T], num_resolutions: int, num_res_blocks: int, x: T) -> T:
    """Encoder.forward, model.py:407-434.  `sd` keys are relative to ``encoder.``."""
    h = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    for lvl in range(num_resolutions):
        for b in range(num_res_blocks):
            h = _resnet(sd, f"down.{lvl}.block.{b}.", h)
            if f"down.{lvl}.attn.{b}.norm.weight" in sd:
                h = _attn(sd, f"down.{lvl}.attn.{b}.", h)
        if lvl != num_resolutions - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)  # Downsample, model.py:69-73
            h = F.conv2d(h, sd[f"down.{lvl}.downsample.conv.weight"], sd[f"down.{lvl}.downsample.conv.bias"], stride=2)
    h = _resnet(sd, "mid.block_1.", h)
    h = _attn(sd, "mid.attn_1.", h)
    h = _resnet(sd, "mid.block_2.", h)
    h = _gn_swish(h, sd["norm_out.weight"], sd["norm_out.bias"])
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


@torch.no_grad()
def codes_to_images(sd: Dict[str, T], cfg, codes: T) -> T:
    """TamingARMMWrapper.codes_to_images (wmar/models/taming_wrapper.py:79-84) ->
    decode_to_img (cond_transformer.py:186-192) -> get_codebook_entry (quantize.py:316-331)
    -> VQModel.decode (vqgan.py:70-73).  `sd` keys relative to ``first_stage_model.``."""
    B = codes.shape[0]
    S = cfg.codes_size
    zq = sd["quantize.embedding.weight"][codes.reshape(-1)].view(B, S, S, cfg.embed_dim).permute(0, 3, 1, 2).contiguous()
    z = F.conv2d(zq, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    img = decoder_forward(_sub(sd, "decoder."), cfg.num_resolutions, cfg.num_res_blocks, z)
    return img.clamp(-1, 1)


@torch.no_grad()
def encode_prequant(sd: Dict[str, T], cfg, images: T) -> T:
    """Encoder + quant_conv (vqgan.py:64-66): the vectors handed to the quantizer, [B*S*S, e_dim]."""
    h = encoder_forward(_sub(sd, "encoder."), cfg.num_resolutions, cfg.num_res_blocks, images)
    h = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    return h.permute(0, 2, 3, 1).contiguous().view(-1, cfg.embed_dim)


@torch.no_grad()
def quantize_argmin(emb: T, z_flat: T) -> T:
    """VectorQuantizer2.forward distance + argmin, quantize.py:277-285."""
    d = torch.sum(z_flat ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1) - 2 * torch.einsum(
        "bd,dn->bn", z_flat, emb.t().contiguous())
    return torch.argmin(d, dim=1)


@torch.no_grad()
def images_to_codes(sd: Dict[str, T], cfg, images: T) -> T:
    """TamingARMMWrapper.images_to_codes (taming_wrapper.py:88-92) -> encode_to_z
    (cond_transformer.py:170-174) -> VQModel.encode (vqgan.py:64-68)."""
    z = encode_prequant(sd, cfg, images)
    return quantize_argmin(sd["quantize.embedding.weight"], z).view(images.shape[0], -1)


# -------------------------------------------------------------------- metrics (A13)

def chw_to_uint8(x: np.ndarray) -> np.ndarray:
    """chw_to_pillow's arithmetic (wmar/utils/utils.py:74-80): HWC uint8, round-half-even."""
    x = (255 * ((x.transpose(1, 2, 0) + 1.0) / 2.0)).clip(0, 255)
    return np.round(x).astype(np.uint8)


def psnr_uint8(a: np.ndarray, b: np.ndarray) -> float:
    """compute_psnr, wmar/utils/metrics.py:20-22."""
    mse = np.mean((a * 1.0 - b * 1.0) ** 2)
    return float(10 * np.log10(255.0 ** 2 / mse))
