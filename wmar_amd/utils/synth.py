"""Model configs and synthetic (seeded, random-init) checkpoints in the reference's layout.

There is no network and no pretrained checkpoint on the GPU box, so benchmarks and
parity tests run on random-init weights of the real architectures.  The tensors
carry exactly the key names and shapes of the reference's Lightning checkpoint
(``net2net.ckpt["state_dict"]``; reference modules:
deps/taming/modules/transformer/mingpt.py:125-146 (GPT),
deps/taming/modules/diffusionmodules/model.py:343-538 (Encoder / Decoder),
deps/taming/models/vqgan.py:30-38 (quantize / quant_conv / post_quant_conv)),
which tests/golden/make_golden.py pins with ``load_state_dict(strict=True)``
against the reference module trees.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Tuple

import torch


@dataclasses.dataclass
class GPTConfig:
    vocab_size: int = 16384
    block_size: int = 256
    n_layer: int = 48
    n_head: int = 24
    n_embd: int = 1536

    @property
    def head_dim(self) -> int:
        return self.n_embd // self.n_head


@dataclasses.dataclass
class VQConfig:
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = (16,)
    resolution: int = 256
    in_channels: int = 3
    out_ch: int = 3
    z_channels: int = 256
    embed_dim: int = 256
    n_embed: int = 16384

    @property
    def num_resolutions(self) -> int:
        return len(self.ch_mult)

    @property
    def codes_size(self) -> int:
        return self.resolution // (2 ** (self.num_resolutions - 1))


TAMING_GPT = GPTConfig()
TAMING_VQ = VQConfig()
# the reduced Taming model the reference's own generate.py was run on for tests/golden/harness_vectors.npz (seeds 21)
HARNESS_GPT = dict(vocab_size=16384, block_size=64, n_layer=2, n_head=4, n_embd=128)
HARNESS_VQ = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32, z_channels=16, embed_dim=8,
                  n_embed=16384)


def gpt_shapes(cfg: GPTConfig, with_mask: bool = False) -> Dict[str, Tuple[int, ...]]:
    d, v = cfg.n_embd, cfg.vocab_size
    s: Dict[str, Tuple[int, ...]] = {"tok_emb.weight": (v, d), "pos_emb": (1, cfg.block_size, d)}
    for i in range(cfg.n_layer):
        p = f"blocks.{i}."
        for ln in ("ln1", "ln2"):
            s[p + ln + ".weight"] = (d,)
            s[p + ln + ".bias"] = (d,)
        for lin in ("key", "query", "value", "proj"):
            s[p + f"attn.{lin}.weight"] = (d, d)
            s[p + f"attn.{lin}.bias"] = (d,)
        if with_mask:
            s[p + "attn.mask"] = (1, 1, cfg.block_size, cfg.block_size)
        s[p + "mlp.0.weight"] = (4 * d, d)
        s[p + "mlp.0.bias"] = (4 * d,)
        s[p + "mlp.2.weight"] = (d, 4 * d)
        s[p + "mlp.2.bias"] = (d,)
    s["ln_f.weight"] = (d,)
    s["ln_f.bias"] = (d,)
    s["head.weight"] = (v, d)
    return s


def _res(s, p, cin, cout):
    s[p + "norm1.weight"] = (cin,)
    s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3)
    s[p + "conv1.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,)
    s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3)
    s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "nin_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "nin_shortcut.bias"] = (cout,)


def _attn(s, p, c):
    s[p + "norm.weight"] = (c,)
    s[p + "norm.bias"] = (c,)
    for n in ("q", "k", "v", "proj_out"):
        s[p + n + ".weight"] = (c, c, 1, 1)
        s[p + n + ".bias"] = (c,)


def encoder_shapes(cfg: VQConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    s["conv_in.weight"] = (cfg.ch, cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (cfg.ch,)
    in_mult = (1,) + tuple(cfg.ch_mult)
    res = cfg.resolution
    block_in = cfg.ch
    for lvl in range(cfg.num_resolutions):
        block_in = cfg.ch * in_mult[lvl]
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks):
            _res(s, f"down.{lvl}.block.{b}.", block_in, block_out)
            block_in = block_out
            if res in cfg.attn_resolutions:
                _attn(s, f"down.{lvl}.attn.{b}.", block_in)
        if lvl != cfg.num_resolutions - 1:
            s[f"down.{lvl}.downsample.conv.weight"] = (block_in, block_in, 3, 3)
            s[f"down.{lvl}.downsample.conv.bias"] = (block_in,)
            res //= 2
    _res(s, "mid.block_1.", block_in, block_in)
    _attn(s, "mid.attn_1.", block_in)
    _res(s, "mid.block_2.", block_in, block_in)
    s["norm_out.weight"] = (block_in,)
    s["norm_out.bias"] = (block_in,)
    s["conv_out.weight"] = (cfg.z_channels, block_in, 3, 3)
    s["conv_out.bias"] = (cfg.z_channels,)
    return s


def decoder_shapes(cfg: VQConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    block_in = cfg.ch * cfg.ch_mult[-1]
    res = cfg.resolution // 2 ** (cfg.num_resolutions - 1)
    s["conv_in.weight"] = (block_in, cfg.z_channels, 3, 3)
    s["conv_in.bias"] = (block_in,)
    _res(s, "mid.block_1.", block_in, block_in)
    _attn(s, "mid.attn_1.", block_in)
    _res(s, "mid.block_2.", block_in, block_in)
    for lvl in reversed(range(cfg.num_resolutions)):
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks + 1):
            _res(s, f"up.{lvl}.block.{b}.", block_in, block_out)
            block_in = block_out
            if res in cfg.attn_resolutions:
                _attn(s, f"up.{lvl}.attn.{b}.", block_in)
        if lvl != 0:
            s[f"up.{lvl}.upsample.conv.weight"] = (block_in, block_in, 3, 3)
            s[f"up.{lvl}.upsample.conv.bias"] = (block_in,)
            res *= 2
    s["norm_out.weight"] = (block_in,)
    s["norm_out.bias"] = (block_in,)
    s["conv_out.weight"] = (cfg.out_ch, block_in, 3, 3)
    s["conv_out.bias"] = (cfg.out_ch,)
    return s


def vq_shapes(cfg: VQConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    for k, v in encoder_shapes(cfg).items():
        s["encoder." + k] = v
    for k, v in decoder_shapes(cfg).items():
        s["decoder." + k] = v
    s["quantize.embedding.weight"] = (cfg.n_embed, cfg.embed_dim)
    s["quant_conv.weight"] = (cfg.embed_dim, cfg.z_channels, 1, 1)
    s["quant_conv.bias"] = (cfg.embed_dim,)
    s["post_quant_conv.weight"] = (cfg.z_channels, cfg.embed_dim, 1, 1)
    s["post_quant_conv.bias"] = (cfg.z_channels,)
    return s


def _fill(shapes, gen: torch.Generator, device, kind: str, logit_scale: float = 1.0):
    """Seeded init.  `kind` picks the distribution family per tensor role."""
    out = {}
    for k, shp in shapes.items():
        leaf = k.rsplit(".", 1)[-1]
        is_norm = any(t in k for t in ("ln1.", "ln2.", "ln_f.", "norm")) and leaf in ("weight", "bias")
        if k.endswith("attn.mask"):
            n = shp[-1]
            t = torch.tril(torch.ones(n, n)).view(shp)
        elif is_norm:
            t = torch.randn(shp, generator=gen) * 0.1
            if leaf == "weight":
                t = t + 1.0
        elif leaf == "bias":
            t = torch.randn(shp, generator=gen) * 0.02
        elif k == "pos_emb":
            t = torch.randn(shp, generator=gen) * 0.02
        elif kind == "gpt":
            t = torch.randn(shp, generator=gen) * 0.02
            if k == "head.weight":
                t = t * logit_scale
        elif k == "quantize.embedding.weight":
            t = torch.rand(shp, generator=gen) * 2.0 - 1.0
        else:  # conv weights: kaiming-uniform-like fan-in scaling
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            bound = (3.0 / fan_in) ** 0.5
            t = (torch.rand(shp, generator=gen) * 2.0 - 1.0) * bound
        out[k] = t.to(torch.float32).to(device)
    return out


def synth_gpt_state(cfg: GPTConfig, seed: int = 0, device="cpu", logit_scale: float = 1.0,
                    with_mask: bool = False) -> Dict[str, torch.Tensor]:
    """Random-init GPT tensors keyed like ``transformer.*`` minus the prefix.

    ``logit_scale`` multiplies ``head.weight`` so that logits are not flat (flat logits make
    the watermark saturate: every sampled token green)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return _fill(gpt_shapes(cfg, with_mask), g, device, "gpt", logit_scale)


def synth_vq_state(cfg: VQConfig, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """Random-init VQGAN tensors keyed like ``first_stage_model.*`` minus the prefix."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 7919)
    return _fill(vq_shapes(cfg), g, device, "vq")


def synth_gpt_state_fast(cfg: GPTConfig, seed: int = 0, device="cuda", logit_scale: float = 1.0):
    """Same layout, drawn on `device` directly (full-size benchmark weights: 1.4 G params)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for k, shp in gpt_shapes(cfg).items():
        leaf = k.rsplit(".", 1)[-1]
        is_norm = any(t in k for t in ("ln1.", "ln2.", "ln_f."))
        if is_norm:
            t = torch.randn(shp, generator=g, device=device) * 0.1
            if leaf == "weight":
                t += 1.0
        elif leaf == "bias":
            t = torch.randn(shp, generator=g, device=device) * 0.02
        else:
            t = torch.randn(shp, generator=g, device=device) * 0.02
            if k == "head.weight":
                t *= logit_scale
        out[k] = t
    return out


def synth_vq_state_fast(cfg: VQConfig, seed: int = 0, device="cuda"):
    g = torch.Generator(device=device)
    g.manual_seed(seed + 7919)
    out = {}
    for k, shp in vq_shapes(cfg).items():
        leaf = k.rsplit(".", 1)[-1]
        if "norm" in k:
            t = torch.randn(shp, generator=g, device=device) * 0.1
            if leaf == "weight":
                t += 1.0
        elif leaf == "bias":
            t = torch.randn(shp, generator=g, device=device) * 0.02
        elif k == "quantize.embedding.weight":
            t = torch.rand(shp, generator=g, device=device) * 2.0 - 1.0
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = (torch.rand(shp, generator=g, device=device) * 2.0 - 1.0) * (3.0 / fan_in) ** 0.5
        out[k] = t
    return out


# ----------------------------------------------------------------------------- RAR
@dataclasses.dataclass
class RARConfig:
    """deps/rar/modeling/rar.py:179-250 (dims: wmar/models/rar_wrapper.py:43-52, rar.yaml)."""
    hidden_size: int = 1280
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    intermediate_size: int = 5120
    image_seq_len: int = 256
    codebook_size: int = 1024
    condition_num_classes: int = 1000

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def n_embeddings(self) -> int:
        return self.codebook_size + 1 + self.condition_num_classes + 1

    @property
    def none_condition_id(self) -> int:
        return self.condition_num_classes + self.codebook_size + 1


@dataclasses.dataclass
class MaskgitVQConfig:
    """deps/rar/modeling/titok.py:44-56 (PretrainedTokenizer) / maskgit_vqgan.py."""
    hidden_channels: int = 128
    channel_mult: Tuple[int, ...] = (1, 1, 2, 2, 4)
    num_res_blocks: int = 2
    resolution: int = 256
    num_channels: int = 3
    z_channels: int = 256
    num_embeddings: int = 1024

    @property
    def num_resolutions(self) -> int:
        return len(self.channel_mult)

    @property
    def codes_size(self) -> int:
        return self.resolution // (2 ** (self.num_resolutions - 1))


RAR_XL = RARConfig()
MASKGIT_VQ = MaskgitVQConfig()


def rar_shapes(cfg: RARConfig) -> Dict[str, Tuple[int, ...]]:
    d, L, hd = cfg.hidden_size, cfg.image_seq_len, cfg.head_dim
    s: Dict[str, Tuple[int, ...]] = {"cls_token": (1, 1, d)}
    for i in range(cfg.num_hidden_layers):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (d,); s[p + "norm1.bias"] = (d,)
        s[p + "attn.qkv.weight"] = (3 * d, d); s[p + "attn.qkv.bias"] = (3 * d,)
        s[p + "attn.q_norm.weight"] = (hd,); s[p + "attn.q_norm.bias"] = (hd,)
        s[p + "attn.k_norm.weight"] = (hd,); s[p + "attn.k_norm.bias"] = (hd,)
        s[p + "attn.proj.weight"] = (d, d); s[p + "attn.proj.bias"] = (d,)
        s[p + "norm2.weight"] = (d,); s[p + "norm2.bias"] = (d,)
        s[p + "mlp.fc1.weight"] = (cfg.intermediate_size, d); s[p + "mlp.fc1.bias"] = (cfg.intermediate_size,)
        s[p + "mlp.fc2.weight"] = (d, cfg.intermediate_size); s[p + "mlp.fc2.bias"] = (d,)
        s[p + "adaLN_modulation.1.weight"] = (6 * d, d); s[p + "adaLN_modulation.1.bias"] = (6 * d,)
    s["embeddings.weight"] = (cfg.n_embeddings, d)
    s["pos_embed"] = (1, L + 1024, d)
    s["target_aware_pos_embed"] = (1, L + 1024, d)
    s["timesteps_embeddings"] = (1, L + 100, d)
    s["adaln_before_head.adaLN_modulation.1.weight"] = (2 * d, d)
    s["adaln_before_head.adaLN_modulation.1.bias"] = (2 * d,)
    s["lm_head.weight"] = (cfg.codebook_size, d)
    s["lm_head.bias"] = (cfg.codebook_size,)
    return s


def _mres(s, p, cin, cout):
    s[p + "norm1.weight"] = (cin,); s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3)
    s[p + "norm2.weight"] = (cout,); s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3)
    if cin != cout:
        s[p + "nin_shortcut.weight"] = (cout, cout, 1, 1)   # applied to the block OUTPUT (maskgit_vqgan.py:69-87)


def maskgit_shapes(cfg: MaskgitVQConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    hc, mult, R = cfg.hidden_channels, cfg.channel_mult, cfg.num_resolutions
    s["encoder.conv_in.weight"] = (hc, cfg.num_channels, 3, 3)
    in_mult = (1,) + tuple(mult)
    for lvl in range(R):
        bi, bo = hc * in_mult[lvl], hc * mult[lvl]
        for b in range(cfg.num_res_blocks):
            _mres(s, f"encoder.down.{lvl}.block.{b}.", bi, bo)
            bi = bo
    mid = hc * mult[-1]
    for b in range(cfg.num_res_blocks):
        _mres(s, f"encoder.mid.{b}.", mid, mid)
    s["encoder.norm_out.weight"] = (mid,); s["encoder.norm_out.bias"] = (mid,)
    s["encoder.conv_out.weight"] = (cfg.z_channels, mid, 1, 1); s["encoder.conv_out.bias"] = (cfg.z_channels,)
    s["decoder.conv_in.weight"] = (mid, cfg.z_channels, 3, 3); s["decoder.conv_in.bias"] = (mid,)
    for b in range(cfg.num_res_blocks):
        _mres(s, f"decoder.mid.{b}.", mid, mid)
    for lvl in range(R):
        bi = hc * mult[-1] if lvl == R - 1 else hc * mult[lvl + 1]
        bo = hc * mult[lvl]
        for b in range(cfg.num_res_blocks):
            _mres(s, f"decoder.up.{lvl}.block.{b}.", bi, bo)
            bi = bo
        if lvl != 0:
            s[f"decoder.up.{lvl}.upsample_conv.weight"] = (bo, bo, 3, 3)
            s[f"decoder.up.{lvl}.upsample_conv.bias"] = (bo,)
    s["decoder.norm_out.weight"] = (hc * mult[0],); s["decoder.norm_out.bias"] = (hc * mult[0],)
    s["decoder.conv_out.weight"] = (cfg.num_channels, hc * mult[0], 3, 3); s["decoder.conv_out.bias"] = (cfg.num_channels,)
    s["quantize.embedding.weight"] = (cfg.num_embeddings, cfg.z_channels)
    return s


def _fill_rar(shapes, gen, device, logit_scale):
    out = {}
    for k, shp in shapes.items():
        leaf = k.rsplit(".", 1)[-1]
        is_norm = ("norm" in k) and leaf in ("weight", "bias")
        if is_norm:
            t = torch.randn(shp, generator=gen, device=gen.device) * 0.1
            if leaf == "weight":
                t = t + 1.0
        elif leaf == "bias":
            t = torch.randn(shp, generator=gen, device=gen.device) * 0.02
        else:
            t = torch.randn(shp, generator=gen, device=gen.device) * 0.02
            if k == "lm_head.weight":
                t = t * logit_scale
            if "adaLN_modulation" in k:
                t = t * 2.0   # the reference zero-inits these; non-zero so that the modulation path is exercised
        out[k] = t.to(torch.float32).to(device)
    return out


def synth_rar_state(cfg: RARConfig, seed: int = 0, device="cpu", logit_scale: float = 1.0, gen_device="cpu"):
    g = torch.Generator(device=gen_device)
    g.manual_seed(seed + 4242)
    return _fill_rar(rar_shapes(cfg), g, device, logit_scale)


def synth_maskgit_state(cfg: MaskgitVQConfig, seed: int = 0, device="cpu", gen_device="cpu"):
    g = torch.Generator(device=gen_device)
    g.manual_seed(seed + 991)
    out = {}
    for k, shp in maskgit_shapes(cfg).items():
        leaf = k.rsplit(".", 1)[-1]
        if "norm" in k:
            t = torch.randn(shp, generator=g, device=gen_device) * 0.1
            if leaf == "weight":
                t = t + 1.0
        elif leaf == "bias":
            t = torch.randn(shp, generator=g, device=gen_device) * 0.02
        elif k == "quantize.embedding.weight":
            t = torch.rand(shp, generator=g, device=gen_device) * 2.0 - 1.0
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = (torch.rand(shp, generator=g, device=gen_device) * 2.0 - 1.0) * (3.0 / fan_in) ** 0.5
        out[k] = t.to(torch.float32).to(device)
    return out


# ----------------------------------------------------------------------------- Chameleon
@dataclasses.dataclass
class ChameleonConfig:
    """deps/chameleon/inference/transformer.py:18-31 (ModelArgs).  Defaults: the 7B model (Anole-7B)."""
    dim: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    n_kv_heads: int = 32
    vocab_size: int = 65536
    ffn_dim_multiplier: float = 1.0
    multiple_of: int = 256
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    qk_normalization: bool = True
    swin_norm: bool = False

    @property
    def head_dim(self) -> int:
        return self.dim // self.n_heads

    @property
    def ffn_hidden(self) -> int:
        """FeedForward.__init__, transformer.py:175-179."""
        hidden = int(2 * (4 * self.dim) / 3)
        if self.ffn_dim_multiplier is not None:
            hidden = int(self.ffn_dim_multiplier * hidden)
        return self.multiple_of * ((hidden + self.multiple_of - 1) // self.multiple_of)


CHAMELEON_7B = ChameleonConfig()


def chameleon_shapes(cfg: ChameleonConfig) -> Dict[str, Tuple[int, ...]]:
    """Key names of the consolidated checkpoint after the wq/wk/wv -> wqkv and w1/w3 -> w13 load hooks
    (transformer.py:84-98, 197-208)."""
    hd, D, F = cfg.head_dim, cfg.dim, cfg.ffn_hidden
    s: Dict[str, Tuple[int, ...]] = {"tok_embeddings.weight": (cfg.vocab_size, D), "norm.weight": (D,),
                                    "output.weight": (cfg.vocab_size, D)}
    for l in range(cfg.n_layers):
        p = f"layers.{l}."
        s[p + "attention.wqkv.weight"] = ((cfg.n_heads + 2 * cfg.n_kv_heads) * hd, D)
        s[p + "attention.wo.weight"] = (D, cfg.n_heads * hd)
        if cfg.qk_normalization:
            for n in ("q_normalization", "k_normalization"):
                s[p + f"attention.{n}.weight"] = (hd,)
                s[p + f"attention.{n}.bias"] = (hd,)
        s[p + "feed_forward.w13.weight"] = (2 * F, D)
        s[p + "feed_forward.w2.weight"] = (D, F)
        s[p + "attention_norm.weight"] = (D,)
        s[p + "ffn_norm.weight"] = (D,)
    return s


def synth_chameleon_state(cfg: ChameleonConfig, seed: int = 0, device="cpu", logit_scale: float = 1.0, gen_device="cpu",
                          dtype=torch.bfloat16):
    """Random checkpoint in the reference's storage dtype (bf16)."""
    g = torch.Generator(device=gen_device)
    g.manual_seed(seed + 7007)
    out = {}
    for k, shp in chameleon_shapes(cfg).items():
        if k.endswith("norm.weight") or k.endswith("normalization.weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g, device=gen_device)
        elif k.endswith("normalization.bias"):
            t = 0.05 * torch.randn(shp, generator=g, device=gen_device)
        elif k == "tok_embeddings.weight":
            t = torch.randn(shp, generator=g, device=gen_device, dtype=torch.bfloat16 if len(shp) == 2 and shp[0] * shp[1] > 1 << 26 else torch.float32)
        elif k == "output.weight":
            t = torch.randn(shp, generator=g, device=gen_device, dtype=torch.bfloat16 if shp[0] * shp[1] > 1 << 26 else torch.float32) * (logit_scale / shp[1] ** 0.5)
        else:
            big = shp[0] * shp[1] > 1 << 26
            t = torch.randn(shp, generator=g, device=gen_device, dtype=torch.bfloat16 if big else torch.float32) * (1.0 / shp[1] ** 0.5)
        out[k] = t.to(dtype).to(device)
    return out


CHAMELEON_VQ = VQConfig(ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(), resolution=512, z_channels=256,
                        embed_dim=256, n_embed=8192)   # assets/chameleon_patched_config.yaml


def synth_chameleon_vocab(n_vocab: int = 65536, n_img: int = 8192, first_img: int = 4) -> Dict[str, int]:
    """A Chameleon-style tokenizer vocabulary (name -> id): specials, ``IMGIMG<letters>Z`` image tokens whose letters A..J
    spell the VQGAN code (vocab.py:82-93), filler text tokens."""
    vm = {"<s>": 0, "<pad>": 1, "</s>": 2, "<reserved08706>": 3}
    nxt = first_img
    for code in range(n_img):
        vm["IMGIMG" + "".join("ABCDEFGHIJ"[int(c)] for c in str(code)) + "Z"] = nxt
        nxt += 1
    vm["<racm3:break>"] = nxt
    vm["<eoss>"] = nxt + 1
    nxt += 2
    i = 0
    while nxt < n_vocab:
        vm[f"t{i}"] = nxt
        nxt += 1
        i += 1
    return vm


def synth_delta(state: Dict[str, torch.Tensor], prefix: str, seed: int = 0, scale: float = 0.02,
                every: int = 3) -> Dict[str, torch.Tensor]:
    """A ``*_delta.pth``-style patch (wmar/utils/utils.py:47-66) for the sub-module `prefix` ("encoder." / "decoder."):
    seeded perturbations of every `every`-th tensor, keyed RELATIVE to the sub-module like the released delta checkpoints."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 31337)
    out = {}
    keys = sorted(k for k in state if k.startswith(prefix))
    for i, k in enumerate(keys):
        if i % every:
            continue
        t = state[k]
        out[k[len(prefix):]] = (torch.randn(t.shape, generator=g) * scale * float(t.abs().mean().clamp_min(1e-3))).to(torch.float32)
    return out
