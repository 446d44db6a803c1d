"""Cross-check of the Chameleon oracle's STRUCTURE against an independent public implementation of the same model: the
Hugging Face ``transformers`` port (modeling_chameleon).  The reference's own transformer (xformers) cannot run offline;
the HF port was validated against Meta's weights upstream, so agreement on identical random weights pins norm placement, qk
LayerNorm, the rotary pairing convention, SwiGLU, residuals and the head.  fp32 on both sides (rounding is switched off in the
oracle).  CPU only.

Weight-layout conversion (what the port's conversion script does): Meta pairs rotary dimensions (2i, 2i+1), HF pairs
(i, i + hd/2); wq / wk rows and the q/k-norm parameters are permuted per head accordingly."""
import numpy as np
import pytest
import torch

from oracle import cham_oracle as CO
from wmar_amd.utils import synth

mc = pytest.importorskip("transformers.models.chameleon.modeling_chameleon")


def _perm(hd):
    """HF index j  <-  Meta index: first half takes the even (real) parts, second half the odd ones."""
    return torch.cat([torch.arange(0, hd, 2), torch.arange(1, hd, 2)])


def _build_hf(cfg, sd):
    hc = mc.ChameleonConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.dim, intermediate_size=cfg.ffn_hidden,
                            num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads,
                            rms_norm_eps=cfg.norm_eps, rope_theta=cfg.rope_theta, swin_norm=False, max_position_embeddings=64,
                            attention_dropout=0.0, attention_bias=False, mlp_bias=False, hidden_act="silu")
    hc._attn_implementation = "eager"
    H, Hkv, hd, D = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.dim
    P = _perm(hd)
    layers = []
    for l in range(cfg.n_layers):
        layer = mc.ChameleonDecoderLayer(hc, l).float().eval()
        p = f"layers.{l}."
        wqkv = sd[p + "attention.wqkv.weight"].float()
        wq, wk, wv = wqkv[: H * hd], wqkv[H * hd: (H + Hkv) * hd], wqkv[(H + Hkv) * hd:]
        with torch.no_grad():
            layer.self_attn.q_proj.weight.copy_(wq.view(H, hd, D)[:, P].reshape(H * hd, D))
            layer.self_attn.k_proj.weight.copy_(wk.view(Hkv, hd, D)[:, P].reshape(Hkv * hd, D))
            layer.self_attn.v_proj.weight.copy_(wv)
            layer.self_attn.o_proj.weight.copy_(sd[p + "attention.wo.weight"].float())
            layer.self_attn.q_norm.weight.copy_(sd[p + "attention.q_normalization.weight"].float()[P].expand(H, hd))
            layer.self_attn.q_norm.bias.copy_(sd[p + "attention.q_normalization.bias"].float()[P].expand(H, hd))
            layer.self_attn.k_norm.weight.copy_(sd[p + "attention.k_normalization.weight"].float()[P].expand(Hkv, hd))
            layer.self_attn.k_norm.bias.copy_(sd[p + "attention.k_normalization.bias"].float()[P].expand(Hkv, hd))
            w13 = sd[p + "feed_forward.w13.weight"].float()
            layer.mlp.gate_proj.weight.copy_(w13[: cfg.ffn_hidden])
            layer.mlp.up_proj.weight.copy_(w13[cfg.ffn_hidden:])
            layer.mlp.down_proj.weight.copy_(sd[p + "feed_forward.w2.weight"].float())
            layer.input_layernorm.weight.copy_(sd[p + "attention_norm.weight"].float())
            layer.post_attention_layernorm.weight.copy_(sd[p + "ffn_norm.weight"].float())
        layers.append(layer)
    rot = mc.ChameleonRotaryEmbedding(hc)
    fnorm = mc.ChameleonRMSNorm(D, eps=cfg.norm_eps).float()
    with torch.no_grad():
        fnorm.weight.copy_(sd["norm.weight"].float())
    return layers, rot, fnorm


@pytest.mark.parametrize("kv", [4, 2])
def test_oracle_structure_equals_hf_port(kv):
    cfg = synth.ChameleonConfig(dim=256, n_layers=2, n_heads=4, n_kv_heads=kv, vocab_size=512, multiple_of=64)
    sd = synth.synth_chameleon_state(cfg, seed=21, logit_scale=4.0, dtype=torch.float32)
    layers, rot, fnorm = _build_hf(cfg, sd)
    B, Tn = 3, 9
    rs = np.random.RandomState(0)
    toks = torch.from_numpy(rs.randint(0, cfg.vocab_size, size=(B, Tn)).astype(np.int64))
    # HF: whole sequences at once under a causal mask
    with torch.no_grad():
        h = sd["tok_embeddings.weight"].float()[toks]
        pos = torch.arange(Tn)[None].expand(B, Tn)
        cos_sin = rot(h, pos)
        mask = torch.full((Tn, Tn), float("-inf")).triu(1)[None, None].expand(B, 1, Tn, Tn)
        for layer in layers:
            h = layer(h, attention_mask=mask, position_ids=pos, position_embeddings=cos_sin)[0]
        ref = fnorm(h) @ sd["output.weight"].float().t()              # [B, T, V]
    # oracle: one token per row and step, KV cache, no bf16 rounding
    CO.ROUND_BF16 = False
    try:
        cache, cache_f = CO.Cache(cfg.n_layers, B), CO.Cache(cfg.n_layers, B)
        for t in range(Tn):
            lg = CO.forward_tokens(sd, cfg, toks[:, t], torch.full((B,), t), cache, fold=False)
            np.testing.assert_allclose(lg.numpy(), ref[:, t].numpy(), rtol=0, atol=2e-4)
            # the engine's algebra (RMSNorm weight folded into the next Linear, 1/rms on its output) is the same function
            lf = CO.forward_tokens(sd, cfg, toks[:, t], torch.full((B,), t), cache_f, fold=True)
            np.testing.assert_allclose(lf.numpy(), ref[:, t].numpy(), rtol=0, atol=2e-4)
    finally:
        CO.ROUND_BF16 = True
