"""The wrappers' ``modelpath`` constructors on synthetic checkpoint directories laid out like the reference's downloads
(README / wmar/models/*_wrapper.py): key prefixes, config parsing, tokenizer json, alive-id assets, delta checkpoints.
No real weights exist offline; this pins the file formats and key names end to end."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from wmar_amd.utils import synth  # noqa: E402


def test_taming_modelpath_and_delta(tmp_path):
    from wmar_amd.models.taming_wrapper import TamingARMMWrapper
    from wmar_amd.utils.utils import update_weights
    g = synth.GPTConfig(vocab_size=256, block_size=64, n_layer=1, n_head=2, n_embd=64)
    v = synth.VQConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=16, z_channels=32, embed_dim=32, n_embed=256)
    (tmp_path / "configs").mkdir()
    (tmp_path / "checkpoints").mkdir()
    cfg = {"model": {"params": {
        "transformer_config": {"params": dict(vocab_size=g.vocab_size, block_size=g.block_size, n_layer=g.n_layer, n_head=g.n_head, n_embd=g.n_embd)},
        "first_stage_config": {"params": {"embed_dim": v.embed_dim, "n_embed": v.n_embed, "ddconfig": dict(
            ch=v.ch, ch_mult=list(v.ch_mult), num_res_blocks=v.num_res_blocks, attn_resolutions=list(v.attn_resolutions),
            resolution=v.resolution, in_channels=3, out_ch=3, z_channels=v.z_channels)}}}}}
    yaml.safe_dump(cfg, open(tmp_path / "configs" / "net2net.yaml", "w"))
    gs, vs = synth.synth_gpt_state(g, 1, "cpu", 8.0), synth.synth_vq_state(v, 1, "cpu")
    sd = {"transformer." + k: t for k, t in gs.items()}
    sd.update({"first_stage_model." + k: t for k, t in vs.items()})
    sd["first_stage_model.loss.discriminator.weight"] = torch.zeros(3)      # present in real checkpoints, ignored
    torch.save({"state_dict": sd, "global_step": 7}, tmp_path / "checkpoints" / "net2net.ckpt")
    a = TamingARMMWrapper(str(tmp_path), max_batch=4)
    b = TamingARMMWrapper(None, gpt_cfg=g, vq_cfg=v, gpt_state=gs, vq_state=vs, max_batch=4)
    torch.manual_seed(0)
    q = a.draw_noise(a.codes_size ** 2, 3)
    gp = {"temperature": 1.0, "top_k": 50, "top_p": 0.9}
    ca = a.sample([1, 2, 3], gp, q=q)
    assert torch.equal(ca, b.sample([1, 2, 3], gp, q=q))
    ia = a.codes_to_images(ca)
    assert torch.equal(ia, b.codes_to_images(ca))
    # decoder delta checkpoint (wmar/utils/utils.py:47-66): tensors are ADDED key-wise
    delta = {"conv_out.bias": 0.01 * torch.ones_like(vs["decoder.conv_out.bias"])}
    torch.save(delta, tmp_path / "dec_delta.pth")
    update_weights(a, "decoder", str(tmp_path / "dec_delta.pth"))
    ib = a.codes_to_images(ca)
    inner = ia.abs() < 0.98                       # away from the [-1, 1] clamp the output moves by exactly the bias delta
    assert float(((ib - ia)[inner] - 0.01).abs().max()) < 1e-5


def test_rar_modelpath(tmp_path):
    from wmar_amd.models.rar_wrapper import RarARMMWrapper
    import wmar_amd.models.rar_wrapper as rw
    rcfg = synth.RARConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, image_seq_len=256,
                           codebook_size=1024, condition_num_classes=1000)
    rs, vs = synth.synth_rar_state(rcfg, 3), synth.synth_maskgit_state(synth.MASKGIT_VQ, 3)
    torch.save(rs, tmp_path / "rar_t.bin")
    torch.save(vs, tmp_path / "maskgit-vqgan-imagenet-f16-256.bin")
    rw._RAR_SIZES["rar_t"] = (128, 2, 512)
    try:
        saved = rw.RARConfig
        a = None
        # the size table fixes 16 heads for the released models; the test size uses 4
        rw.RARConfig = lambda **kw: saved(**{**kw, "num_attention_heads": 4})
        a = RarARMMWrapper(str(tmp_path), "rar_t", max_batch=2)
    finally:
        rw.RARConfig = saved
        rw._RAR_SIZES.pop("rar_t")
    b = RarARMMWrapper(None, rar_cfg=rcfg, vq_cfg=synth.MASKGIT_VQ, rar_state=rs, vq_state=vs, max_batch=2)
    torch.manual_seed(0)
    q = a.draw_noise(2)
    ca = a.sample([5, 900], None, q=q)
    assert ca.shape == (2, 256) and torch.equal(ca, b.sample([5, 900], None, q=q))
    assert a.get_vq().alive_ids.numel() == 1024 and a.get_vq().dead_ids.numel() == 0      # assets/rar_all_ids.txt
    assert torch.equal(a.codes_to_images(ca), b.codes_to_images(ca))


def test_chameleon_modelpath_with_text_tokenizer(tmp_path):
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper
    import wmar_amd.models.chameleon_wrapper as cw
    cfg = synth.ChameleonConfig(dim=256, n_layers=2, n_heads=4, n_kv_heads=4, vocab_size=2048, multiple_of=64)
    vq = synth.VQConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(), resolution=16, z_channels=32, embed_dim=32, n_embed=512)
    vm = synth.synth_chameleon_vocab(2048, 512)
    words = {"a": "t0", "red": "t1", "cat": "t2", "on": "t3", "mat": "t4"}
    for w, t in words.items():                       # give a few text tokens real names
        vm[w] = vm.pop(t)
    vm["<unk>"] = vm.pop("t5")
    (tmp_path / "models" / "7b").mkdir(parents=True)
    (tmp_path / "tokenizer").mkdir()
    tok = Tokenizer(WordLevel(vm, unk_token="<unk>"))
    tok.pre_tokenizer = Whitespace()
    tok.save(str(tmp_path / "tokenizer" / "text_tokenizer.json"))
    sd = synth.synth_chameleon_state(cfg, 2, logit_scale=6.0)
    # released checkpoints keep wq / wk / wv and w1 / w3 apart (the reference fuses them in load hooks) and carry rope.freqs
    split = {}
    for k, t in sd.items():
        if k.endswith("attention.wqkv.weight"):
            q_, k_, v_ = t.chunk(3)
            p = k[: -len("wqkv.weight")]
            split[p + "wq.weight"], split[p + "wk.weight"], split[p + "wv.weight"] = q_.clone(), k_.clone(), v_.clone()
        elif k.endswith("feed_forward.w13.weight"):
            w1, w3 = t.chunk(2)
            p = k[: -len("w13.weight")]
            split[p + "w1.weight"], split[p + "w3.weight"] = w1.clone(), w3.clone()
        else:
            split[k] = t
    split["rope.freqs"] = torch.zeros(cfg.head_dim // 2)
    torch.save(split, tmp_path / "models" / "7b" / "consolidated.pth")
    json.dump({"model": dict(dim=cfg.dim, n_layers=cfg.n_layers, n_heads=cfg.n_heads, n_kv_heads=cfg.n_kv_heads, vocab_size=cfg.vocab_size,
                             ffn_dim_multiplier=1.0, multiple_of=64, norm_eps=1e-5, rope_theta=10000.0, qk_normalization=True,
                             swin_norm=False)}, open(tmp_path / "models" / "7b" / "params.json", "w"))
    json.dump({"world_size": 1}, open(tmp_path / "models" / "7b" / "consolidate_params.json", "w"))
    vs = synth.synth_vq_state(vq, 2)
    torch.save({"state_dict": vs}, tmp_path / "tokenizer" / "vqgan_patched.ckpt")
    saved = cw.CHAMELEON_VQ
    cw.CHAMELEON_VQ = vq
    try:
        a = ChameleonARMMWrapper(str(tmp_path), 0, max_batch=2, max_prompt_len=16)
    finally:
        cw.CHAMELEON_VQ = saved
    b = ChameleonARMMWrapper(None, 0, cfg=cfg, state=sd, vocab_map=vm, vq_cfg=vq, vq_state=vs, max_batch=2, max_prompt_len=16)
    assert a.get_total_vocab_size() == 2048 and a.n_image_tokens == 64
    prompts = [(0, "a red cat"), (1, "cat on a mat")]
    ids = [(0, [vm["a"], vm["red"], vm["cat"]]), (1, [vm["cat"], vm["on"], vm["a"], vm["mat"]])]
    torch.manual_seed(1)
    q = a.draw_noise(2)
    gp = {"temperature": 0.8, "top_p": 0.9}
    ca = a.sample(prompts, gp, q=q)                  # strings through the text tokenizer
    assert torch.equal(ca, b.sample(ids, gp, q=q))   # == explicit ids through the in-memory constructor
    assert torch.equal(a.codes_to_images(ca), b.codes_to_images(ca))
