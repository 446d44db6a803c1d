"""HIP Chameleon decode engine (bf16, RMSNorm fold, qk-norm, RoPE, SwiGLU, per-row positions) against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cham_oracle as CO  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402


def _cfg(hd=64, kv=None, qk=True, dim=256, layers=2, vocab=1024):
    H = dim // hd
    return synth.ChameleonConfig(dim=dim, n_layers=layers, n_heads=H, n_kv_heads=kv or H, vocab_size=vocab, multiple_of=64,
                                 qk_normalization=qk)


def _close(got, ref, frac):
    """bf16 pipelines agree to a fraction of the logit scale (+ one bf16 ulp of the largest logit: outputs ARE bf16 values)."""
    scale = float(ref.std())
    err = float((got - ref).abs().max())
    assert err <= frac * scale + 2.0 ** -7 * float(ref.abs().max()), (err, scale)
    assert float((got - ref).abs().mean()) <= 0.25 * frac * scale


@pytest.mark.parametrize("hd,kv,qk,M", [(64, None, True, 5), (128, None, True, 48), (64, 2, False, 33), (128, 1, True, 96), (64, 4, True, 120)])
def test_forward_tokens_vs_oracle(hd, kv, qk, M):
    from wmar_amd.models.engine import ChameleonEngine
    cfg = _cfg(hd=hd, kv=kv, qk=qk, dim=256 if hd == 64 else 512)
    sd = synth.synth_chameleon_state(cfg, seed=3, logit_scale=4.0)
    e = ChameleonEngine(cfg, sd, max_batch=(M + 2) // 3, max_seq_len=32)
    rs = np.random.RandomState(M)
    cache_f, cache_r = CO.Cache(cfg.n_layers, M), CO.Cache(cfg.n_layers, M)
    start = torch.from_numpy(rs.randint(0, 3, size=M).astype(np.int32)) * 0     # every row starts at position 0
    for step in range(6):
        tok = torch.from_numpy(rs.randint(0, cfg.vocab_size, size=M).astype(np.int64))
        pos = start + step
        got = e.forward_tokens(tok.cuda(), pos.cuda()).cpu()
        ref_f = CO.forward_tokens(sd, cfg, tok, pos, cache_f, fold=True)
        ref_r = CO.forward_tokens(sd, cfg, tok, pos, cache_r, fold=False)
        _close(got, ref_f, 0.03)     # same algebra as the engine: differences are summation order + rare bf16 flips
        _close(got, ref_r, 0.08)     # the reference's rounding points


def test_ragged_positions_and_cache_rewrite():
    """rows at different positions in one step; a row restarted at position 0 ignores its stale cache."""
    from wmar_amd.models.engine import ChameleonEngine
    cfg = _cfg(hd=64, dim=256)
    sd = synth.synth_chameleon_state(cfg, seed=5, logit_scale=4.0)
    e = ChameleonEngine(cfg, sd, max_batch=2, max_seq_len=16)
    prompts = [[5, 6, 7, 8, 9], [11, 12], [3], [1, 2, 3, 4], [9, 9, 9], [7, 700]]
    M = len(prompts)
    maxlen = max(len(p) for p in prompts)
    cache = CO.Cache(cfg.n_layers, M)
    ref, nxt = CO.prefill_right_aligned(sd, cfg, prompts, cache, fold=True)
    got = None
    for j in range(maxlen):
        tok, pos = [], []
        for p in prompts:
            i = j - (maxlen - len(p))
            tok.append(p[i] if i >= 0 else 0)
            pos.append(max(i, 0))
        got = e.forward_tokens(torch.tensor(tok).cuda(), torch.tensor(pos, dtype=torch.int32).cuda())
    _close(got.cpu(), ref, 0.03)
    tok = torch.tensor([1, 2, 3, 4, 5, 6])
    got2 = e.forward_tokens(tok.cuda(), nxt.to(torch.int32).cuda()).cpu()
    ref2 = CO.forward_tokens(sd, cfg, tok, nxt, cache, fold=True)
    _close(got2, ref2, 0.03)


@pytest.fixture(scope="module")
def cv():
    import os
    from tests.conftest import REPO
    return np.load(os.path.join(REPO, "tests", "golden", "chameleon_vectors.npz"))


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("name,seed", [("fixed", "fixed"), ("linear", "linear"), ("nowm", None)])
def test_cham_sample_reference_tokens(cv, name, seed, compact):
    """guidance mix -> watermark -> allow-only -> temperature -> top-p -> multinomial in ONE launch reproduces the reference's tokens."""
    import ctypes as C
    from wmar_amd import _lib
    from wmar_amd.models.chameleon_wrapper import allow_bitmap
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    h, delta, temp, top_p = cv[f"cham_{name}_params"]
    lg = torch.from_numpy(cv["cham_logits3"]).cuda()
    B, V = lg.shape[0] // 3, lg.shape[1]
    alive = cv["cham_image_tokens"]
    dead = sorted(set(range(V)) - set(alive.tolist()))
    ctx = None
    if seed:
        vq = {"alive_ids": torch.tensor(alive), "dead_ids": torch.tensor(dead), "embedding": torch.zeros(V, 4)}
        wm = GentimeWatermark(vq, V, SeedStrategy(seed), SplitStrategy.RANDOM_STRATIFIED, int(h), float(delta), 0.25, device="cuda")
        ctx = wm.wm_ctx()
    past = torch.from_numpy(cv["cham_input_ids"][:B]).cuda().contiguous()
    q = torch.from_numpy(cv[f"cham_{name}_q"]).cuda()
    out = torch.empty(B, dtype=torch.int64, device="cuda")
    scratch = torch.empty(B, V, device="cuda")
    allow = allow_bitmap(alive.tolist(), V, "cuda")
    ids = torch.from_numpy(np.sort(alive).astype(np.int32)).cuda()
    _lib.check(_lib.load().wmar_cham_sample(C.byref(ctx) if ctx is not None else None, lg.data_ptr(), B, V, past.data_ptr(), past.shape[1],
                                            past.stride(0), float(temp), float(top_p), 3.0, 1.2, allow.data_ptr(),
                                            ids.data_ptr() if compact else None, ids.numel() if compact else 0, q.data_ptr(),
                                            scratch.data_ptr(), out.data_ptr(), _lib.stream_ptr()))
    assert np.array_equal(out.cpu().numpy(), cv[f"cham_{name}_tok"])


@pytest.mark.parametrize("graph,h", [(True, 1), (False, 1), (True, 2), (True, 3)])
def test_generate_image_loop(graph, h):
    """The captured generation loop: every sampled token equals the oracle's sampling chain applied to the engine's own
    step logits (same prompts, same tokens fed back), with ragged prompts, a LINEAR watermark and top-p.  The watermark context
    is the right-aligned, left-padded input row as the reference's processors see it: with h = 2 the first image token's context
    is the last two prompt tokens, and a prompt shorter than the context would reach into the padding."""
    from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    from oracle import wm_oracle as W
    cfg = _cfg(hd=64, dim=256, vocab=2048)
    vq_cfg = synth.VQConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(), resolution=16, z_channels=32, embed_dim=32,
                            n_embed=512)
    sd = synth.synth_chameleon_state(cfg, seed=8, logit_scale=6.0)
    vm = synth.synth_chameleon_vocab(2048, 512)
    m = ChameleonARMMWrapper(None, 0, cfg=cfg, state=sd, vocab_map=vm, vq_cfg=vq_cfg, vq_state=synth.synth_vq_state(vq_cfg, 1),
                             max_batch=3, max_prompt_len=16)
    assert m.n_image_tokens == 64
    m.use_graph = graph
    wm = GentimeWatermark(m.get_vq(), 2048, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, h, 3.0, 0.25, device="cuda")
    m.set_watermarker(wm)
    text = m.vocab.text_tokens
    cond = [(0, [text[5], text[9], text[100]]), (1, [text[7]]), (2, [text[1], text[2], text[3], text[4], text[400]])]
    torch.manual_seed(3)
    q = m.draw_noise(3)
    codes = m.sample(cond, {"temperature": 0.9, "top_p": 0.8}, apply_watermark=True, q=q)
    assert set(codes.flatten().tolist()) <= set(m.vocab.image_tokens)
    # replay on a second engine, step by step
    from wmar_amd.models.engine import ChameleonEngine
    e2 = ChameleonEngine(cfg, sd, max_batch=3, max_seq_len=16 + 64)
    prompts = m.split_inputs_for_cfg([m.tokens_from_ui([{"type": "ids", "value": p}, {"type": "sentinel", "value": "<END-OF-TURN>"}])
                                      for _, p in cond])
    M, maxlen = len(prompts), max(len(p) for p in prompts)
    lg = None
    for j in range(maxlen):
        tok = [p[j - (maxlen - len(p))] if j - (maxlen - len(p)) >= 0 else 0 for p in prompts]
        pos = [max(j - (maxlen - len(p)), 0) for p in prompts]
        lg = e2.forward_tokens(torch.tensor(tok).cuda(), torch.tensor(pos, dtype=torch.int32).cuda())
    key = W.KeyParams(wm._alive_host, wm._dead_host, 2048, 0.25, seed="linear", context_size=h)
    padded = [[m.vocab.pad_id] * (maxlen - len(p)) + list(p) for p in prompts[:3]]        # AlignPromptRight, alignment.py:27-41
    past = np.array(padded, dtype=np.int64)
    pos = torch.tensor([len(p) for p in prompts], dtype=torch.int32)
    for n in range(64):
        tok, _ = CO.sample_step(lg.cpu(), q[n].cpu().numpy(), 0.9, 0.8, 3.0, 1.2, allow_ids=m.vocab.image_tokens, key=key, past_ids=past,
                                delta=3.0)
        assert np.array_equal(tok, codes[:, n].cpu().numpy()), n
        past = np.concatenate([past, tok[:, None]], axis=1)
        if n < 63:
            t3 = torch.from_numpy(np.concatenate([tok, tok, tok]))
            lg = e2.forward_tokens(t3.cuda(), (pos + n).cuda())
    # decode -> re-encode -> detect through the wrapper
    imgs = m.codes_to_images(codes)
    assert imgs.shape == (3, 3, 16, 16)
    c2 = m.images_to_codes(imgs)
    assert c2.shape == codes.shape and set(c2.flatten().tolist()) <= set(m.vocab.image_tokens)
    pv = wm.detect(codes)
    assert float(pv.median()) < 5e-2      # 64 tokens only: the watermark is visible, not overwhelming


def test_chameleon_vqgan_reference_vectors(cv):
    """Chameleon's own VQGAN (deps/chameleon/inference/vqgan.py, run by make_golden.py) == the VQGAN engine on the same weights."""
    from wmar_amd.models.engine import VQGANEngine
    vcfg = synth.VQConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(), resolution=32, z_channels=32, embed_dim=32,
                          n_embed=256)
    e = VQGANEngine(vcfg, synth.synth_vq_state(vcfg, seed=5), max_batch=2)
    img = e.decode(torch.from_numpy(cv["chvq_codes"]).cuda())
    np.testing.assert_allclose(img.cpu().numpy(), np.clip(cv["chvq_images"], -1, 1), rtol=0, atol=2e-4)
    codes, pre = e.encode(torch.from_numpy(cv["chvq_images"]).cuda(), return_prequant=True)
    np.testing.assert_allclose(pre.cpu().numpy(), cv["chvq_prequant"], rtol=0, atol=3e-4)
    assert (codes.cpu().numpy() == cv["chvq_codes_roundtrip"]).mean() >= 0.99
