"""BASELINE.json's full-size configurations (random-init weights of the real architectures) through size-independent
properties: replay determinism, graph == eager loop, delta = 0 equals no watermark, detector counts equal a host recount from
the key table, decode -> re-encode -> decode fixed points.  Full 256-step generations at batch 64 are beyond what the oracle
finishes in seconds (0.4 s per step on the box's host cores); the oracle checks of the 48-layer / 32-layer engines at a few positions
and short loops are in tests/test_gpu_depth.py, parity at production width in tests/test_gpu_prod_shapes.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from wmar_amd.utils import synth  # noqa: E402


def _host_counts(table: np.ndarray, codes: np.ndarray, h: int = 1):
    """unique (h+1)-grams scored against the key bitmap (gentime_watermark.py:285-318), LINEAR seeding."""
    out = []
    for row in codes:
        seen, ng = set(), 0
        for i in range(h, len(row)):
            gram = tuple(row[i - h:i + 1])
            if gram in seen:
                continue
            seen.add(gram)
            r, t = int(sum(gram[:-1])), int(gram[-1])
            ng += (int(table[r, t >> 5]) >> (t & 31)) & 1
        out.append((len(seen), ng))
    return out


@pytest.fixture(scope="module")
def taming():
    from wmar_amd.models.taming_wrapper import TamingARMMWrapper
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    m = TamingARMMWrapper.synthetic(synth.TAMING_GPT, synth.TAMING_VQ, seed=0, max_batch=64)
    wm = GentimeWatermark(m.get_vq(), m.get_total_vocab_size(), SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25,
                          device="cuda")
    m.set_watermarker(wm)
    return m, wm


def test_taming_full_size_batch64(taming):
    m, wm = taming
    gp = {"temperature": 1.0, "top_k": 250, "top_p": 0.92}
    cond = [(i * 37) % 1000 for i in range(64)]
    torch.manual_seed(1)
    q = m.draw_noise(256, 64)
    a = m.sample(cond, gp, apply_watermark=True, q=q)
    assert a.shape == (64, 256) and int(a.min()) >= 0 and int(a.max()) < 16384
    assert torch.equal(a, m.sample(cond, gp, apply_watermark=True, q=q))            # replay of the cached graph is deterministic
    m.use_graph = False
    assert torch.equal(a[:8], m.sample(cond[:8], gp, apply_watermark=True, q=q[:, :8].contiguous()))   # rows are independent; eager == graph
    m.use_graph = True
    # delta = 0 is the unwatermarked model
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    wm0 = GentimeWatermark(m.get_vq(), 16384, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 0.0, 0.25, device="cuda")
    m.set_watermarker(wm0)
    b = m.sample(cond, gp, apply_watermark=True, q=q)
    c = m.sample(cond, gp, apply_watermark=False, q=q)
    m.set_watermarker(wm)
    assert torch.equal(b, c) and not torch.equal(a, b)
    # detector counts == host recount from the key table; watermarked codes are greener
    ns, ng = wm.detect_counts(a)[1:3]
    ref = _host_counts(wm.key_table_host().view(np.uint32), a.cpu().numpy())
    assert [(int(x), int(y)) for x, y in zip(ns.cpu(), ng.cpu())] == ref
    ns0, ng0 = wm.detect_counts(b)[1:3]
    assert float(ng.sum()) / float(ns.sum()) > float(ng0.sum()) / float(ns0.sum()) + 0.02
    pv = wm.detect(a)
    assert pv.dtype == torch.float64 and torch.isfinite(pv).all()
    # tokenizer: decode is deterministic; a re-encoded image re-decodes to a fixed point of encode∘decode after one more trip
    img = m.codes_to_images(a[:16])
    assert img.shape == (16, 3, 256, 256) and torch.equal(img, m.codes_to_images(a[:16]))
    c1 = m.images_to_codes(img)
    assert torch.equal(c1, m.images_to_codes(img))
    assert float(img.abs().max()) <= 1.0 + 1e-6


def test_rar_xl_full_size_guidance_and_gumbel():
    from wmar_amd.models.rar_wrapper import RarARMMWrapper
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    from wmar_amd.watermarking.gumbel_watermark import GumbelWatermark
    m = RarARMMWrapper.synthetic(max_batch=64, logit_scale=6.0)
    wm = GentimeWatermark(m.get_vq(), 1024, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
    m.set_watermarker(wm)
    cond = torch.arange(64) * 13 % 1000
    torch.manual_seed(2)
    q = m.draw_noise(64)
    a = m.sample(cond, None, apply_watermark=True, q=q)
    assert a.shape == (64, 256) and int(a.max()) < 1024
    assert torch.equal(a, m.sample(cond, None, apply_watermark=True, q=q))
    assert torch.equal(a[:5], m.sample(cond[:5], None, apply_watermark=True, q=q[:, :5].contiguous()))     # batch-size independent
    ns, ng = wm.detect_counts(a)[1:3]
    ref = _host_counts(wm.key_table_host().view(np.uint32), a.cpu().numpy())
    assert [(int(x), int(y)) for x, y in zip(ns.cpu(), ng.cpu())] == ref
    plain = m.sample(cond, None, apply_watermark=False, q=q)
    assert float(wm.detect(a).median()) < float(wm.detect(plain).median())
    img = m.codes_to_images(a[:8])
    assert img.shape == (8, 3, 256, 256) and float(img.min()) >= -1.0 - 1e-6 and float(img.max()) <= 1.0 + 1e-6
    assert m.images_to_codes(img).shape == (8, 256)
    # Gumbel key (BASELINE config "RAR-XL ... Gumbel-key watermark"): keyed codes score far above unkeyed ones
    gw = GumbelWatermark(1024, seed=7, temperature=1.0, device="cuda")
    m.set_watermarker(gw)
    g = m.sample(cond, None, apply_watermark=True)
    assert torch.equal(g, m.sample(cond, None, apply_watermark=True))               # the key replaces the noise: fully deterministic
    pg = gw.detect(g)      # guidance 4.0 sharpens the distributions: low-entropy rows carry less key signal
    assert float(pg.median()) < 1e-6 and float(pg.max()) < 5e-2 and float(gw.detect(plain).median()) > 1e-3


def test_chameleon_7b_full_size_short_run():
    """the 7B architecture at 48 sequences, 24 image tokens: replay determinism, allowed-token set, graph == eager, watermark bias."""
    from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    vq = synth.VQConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(), resolution=64, z_channels=32, embed_dim=32,
                        n_embed=8192)
    m = ChameleonARMMWrapper.synthetic(vq_cfg=vq, max_batch=16, logit_scale=8.0)
    m.n_image_tokens = 24
    m.is_codes_shaped = lambda c: True
    wm = GentimeWatermark(m.get_vq(), 65536, SeedStrategy.FIXED, SplitStrategy.RANDOM_STRATIFIED, 0, 4.0, 0.25, device="cuda")
    m.set_watermarker(wm)
    text = m.vocab.text_tokens
    cond = [(i, [text[(i * 37 + j * 11) % len(text)] for j in range(3 + i % 9)]) for i in range(16)]
    gp = {"temperature": 0.7, "top_p": 0.9}
    torch.manual_seed(5)
    q = m.draw_noise(16)
    a = m.sample(cond, gp, apply_watermark=True, q=q)
    assert a.shape == (16, 24) and set(a.flatten().tolist()) <= set(m.vocab.image_tokens)
    assert torch.equal(a, m.sample(cond, gp, apply_watermark=True, q=q))
    m.use_graph = False
    assert torch.equal(a, m.sample(cond, gp, apply_watermark=True, q=q))
    m.use_graph = True
    plain = m.sample(cond, gp, apply_watermark=False, q=q)
    ns, ng = wm.detect_counts(a)[1:3]
    ns0, ng0 = wm.detect_counts(plain)[1:3]
    assert float(ng.sum()) / float(ns.sum()) > float(ng0.sum()) / float(ns0.sum())


def test_chameleon_7b_real_workload_1024_tokens_vqgan512():
    """BASELINE configs[3] at its real size: 7B transformer, batch 16 (48 sequences), 1024 image tokens, VQGAN-512 decode +
    re-encode + detection.  Properties: the captured loop equals the eager loop on the first 64 tokens, a full replay is
    deterministic, only image tokens are sampled, detector counts equal a host recount, images are [-1,1] 512x512."""
    import time
    from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    m = ChameleonARMMWrapper.synthetic(max_batch=16, logit_scale=8.0)
    assert m.n_image_tokens == 1024 and m.image_size == 512
    wm = GentimeWatermark(m.get_vq(), 65536, SeedStrategy.FIXED, SplitStrategy.RANDOM_STRATIFIED, 0, 2.0, 0.25, device="cuda")
    m.set_watermarker(wm)
    text = m.vocab.text_tokens
    cond = [(i, [text[(i * 37 + j * 11) % len(text)] for j in range(8 + i % 7)]) for i in range(16)]
    gp = {"temperature": 0.9, "top_p": 0.9}        # configs/chameleon_generate.json
    torch.manual_seed(7)
    q = m.draw_noise(16)
    # first 64 tokens: captured loop == eager loop
    full_n, shaped = m.n_image_tokens, m.is_codes_shaped
    m.n_image_tokens, m.is_codes_shaped = 64, (lambda c: True)
    q64 = q[:64].contiguous()
    g64 = m.sample(cond, gp, apply_watermark=True, q=q64)
    m.use_graph = False
    e64 = m.sample(cond, gp, apply_watermark=True, q=q64)
    m.use_graph = True
    m.n_image_tokens = full_n
    del m.is_codes_shaped
    assert torch.equal(g64, e64)
    # the whole image
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = m.sample(cond, gp, apply_watermark=True, q=q)
    torch.cuda.synchronize(); t_gen = time.perf_counter() - t0
    assert a.shape == (16, 1024) and torch.equal(a[:, :64], g64)
    assert set(a.flatten().tolist()) <= set(m.vocab.image_tokens)
    assert torch.equal(a, m.sample(cond, gp, apply_watermark=True, q=q))
    pv, ns, ng = wm.detect_counts(a)
    ref = _host_counts(wm.key_table_host().view(np.uint32), a.cpu().numpy(), h=0)
    assert [(int(x), int(y)) for x, y in zip(ns.cpu(), ng.cpu())] == ref
    plain = m.sample(cond, gp, apply_watermark=False, q=q)
    ns0, ng0 = wm.detect_counts(plain)[1:3]
    assert float(ng.sum()) / float(ns.sum()) > float(ng0.sum()) / float(ns0.sum()) + 0.02
    assert float(pv.median()) < float(wm.detect(plain).median())
    img = m.codes_to_images(a)
    assert img.shape == (16, 3, 512, 512) and float(img.abs().max()) <= 1.0 + 1e-6 and torch.isfinite(img).all()
    c2 = m.images_to_codes(img)
    assert c2.shape == (16, 1024) and set(c2.flatten().tolist()) <= set(m.vocab.image_tokens)
    assert torch.equal(c2, m.images_to_codes(img))
    print(f"chameleon-7b b16: 1024 tokens in {t_gen:.2f} s ({t_gen / 1024 * 1e3:.2f} ms/step)")
