"""HIP RAR engine (adaLN blocks, qk-norm, KV cache, guidance) against the oracle and the
reference's golden logits/tokens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import rar_oracle as R  # noqa: E402
from oracle import wm_oracle as W  # noqa: E402
from tests.conftest import REPO  # noqa: E402
from tests.test_gpu_watermark import _wm  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402

RCFG = synth.RARConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                       image_seq_len=16, codebook_size=1024, condition_num_classes=1000)


@pytest.fixture(scope="module")
def rv():
    return np.load(os.path.join(REPO, "tests", "golden", "rar_vectors.npz"))


@pytest.fixture(scope="module")
def eng():
    from wmar_amd.models.engine import RAREngine
    sd = synth.synth_rar_state(RCFG, seed=2, logit_scale=30.0)
    return RAREngine(RCFG, sd, max_batch=8), sd


def test_forward_positions_golden(rv, eng):
    """teacher-forced positions: logits equal the reference's (cond rows then uncond rows)."""
    e, _ = eng
    toks = rv["rar_tokens_wm"]
    B = toks.shape[0]
    cond = torch.from_numpy(rv["rar_cond"]) + RCFG.codebook_size + 1
    both = torch.cat([cond, torch.full_like(cond, RCFG.none_condition_id)]).cuda()
    e.forward_position(torch.full((2 * B,), -1, dtype=torch.int64).cuda(), both, 0)
    tok = both
    for n in range(rv["rar_logits"].shape[0]):
        lg = e.forward_position(tok, both, n + 1).cpu().numpy()
        np.testing.assert_allclose(lg, rv["rar_logits"][n], rtol=0, atol=5e-4)
        t = torch.from_numpy(toks[:, n])
        tok = torch.cat([t, t]).cuda()


@pytest.mark.parametrize("hd_cfg", [(160, 2, 640), (192, 4, 384)])   # head_dim 80 and 48 (padded row groups)
def test_forward_odd_head_dims_vs_oracle(hd_cfg):
    from wmar_amd.models.engine import RAREngine
    d, H, F = hd_cfg
    cfg = synth.RARConfig(hidden_size=d, num_hidden_layers=2, num_attention_heads=H, intermediate_size=F,
                          image_seq_len=16, codebook_size=256, condition_num_classes=10)
    sd = synth.synth_rar_state(cfg, seed=7, logit_scale=20.0)
    e = RAREngine(cfg, sd, max_batch=4)
    M = 6
    rs = np.random.RandomState(0)
    cond = torch.from_numpy(rs.randint(257, 267, size=M).astype(np.int64))
    ce = sd["embeddings.weight"][cond]
    cls = sd["cls_token"][0, 0].expand(M, -1)
    _, kc, vc = R.rar_position(sd, cfg, cls, ce, 0, None, None)
    e.forward_position(torch.full((M,), -1, dtype=torch.int64).cuda(), cond.cuda(), 0)
    tok = cond
    for p in range(1, 6):
        ref, kc, vc = R.rar_position(sd, cfg, sd["embeddings.weight"][tok], ce, p, kc, vc)
        lg = e.forward_position(tok.cuda(), cond.cuda(), p).cpu().numpy()
        np.testing.assert_allclose(lg, ref.numpy(), rtol=0, atol=5e-4)
        tok = torch.from_numpy(rs.randint(0, 256, size=M).astype(np.int64))


def test_forward_128_rows_wide_mlp_vs_oracle():
    """M = 128 rows (batch 64 under guidance) with an MLP wide enough that the GELU GEMM runs four row tiles per workgroup."""
    from wmar_amd.models.engine import RAREngine
    cfg = synth.RARConfig(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=4224,
                          image_seq_len=16, codebook_size=256, condition_num_classes=10)
    sd = synth.synth_rar_state(cfg, seed=11, logit_scale=20.0)
    e = RAREngine(cfg, sd, max_batch=64)
    M = 128
    rs = np.random.RandomState(3)
    cond = torch.from_numpy(rs.randint(257, 267, size=M).astype(np.int64))
    ce = sd["embeddings.weight"][cond]
    cls = sd["cls_token"][0, 0].expand(M, -1)
    _, kc, vc = R.rar_position(sd, cfg, cls, ce, 0, None, None)
    e.forward_position(torch.full((M,), -1, dtype=torch.int64).cuda(), cond.cuda(), 0)
    tok = cond
    for p in range(1, 4):
        ref, kc, vc = R.rar_position(sd, cfg, sd["embeddings.weight"][tok], ce, p, kc, vc)
        lg = e.forward_position(tok.cuda(), cond.cuda(), p).cpu().numpy()
        np.testing.assert_allclose(lg, ref.numpy(), rtol=0, atol=5e-4)
        tok = torch.from_numpy(rs.randint(0, 256, size=M).astype(np.int64))


@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("tag,gs,gp,T,use_wm", [("wm", 4.0, 0.0, 1.0, True), ("nowm", 4.0, 0.0, 1.0, False),
                                                ("pow", 3.0, 1.5, 0.9, True)])
def test_generate_reproduces_reference_tokens(rv, kat, eng, tag, gs, gp, T, use_wm, graph):
    e, _ = eng
    wm = _wm(kat["keys"]["rar"])
    torch.manual_seed(21)
    torch.rand(4, 1)   # the label-drop mask draw of preprocess_condition precedes the sampling noise (rar.py:305)
    q = torch.stack([torch.empty(4, 1024).exponential_(1) for _ in range(16)]).cuda()
    toks = e.generate(torch.from_numpy(rv["rar_cond"]).cuda(), q, R.cfg_scales(16, gs, gp), T,
                      wm.wm_ctx() if use_wm else None, use_graph=graph)
    assert np.array_equal(toks.cpu().numpy(), rv[f"rar_tokens_{tag}"])


def test_generate_without_guidance_vs_oracle(eng, kat, key_factory):
    e, sd = eng
    wm = _wm(kat["keys"]["rar"])
    key = key_factory(kat["keys"]["rar"])
    B = 5
    g = torch.Generator().manual_seed(3)
    q = torch.empty(16, B, 1024).exponential_(1, generator=g)
    cond = torch.tensor([0, 999, 17, 400, 123])
    ref = R.generate(sd, RCFG, cond, guidance_scale=0, key=key, delta=2.0, q_source=lambda n, b, v: q[n],
                     draw_drop_mask=False)
    toks = e.generate(cond.cuda(), q.cuda(), None, 1.0, wm.wm_ctx())
    assert np.array_equal(toks.cpu().numpy(), ref.numpy())
    pv = wm.detect(toks)
    rpv, _, _ = W.detect(key, ref.numpy())
    assert np.allclose(pv.cpu().numpy(), rpv, rtol=1e-9, atol=0, equal_nan=True)


MCFG = synth.MaskgitVQConfig(hidden_channels=32, channel_mult=(1, 2, 2), num_res_blocks=1, resolution=32, z_channels=16,
                             num_embeddings=256)


def test_maskgit_tokenizer_golden(rv):
    from wmar_amd.models.engine import MaskgitVQEngine
    sd = synth.synth_maskgit_state(MCFG, seed=4)
    e = MaskgitVQEngine(MCFG, sd, max_batch=4)
    img = e.decode(torch.from_numpy(rv["mg_codes"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(img, rv["mg_images"], rtol=0, atol=2e-4)
    codes, pre = e.encode(torch.from_numpy(rv["mg_images"]).cuda(), return_prequant=True)
    np.testing.assert_allclose(pre.cpu().numpy(), rv["mg_prequant"], rtol=0, atol=2e-4)
    assert np.array_equal(codes.cpu().numpy(), rv["mg_codes_roundtrip"])


def test_maskgit_mid_config_vs_oracle():
    from wmar_amd.models.engine import MaskgitVQEngine
    cfg = synth.MaskgitVQConfig(hidden_channels=32, channel_mult=(1, 1, 2, 4), num_res_blocks=2, resolution=64, z_channels=32,
                                num_embeddings=512)
    sd = synth.synth_maskgit_state(cfg, seed=9)
    e = MaskgitVQEngine(cfg, sd, max_batch=4)
    rs = np.random.RandomState(1)
    codes = torch.from_numpy(rs.randint(0, 512, size=(3, cfg.codes_size ** 2)).astype(np.int64))
    ref = R.maskgit_decode(sd, cfg, codes)
    img = e.decode(codes.cuda())
    np.testing.assert_allclose(img.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-4)
    rz = R.maskgit_prequant(sd, cfg, ref)
    got, pre = e.encode(ref.cuda(), return_prequant=True)
    np.testing.assert_allclose(pre.cpu().numpy(), rz.numpy(), rtol=0, atol=5e-4)
    rc = R.maskgit_argmin(sd["quantize.embedding.weight"], rz).view(3, -1)
    assert (got.cpu() == rc).float().mean().item() >= 0.99


def test_rar_wrapper_end_to_end(kat):
    """sample -> codes_to_images -> images_to_codes -> detect through the wrapper API."""
    from wmar_amd.models.rar_wrapper import RarARMMWrapper
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    rcfg = synth.RARConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                           image_seq_len=64, codebook_size=256, condition_num_classes=1000)
    vcfg = synth.MaskgitVQConfig(hidden_channels=32, channel_mult=(1, 2, 2), num_res_blocks=1, resolution=32, z_channels=16,
                                 num_embeddings=256)
    rsd = synth.synth_rar_state(rcfg, seed=1, logit_scale=20.0)
    vsd = synth.synth_maskgit_state(vcfg, seed=1)
    m = RarARMMWrapper(None, rar_cfg=rcfg, vq_cfg=vcfg, rar_state=rsd, vq_state=vsd, max_batch=4)
    wm = GentimeWatermark(m.get_vq(), 256, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 4.0, 0.25, device="cuda")
    m.set_watermarker(wm)
    torch.manual_seed(0)
    q = m.draw_noise(6)
    codes = m.sample([1, 2, 3, 4, 5, 6], None, apply_watermark=True, q=q)
    key = W.KeyParams(wm._alive_host, wm._dead_host, 256, 0.25)
    ref = R.generate(rsd, rcfg, torch.tensor([1, 2, 3, 4, 5, 6]), 4.0, 0.0, 1.0, key, 4.0,
                     q_source=lambda n, b, v: q[n].cpu(), draw_drop_mask=False)
    assert torch.equal(codes.cpu(), ref)
    imgs = m.codes_to_images(codes)
    np.testing.assert_allclose(imgs.cpu().numpy(), R.maskgit_decode(vsd, vcfg, ref).numpy(), rtol=0, atol=5e-4)
    c2 = m.images_to_codes(imgs)
    assert c2.shape == codes.shape
    pv = wm.detect(codes)
    assert float(pv.min()) < 1e-3     # delta=4 watermark is clearly detectable on the generated codes


_CHILD_RM = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import rar_oracle as R
from wmar_amd.utils import synth
from wmar_amd.models.engine import RAREngine
cfg = synth.RARConfig(hidden_size=1280, num_hidden_layers=2, num_attention_heads=16, intermediate_size=5120, image_seq_len=256,
                      codebook_size=1024, condition_num_classes=1000)
eng = RAREngine(cfg, synth.synth_rar_state(cfg, seed=12, logit_scale=8.0), max_batch=64)
print("STATUS0", eng.launch_status())
B = 64
q = torch.empty(256, B, 1024).exponential_(1, generator=torch.Generator().manual_seed(3)).cuda()
cond = (torch.arange(B) * 13 %% 1000).cuda()
tok = eng.generate(cond, q, R.cfg_scales(256, 4.0, 0.0), 1.0, None)
ids = cond.cpu() + cfg.codebook_size + 1
both = torch.cat([ids, torch.full_like(ids, cfg.none_condition_id)]).cuda()
lg = eng.forward_position(torch.full((2 * B,), -1, dtype=torch.int64).cuda(), both, 0)
print("STATUS1", eng.launch_status())
np.savez(sys.argv[1], tokens=tok.cpu().numpy(), logits=lg.cpu().numpy())
"""


def test_resid_mod_fallback_pair_is_bit_identical_and_recovers_a_failed_wait(tmp_path):
    """k_resid_mod as ONE launch (in-launch wait), as the two-launch pair from the start (WMAR_NO_XR=1), and with the wait flag
    raised in front of the first call (WMAR_INJECT_SYNC_FAIL=1: the call notices, switches to the pair and re-runs): the same
    tokens and logits bit for bit at RAR-XL width (2 layers x 1280, 128 rows under guidance, 256 positions)."""
    import subprocess
    import sys
    runs = {}
    for name, env in (("fused", {}), ("pair", {"WMAR_NO_XR": "1"}), ("hurt", {"WMAR_INJECT_SYNC_FAIL": "1"})):
        out = tmp_path / f"{name}.npz"
        res = subprocess.run([sys.executable, "-c", _CHILD_RM % REPO, str(out)], env=dict(os.environ, **env), capture_output=True,
                             text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        st = {l.split()[0]: eval(l.split(None, 1)[1]) for l in res.stdout.splitlines() if l.startswith("STATUS")}
        runs[name] = (np.load(out), st)
    assert runs["fused"][1]["STATUS0"] == {"fused": 1, "fallbacks": 0} == runs["fused"][1]["STATUS1"]
    assert runs["pair"][1]["STATUS1"] == {"fused": 0, "fallbacks": 0}
    assert runs["hurt"][1]["STATUS0"]["fused"] == 1 and runs["hurt"][1]["STATUS1"] == {"fused": 0, "fallbacks": 1}
    for name in ("pair", "hurt"):
        assert np.array_equal(runs[name][0]["tokens"], runs["fused"][0]["tokens"]), name
        assert np.array_equal(runs[name][0]["logits"], runs["fused"][0]["logits"]), name
