"""Gumbel-key sampler / score kernels against the reference's own outputs
(tests/golden/gumbel_vectors.npz, made by importing wmar_audio/watermark/engine.py) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import rar_oracle as R  # noqa: E402
from oracle import wm_oracle as W  # noqa: E402
from tests.conftest import REPO  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402


@pytest.fixture(scope="module")
def gv():
    return np.load(os.path.join(REPO, "tests", "golden", "gumbel_vectors.npz"))


def test_key_build_matches_torch_rand(gv):
    from wmar_amd.watermarking.gumbel_watermark import key_for
    rs, lr, sc = key_for(42, 1024, "cuda")
    assert np.array_equal(rs.cpu().numpy(), gv["gum_rs_42"])
    np.testing.assert_allclose(lr.cpu().numpy(), np.log(gv["gum_rs_42"].astype(np.float64)), rtol=1e-6)


@pytest.mark.parametrize("hname", ["same", "rows"])
def test_gumbel_sample_reference_tokens(gv, hname):
    from wmar_amd.watermarking.gumbel_watermark import gumbel_sample
    lg = torch.from_numpy(gv["gum_logits"]).cuda()
    h = torch.from_numpy(gv["gum_hash_" + hname])
    for (t, p, k), name in zip(gv["gum_cases"], gv["gum_case_names"]):
        tok = gumbel_sample(lg, h, use_sampling=True, temp=float(t), top_p=float(p), top_k=int(k))
        assert np.array_equal(tok.cpu().numpy(), gv[f"gum_tok_{hname}_{name}"]), name
    tok = gumbel_sample(lg, h, use_sampling=False)
    assert np.array_equal(tok.cpu().numpy(), gv[f"gum_tok_{hname}_greedy"])


@pytest.mark.parametrize("V,B", [(1024, 64), (16384, 5), (1000, 3), (37, 2)])
def test_gumbel_sample_vs_oracle(V, B):
    from wmar_amd.watermarking.gumbel_watermark import gumbel_sample
    rs = np.random.RandomState(V + B)
    lg = (rs.randn(B, V) * 4).astype(np.float32)
    lg[0, : V // 2] = lg[0, 0]          # a block of exactly equal logits (sort ties)
    h = torch.from_numpy(rs.randint(0, 2 ** 31, size=B).astype(np.int64))
    for temp, top_p, top_k in [(1.0, 0.0, 0), (0.6, 0.0, 0), (1.0, 0.3, 0), (1.3, 0.95, 0), (1.0, 1.0, 0), (1.0, 0.0, 7),
                               (0.9, 0.0, 10 ** 6)]:
        got = gumbel_sample(torch.from_numpy(lg).cuda(), h, True, temp, top_p, top_k).cpu().numpy()
        ref = W.gumbel_sample(lg, h.numpy(), True, temp, top_p, top_k)
        assert np.array_equal(got, ref), (temp, top_p, top_k)


@pytest.mark.parametrize("hname", ["same", "rows"])
def test_gumbel_score_reference(gv, hname):
    from wmar_amd.watermarking.gumbel_watermark import gumbel_score_tok
    toks = torch.from_numpy(gv["gum_score_tokens"]).cuda()
    got = gumbel_score_tok(toks, torch.from_numpy(gv["gum_hash_" + hname]), 1024)
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), gv["gum_score_" + hname])


def test_rar_generate_gumbel_vs_oracle_and_detect():
    """RAR + Gumbel key inside the captured step: tokens equal a step-by-step oracle run; the detector
    separates keyed from unkeyed codes."""
    from wmar_amd.models.rar_wrapper import RarARMMWrapper
    from wmar_amd.watermarking.gumbel_watermark import GumbelWatermark
    rcfg = synth.RARConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                           image_seq_len=64, codebook_size=256, condition_num_classes=1000)
    vcfg = synth.MaskgitVQConfig(hidden_channels=32, channel_mult=(1, 2, 2), num_res_blocks=1, resolution=32, z_channels=16,
                                 num_embeddings=256)
    rsd = synth.synth_rar_state(rcfg, seed=1, logit_scale=6.0)
    m = RarARMMWrapper(None, rar_cfg=rcfg, vq_cfg=vcfg, rar_state=rsd, vq_state=synth.synth_maskgit_state(vcfg, seed=1), max_batch=4)
    wm = GumbelWatermark(256, seed=1234, temperature=1.0, device="cuda")
    m.set_watermarker(wm)
    cond = torch.tensor([3, 500, 77, 9, 12, 999])
    for graph in (True, False):
        m.use_graph = graph
        codes = m.sample(cond, None, apply_watermark=True)
        key = W.gumbel_key(1234, 256)
        ref = R.generate(rsd, rcfg, cond, 4.0, 0.0, 1.0, None, 0.0, draw_drop_mask=False,
                         sampler=lambda lg, n: torch.from_numpy(W.gumbel_sample(lg.numpy(), [1234] * lg.shape[0], True, 1.0, 0.0, 0)))
        assert torch.equal(codes.cpu(), ref)
    p_wm = wm.detect(codes)
    torch.manual_seed(0)
    plain = m.sample(cond, None, apply_watermark=False)
    p_plain = wm.detect(plain)
    assert float(p_wm.max()) < 1e-6 and float(p_plain.min()) > 1e-4
    s = wm.scores(codes).cpu().numpy()
    np.testing.assert_allclose(s[0], -np.log(1.0 - key[codes[0].cpu().numpy()].astype(np.float64)), rtol=1e-6)
