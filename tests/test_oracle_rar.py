"""RAR oracle (oracle/rar_oracle.py) against vectors produced by running the reference's RAR
generator and MaskGIT-VQGAN tokenizer (tests/golden/make_golden.py::rar_vectors)."""
import os

import numpy as np
import pytest
import torch

from oracle import rar_oracle as R
from oracle import wm_oracle as W
from tests.conftest import REPO
from wmar_amd.utils import synth

RCFG = synth.RARConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                       image_seq_len=16, codebook_size=1024, condition_num_classes=1000)
MCFG = synth.MaskgitVQConfig(hidden_channels=32, channel_mult=(1, 2, 2), num_res_blocks=1, resolution=32, z_channels=16,
                             num_embeddings=256)


@pytest.fixture(scope="module")
def rv():
    return np.load(os.path.join(REPO, "tests", "golden", "rar_vectors.npz"))


@pytest.fixture(scope="module")
def rar_key(kat, key_factory):
    return key_factory(kat["keys"]["rar"])


def test_rar_logits(rv):
    sd = synth.synth_rar_state(RCFG, seed=2, logit_scale=30.0)
    rec = []
    torch.manual_seed(21)
    R.generate(sd, RCFG, torch.from_numpy(rv["rar_cond"]), 4.0, 0.0, 1.0, record=rec)
    # the unwatermarked run shares its first step with the recorded (watermarked) logits
    got = np.concatenate([rec[0]["cond_logits"], rec[0]["uncond_logits"]])
    np.testing.assert_allclose(got, rv["rar_logits"][0], rtol=0, atol=3e-4)


@pytest.mark.parametrize("tag,gs,gp,T,wm", [("wm", 4.0, 0.0, 1.0, True), ("nowm", 4.0, 0.0, 1.0, False),
                                            ("pow", 3.0, 1.5, 0.9, True)])
def test_rar_generate_tokens(rv, rar_key, tag, gs, gp, T, wm):
    sd = synth.synth_rar_state(RCFG, seed=2, logit_scale=30.0)
    torch.manual_seed(21)
    toks = R.generate(sd, RCFG, torch.from_numpy(rv["rar_cond"]), gs, gp, T, rar_key if wm else None, 2.0)
    assert np.array_equal(toks.numpy(), rv[f"rar_tokens_{tag}"])


def test_rar_sampling_stage_on_reference_logits(rv, rar_key):
    """CFG mix + watermark + sampling on the reference's own logits and noise -> its tokens."""
    toks = rv["rar_tokens_wm"]
    B = toks.shape[0]
    scales = R.cfg_scales(16, 4.0, 0.0)
    for n in range(rv["rar_logits"].shape[0]):
        cl, ul = torch.from_numpy(rv["rar_logits"][n][:B]), torch.from_numpy(rv["rar_logits"][n][B:])
        mixed = (ul + (cl - ul) * scales[n]).numpy()
        lg = W.process_logits(rar_key, toks[:, :n], mixed, 2.0)
        got = W.sample_rows(lg, rv["rar_q"][n], 1.0, None, None)
        assert got.tolist() == toks[:, n].tolist(), n


def test_rar_first_token_never_watermarked(rar_key):
    lg = np.random.RandomState(0).randn(2, 1024).astype(np.float32)
    assert np.array_equal(W.process_logits(rar_key, np.zeros((2, 0), dtype=np.int64), lg, 2.0), lg)


def test_rar_detector(rv, rar_key):
    pv, _, _ = W.detect(rar_key, rv["rar_tokens_wm"])
    assert np.all(np.abs(np.log10(pv) - np.log10(rv["rar_pvals_wm"])) < 1e-9)


def test_maskgit_tokenizer(rv):
    sd = synth.synth_maskgit_state(MCFG, seed=4)
    img = R.maskgit_decode(sd, MCFG, torch.from_numpy(rv["mg_codes"]))
    np.testing.assert_allclose(img.numpy(), rv["mg_images"], rtol=0, atol=5e-5)
    z = R.maskgit_prequant(sd, MCFG, torch.from_numpy(rv["mg_images"]))
    np.testing.assert_allclose(z.numpy(), rv["mg_prequant"], rtol=0, atol=5e-5)
    codes = R.maskgit_encode(sd, MCFG, torch.from_numpy(rv["mg_images"]))
    assert np.array_equal(codes.numpy(), rv["mg_codes_roundtrip"])
