"""The CPU oracle against the production-width reference fixtures (tests/golden/prod_vectors.npz): the oracle the GPU parity
tests lean on is itself pinned at 1536 / 1280 wide, not only at the 128-wide fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import model_oracle as M
from oracle import rar_oracle as R
from tests.conftest import REPO
from wmar_amd.utils import synth

GCFG = synth.GPTConfig(vocab_size=16384, block_size=256, n_layer=2, n_head=24, n_embd=1536)
RCFG = synth.RARConfig(hidden_size=1280, num_hidden_layers=2, num_attention_heads=16, intermediate_size=5120,
                       image_seq_len=256, codebook_size=1024, condition_num_classes=1000)


@pytest.fixture(scope="module")
def pv():
    return np.load(os.path.join(REPO, "tests", "golden", "prod_vectors.npz"))


def test_gpt_oracle_logits_first_34_positions(pv):
    torch.set_num_threads(8)
    sd = synth.synth_gpt_state(GCFG, seed=9, logit_scale=10.0)
    seq = torch.from_numpy(pv["gpt_seq"].astype(np.int64))[:8]
    want = {int(p): i for i, p in enumerate(pv["gpt_pos"])}
    pk = pvv = None
    for t in range(34):
        lg, nk, nv = M.gpt_step(sd, GCFG.n_head, seq[:, t:t + 1], pk, pvv, t)
        pk = nk if pk is None else [torch.cat((a, b), -2) for a, b in zip(pk, nk)]
        pvv = nv if pvv is None else [torch.cat((a, b), -2) for a, b in zip(pvv, nv)]
        assert np.array_equal(lg.argmax(-1).numpy(), pv["gpt_argmax"][t][:8].astype(np.int64))
        if t in want:
            np.testing.assert_allclose(lg[:, ::64].numpy(), pv["gpt_logits"][want[t]][:8], rtol=0, atol=2e-4)


def test_gpt_oracle_reproduces_reference_loop_tokens(pv, kat, key_factory):
    torch.set_num_threads(8)
    sd = synth.synth_gpt_state(GCFG, seed=9, logit_scale=10.0)
    key = key_factory(kat["keys"]["taming"])
    torch.manual_seed(11)
    q = [torch.empty(4, 16384).exponential_(1) for _ in range(256)]
    toks = M.sample_with_past(sd, GCFG.n_head, torch.from_numpy(pv["loop_cond"]), 256, 1.0, 250, 0.92, key, 2.0,
                              q_source=lambda n, b, v: q[n])
    assert np.array_equal(toks.numpy(), pv["loop_tokens"].astype(np.int64))


def test_rar_oracle_logits_first_steps(pv):
    torch.set_num_threads(8)
    sd = synth.synth_rar_state(RCFG, seed=12, logit_scale=8.0)
    toks = pv["rar_tokens"].astype(np.int64)
    rows = np.r_[0:4, 64:68]                                     # 4 conditional + their 4 unconditional rows
    cond = torch.from_numpy(pv["rar_cond"].astype(np.int64)) + RCFG.codebook_size + 1
    both = torch.cat([cond, torch.full_like(cond, RCFG.none_condition_id)])[rows]
    ce = sd["embeddings.weight"][both]
    cls = sd["cls_token"][0, 0].expand(len(rows), -1)
    _, kc, vc = R.rar_position(sd, RCFG, cls, ce, 0, None, None)
    want = {int(s): i for i, s in enumerate(pv["rar_steps"])}
    tok = both
    for n in range(18):
        lg, kc, vc = R.rar_position(sd, RCFG, sd["embeddings.weight"][tok], ce, n + 1, kc, vc)
        if n in want:
            np.testing.assert_allclose(lg[:, ::8].numpy(), pv["rar_logits"][want[n]][rows], rtol=0, atol=2e-4)
        t = torch.from_numpy(toks[:4, n])
        tok = torch.cat([t, t])
