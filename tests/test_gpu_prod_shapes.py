"""The kernel variants the benchmark actually runs, against the REFERENCE's outputs at production width
(tests/golden/prod_vectors.npz, made by tests/golden/make_golden.py prod_vectors from mingpt.py:125-214 / rar.py:319-459).

Taming GPT at 2 layers x 1536 x 24 heads, block 256, 64 rows: the 1536-wide skinny GEMMs incl. FC2's non-uniform split-K
(n_hi), the two-launch vocabulary head, 6 LayerNorm-statistics chunks, and -- by walking all 256 positions -- k_attn_decode with
1 / 2 / 4 waves per (sequence, head) (cache <= 112 / <= 208 / longer) and its multi-chunk double-buffered loop.
RAR at 2 layers x 1280 x 16 heads (head_dim 80 on padded lane groups), 128 rows (64 conditions under guidance), 256 steps."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import rar_oracle as R  # noqa: E402
from tests.conftest import REPO  # noqa: E402
from tests.test_gpu_watermark import _wm  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402

GCFG = synth.GPTConfig(vocab_size=16384, block_size=256, n_layer=2, n_head=24, n_embd=1536)
RCFG = synth.RARConfig(hidden_size=1280, num_hidden_layers=2, num_attention_heads=16, intermediate_size=5120,
                       image_seq_len=256, codebook_size=1024, condition_num_classes=1000)
ATOL = 5e-4


@pytest.fixture(scope="module")
def pv():
    return np.load(os.path.join(REPO, "tests", "golden", "prod_vectors.npz"))


@pytest.fixture(scope="module")
def gpt():
    from wmar_amd.models.engine import GPTEngine
    return GPTEngine(GCFG, synth.synth_gpt_state(GCFG, seed=9, logit_scale=10.0), max_batch=64)


@pytest.mark.parametrize("phases", [None, (112, 208), (0, 0)])
def test_gpt_1536_teacher_forced_256_positions(pv, gpt, phases):
    """logits at positions 0,1,31,32,33,111,112,113,207,208,209,255 (every 64th logit) within 5e-4 of the reference's, the
    arg-max token of all 64 rows at ALL 256 positions equal.  phases: the default (one attention wave per (sequence, head) at
    every cache length), the 1 / 2 / 4-wave schedule switching at 112 / 208 rows, and four waves throughout."""
    if phases is not None:
        gpt.set_attention_phases(*phases)
    try:
        _teacher_forced(pv, gpt)
    finally:
        gpt.set_attention_phases(1 << 30, 1 << 30)


def _teacher_forced(pv, gpt):
    seq = torch.from_numpy(pv["gpt_seq"].astype(np.int64)).cuda()
    want = {int(p): i for i, p in enumerate(pv["gpt_pos"])}
    worst, flips = 0.0, 0
    for t in range(256):
        lg = gpt.decode_step(seq[:, t], t)
        am = lg.argmax(-1).cpu().numpy()
        ref_am = pv["gpt_argmax"][t].astype(np.int64)
        if not np.array_equal(am, ref_am):      # an arg-max may only move between two logits closer than the tolerance
            l = lg.cpu().numpy()
            for b in np.nonzero(am != ref_am)[0]:
                assert abs(l[b, am[b]] - l[b, ref_am[b]]) < 2 * ATOL, (t, b)
                flips += 1
        if t in want:
            d = np.abs(lg[:, ::64].cpu().numpy() - pv["gpt_logits"][want[t]]).max()
            worst = max(worst, float(d))
            assert d < ATOL, (t, d)
    assert flips <= 2
    print(f"max |dlogit| over {len(want)} positions x 64 rows: {worst:.2e}; arg-max near-tie flips: {flips}")


@pytest.mark.parametrize("graph,phases", [(True, None), (False, None), (True, (112, 208))])
def test_gpt_1536_watermarked_loop_256_steps(pv, kat, gpt, graph, phases):
    """sample_with_past, 256 steps, 4 rows, greenlist watermark + top-k 250 + top-p 0.92 on the reference's noise: the three
    captured step graphs (1 / 2 / 4 attention waves) reproduce the reference's token ids."""
    wm = _wm(kat["keys"]["taming"])
    torch.manual_seed(11)
    q = torch.stack([torch.empty(4, 16384).exponential_(1) for _ in range(256)]).cuda()
    if phases is not None:
        gpt.set_attention_phases(*phases)
    try:
        toks = gpt.generate(torch.from_numpy(pv["loop_cond"]).view(-1).cuda(), 256, q, 1.0, 250, 0.92, wm.wm_ctx(), use_graph=graph)
    finally:
        gpt.set_attention_phases(1 << 30, 1 << 30)
    ref = pv["loop_tokens"].astype(np.int64)
    got = toks.cpu().numpy()
    assert np.array_equal(got, ref), f"first mismatch per row: {[(int(np.argmax(g != r)) if (g != r).any() else -1) for g, r in zip(got, ref)]}"
    p = wm.detect(toks).cpu().numpy()
    assert np.allclose(np.log10(p), np.log10(pv["loop_pvals"]), rtol=0, atol=1e-9) and np.abs(p - pv["loop_pvals"]).max() < 1e-5


def test_gpt_1536_batch_33_and_1_vs_fixture_rows(pv, gpt):
    """rows are independent: batches of 1 and 33 (one / two row tiles, other GEMM variants) give the same logits."""
    seq = torch.from_numpy(pv["gpt_seq"].astype(np.int64)).cuda()
    for B in (1, 33):
        for t in range(3):
            lg = gpt.decode_step(seq[:B, t], t)
            if t in (0, 1):
                i = list(pv["gpt_pos"]).index(t)
                assert np.abs(lg[:, ::64].cpu().numpy() - pv["gpt_logits"][i][:B]).max() < ATOL


@pytest.fixture(scope="module")
def rar():
    from wmar_amd.models.engine import RAREngine
    return RAREngine(RCFG, synth.synth_rar_state(RCFG, seed=12, logit_scale=8.0), max_batch=64)


def test_rar_1280_teacher_forced_128_rows(pv, rar):
    """RAR.generate's 256 forward passes for 64 conditions under guidance (cond rows then uncond rows): logits at 23 steps
    (every 8th logit) within 5e-4."""
    toks = pv["rar_tokens"].astype(np.int64)
    B = toks.shape[0]
    cond = torch.from_numpy(pv["rar_cond"].astype(np.int64)) + RCFG.codebook_size + 1
    both = torch.cat([cond, torch.full_like(cond, RCFG.none_condition_id)]).cuda()
    rar.forward_position(torch.full((2 * B,), -1, dtype=torch.int64).cuda(), both, 0)
    want = {int(s): i for i, s in enumerate(pv["rar_steps"])}
    tok = both
    worst = 0.0
    for n in range(256):
        lg = rar.forward_position(tok, both, n + 1)
        if n in want:
            d = float(np.abs(lg[:, ::8].cpu().numpy() - pv["rar_logits"][want[n]]).max())
            worst = max(worst, d)
            assert d < ATOL, (n, d)
        t = torch.from_numpy(toks[:, n])
        tok = torch.cat([t, t]).cuda()
    print(f"max |dlogit| over {len(want)} steps x 128 rows: {worst:.2e}")


@pytest.mark.parametrize("graph", [True, False])
def test_rar_1280_watermarked_loop_256_steps(pv, kat, rar, graph):
    wm = _wm(kat["keys"]["rar"])
    torch.manual_seed(21)
    torch.rand(4, 1)   # preprocess_condition's label-drop draw precedes the sampling noise (rar.py:305)
    q = torch.stack([torch.empty(4, 1024).exponential_(1) for _ in range(256)]).cuda()
    toks = rar.generate(torch.from_numpy(pv["rar_loop_cond"]).cuda(), q, R.cfg_scales(256, 4.0, 0.0), 1.0, wm.wm_ctx(), use_graph=graph)
    assert np.array_equal(toks.cpu().numpy(), pv["rar_loop_tokens"].astype(np.int64))
    p = wm.detect(toks).cpu().numpy()
    assert np.allclose(np.log10(p), np.log10(pv["rar_loop_pvals"]), rtol=0, atol=1e-9)


def test_fp32_mfma_opt_out_matches_the_bf16_pipe_on_the_reference_logits(pv):
    """WMAR_NO_BX=1 (read at engine creation, production build): QKV and the output projection stay on the fp32-input MFMA kernels --
    the opt-out for models whose activations may be non-finite (the bf16-piece split turns inf into NaN, bx_split.h).  Both paths
    reproduce the reference's production-width logits."""
    import os
    from wmar_amd.models.engine import GPTEngine
    gcfg = synth.GPTConfig(vocab_size=16384, block_size=256, n_layer=2, n_head=24, n_embd=1536)
    sd = synth.synth_gpt_state(gcfg, seed=9, logit_scale=10.0)
    seq = torch.from_numpy(pv["gpt_seq"].astype(np.int64)).cuda()
    outs = {}
    for tag, env in (("bx", None), ("fp32", "1")):
        if env is None:
            os.environ.pop("WMAR_NO_BX", None)
        else:
            os.environ["WMAR_NO_BX"] = env
        try:
            eng = GPTEngine(gcfg, sd, max_batch=64)
            assert ("bf16 pipe" in eng.plan_info(64)["qkv"]) == (env is None)
            outs[tag] = torch.stack([eng.decode_step(seq[:, t], t) for t in range(2)]).cpu().numpy()      # positions 0 and 1 = fixture rows 0 and 1
        finally:
            os.environ.pop("WMAR_NO_BX", None)
    for tag in outs:
        for t in range(2):
            assert np.abs(outs[tag][t][:, ::64] - pv["gpt_logits"][t]).max() < 5e-4, tag
    assert np.abs(outs["bx"] - outs["fp32"]).max() < 2e-4
