"""HIP GPT decode engine against the torch-fp32 oracle and the reference's golden logits/tokens."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model_oracle as M  # noqa: E402
from oracle import wm_oracle as W  # noqa: E402
from tests.test_gpu_watermark import _wm  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402

SMALL = synth.GPTConfig(vocab_size=16384, block_size=16, n_layer=2, n_head=4, n_embd=128)
LOOPS = {"k250p92": (250, 0.92, 1.0), "k100p80T13": (100, 0.8, 1.3), "nok_p95": (None, 0.95, 0.9),
         "k50nop": (50, None, 1.0), "plain": (None, None, 1.0)}


@pytest.fixture(scope="module")
def small_engine():
    from wmar_amd.models.engine import GPTEngine
    sd = synth.synth_gpt_state(SMALL, seed=3, logit_scale=40.0)
    return GPTEngine(SMALL, sd, max_batch=64), sd


def test_decode_step_logits_golden(golden, small_engine):
    eng, _ = small_engine
    seq = torch.from_numpy(golden["gpt_seq"]).cuda()
    for t in range(seq.shape[1]):
        lg = eng.decode_step(seq[:, t], t).cpu().numpy()
        np.testing.assert_allclose(lg[:, ::16], golden["gpt_logits"][t], rtol=0, atol=5e-4)
        assert np.array_equal(lg.argmax(-1), golden["gpt_logits_argmax"][t])


@pytest.mark.parametrize("B", [1, 5, 33, 64])
def test_decode_step_vs_oracle_batches(small_engine, B):
    eng, sd = small_engine
    rs = np.random.RandomState(B)
    seq = torch.from_numpy(rs.randint(0, 16384, size=(B, 5)).astype(np.int64))
    pk = pv = None
    for t in range(5):
        ref, nk, nv = M.gpt_step(sd, SMALL.n_head, seq[:, t:t + 1], pk, pv, t)
        pk = nk if pk is None else [torch.cat((a, b), -2) for a, b in zip(pk, nk)]
        pv = nv if pv is None else [torch.cat((a, b), -2) for a, b in zip(pv, nv)]
        lg = eng.decode_step(seq[:, t].cuda(), t).cpu().numpy()
        np.testing.assert_allclose(lg, ref.numpy(), rtol=0, atol=5e-4)


@pytest.mark.parametrize("tag", list(LOOPS))
@pytest.mark.parametrize("graph", [True, False])
def test_generate_reproduces_reference_tokens(golden, kat, small_engine, tag, graph):
    """Full loop (decode + fused watermark sampler, hipGraph replay) on the CPU generator's noise:
    token ids equal the reference's."""
    eng, _ = small_engine
    tk, tp, T = LOOPS[tag]
    wm = _wm(kat["keys"]["taming"])
    cond = torch.from_numpy(golden["loop_cond"]).view(-1)
    torch.manual_seed(11)
    q = torch.stack([torch.empty(4, 16384).exponential_(1) for _ in range(16)]).cuda()
    toks = eng.generate(cond.cuda(), 16, q, T, tk, tp, wm.wm_ctx(), use_graph=graph)
    assert np.array_equal(toks.cpu().numpy(), golden[f"loop_{tag}_tokens"])


def test_generate_unwatermarked(golden, small_engine):
    eng, _ = small_engine
    torch.manual_seed(11)
    q = torch.stack([torch.empty(4, 16384).exponential_(1) for _ in range(16)]).cuda()
    toks = eng.generate(torch.from_numpy(golden["loop_cond"]).view(-1).cuda(), 16, q, 1.0, 250, 0.92, None)
    assert np.array_equal(toks.cpu().numpy(), golden["loop_nowm_tokens"])


def test_generate_trace_and_determinism(kat, small_engine):
    eng, sd = small_engine
    wm = _wm(kat["keys"]["taming"])
    g = torch.Generator(device="cuda").manual_seed(5)
    B, steps = 37, 16
    q = torch.empty(steps, B, 16384, device="cuda").exponential_(1, generator=g)
    cond = (torch.arange(B) * 37 % 1000).cuda()
    t1, trace = eng.generate(cond, steps, q, 1.0, 250, 0.92, wm.wm_ctx(), trace_logits=True)
    t2 = eng.generate(cond, steps, q, 1.0, 250, 0.92, wm.wm_ctx(), use_graph=False)
    assert torch.equal(t1, t2)
    # the traced logits + the same noise through the CPU oracle give the same tokens
    key = W.KeyParams(wm._alive_host, wm._dead_host, 16384, 0.25)
    past = cond.view(-1, 1).cpu().numpy()
    for n in range(steps):
        lg = W.process_logits(key, past, trace[n].cpu().numpy(), 2.0)
        exp = W.sample_rows(lg, q[n].cpu().numpy(), 1.0, 250, 0.92)
        assert exp.tolist() == t1[:, n].cpu().tolist(), n
        past = np.concatenate([past, exp[:, None]], axis=1)


@pytest.mark.parametrize("h", [1, 3])
@pytest.mark.parametrize("graph", [True, False])
def test_generate_spatial_seeding_reproduces_reference_tokens(kat, small_engine, h, graph):
    """SeedStrategy.SPATIAL inside the captured step (the fused sampler's spatial context branch): the reference's loop with
    spatial_dim 4, h = 1 and h = 3 (tests/golden/spatial_vectors.npz, make_golden.py spatial_vectors)."""
    import os
    from tests.conftest import REPO
    sv = np.load(os.path.join(REPO, "tests", "golden", "spatial_vectors.npz"))
    eng, _ = small_engine
    wm = _wm(kat["keys"]["taming"], seed="spatial", h=h, spatial_dim=4)
    torch.manual_seed(11)
    q = torch.stack([torch.empty(4, 16384).exponential_(1) for _ in range(16)]).cuda()
    toks = eng.generate(torch.from_numpy(sv["cond"]).view(-1).cuda(), 16, q, 1.0, 250, 0.92, wm.wm_ctx(), use_graph=graph)
    assert np.array_equal(toks.cpu().numpy(), sv[f"tokens_h{h}"])
    pv, masks = wm.detect(toks, return_masks=True)
    assert np.allclose(pv.cpu().numpy(), sv[f"pvals_h{h}"], rtol=1e-9, atol=0, equal_nan=True)
    assert np.array_equal(np.array(masks, dtype=np.int8), sv[f"masks_h{h}"])


@pytest.mark.parametrize("h", [4, 5])
def test_generate_linear_context_beyond_3_reproduces_reference_tokens(kat, h):
    """LINEAR seeding with context sizes 4 and 5 -- the reference takes any `context_size` (gentime_watermark.py:236-241); the key
    table has h * (V - 1) + 1 rows.  The reference's loop, detector (p-values, masks) and logit processor, incl. the rows whose
    context is too short (tests/golden/linear_h_vectors.npz, make_golden.py linear_h_vectors)."""
    import os
    from tests.conftest import REPO
    from wmar_amd.models.engine import GPTEngine
    lv = np.load(os.path.join(REPO, "tests", "golden", "linear_h_vectors.npz"))
    cfg = synth.GPTConfig(vocab_size=16384, block_size=24, n_layer=2, n_head=4, n_embd=128)
    eng = GPTEngine(cfg, synth.synth_gpt_state(cfg, seed=3, logit_scale=40.0), max_batch=4)
    wm = _wm(kat["keys"]["taming"], h=h)
    assert wm.key_table().shape[0] == h * 16383 + 1
    torch.manual_seed(11)
    q = torch.stack([torch.empty(4, 16384).exponential_(1) for _ in range(24)]).cuda()
    for graph in (True, False):
        toks = eng.generate(torch.from_numpy(lv["cond"]).view(-1).cuda(), 24, q, 1.0, 250, 0.92, wm.wm_ctx(), use_graph=graph)
        assert np.array_equal(toks.cpu().numpy(), lv[f"tokens_h{h}"])
    pv, masks = wm.detect(toks, return_masks=True)
    assert np.allclose(pv.cpu().numpy(), lv[f"pvals_h{h}"], rtol=1e-9, atol=0, equal_nan=True)
    assert np.array_equal(np.array(masks, dtype=np.int8), lv[f"masks_h{h}"])
    proc = wm.spawn_logit_processor()
    past = torch.from_numpy(lv[f"proc_past_h{h}"]).cuda()
    lg = torch.from_numpy(lv["proc_logits"]).cuda()
    assert np.array_equal(proc(past_ids=past, logits=lg.clone()).cpu().numpy(), lv[f"proc_out_h{h}"])
    assert np.array_equal(proc(past_ids=past[:, :h - 1], logits=lg.clone()).cpu().numpy(), lv[f"proc_short_out_h{h}"])
    from wmar_amd.watermarking.gentime_watermark import clear_key_cache
    clear_key_cache()       # 128 / 160 MiB tables
