"""Run-to-run reproducibility of the decode engines at their production shapes.

Regression tests for the round-3 finding in csrc/common.h (prod_f64): a chain of dependent v_fmac_f64 in the fused QKV launch's
LayerNorm statistics returned a slightly different sum of squares for rows 48..63 about once per 25,000 launches (one pass in five
differed by ~1e-5 in those rows' logits; first seen as a token flip in a replay of the captured generation graph).

* Taming (48 layers x 1536, 64 rows): 30 teacher-forced passes over the 256 positions must return the first pass's logits bit for bit
  (at the pre-fix rate of one bad pass in five, 30 passes miss a regression with probability 0.8^29 ~ 0.15 %), AND ten replays of the
  captured 256-step generation graph must return the first replay's tokens and traced logits bit for bit -- the path the bug was
  first seen on.
* RAR-XL (128 rows under guidance) and Chameleon-7B (48 rows): one repeated teacher-forced pass each (the long versions are
  scripts/stress_rar_cham.py and scripts/stress_logits.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from wmar_amd.utils import synth  # noqa: E402


def _diff_msg(what, lg, ref):
    rows = (lg != ref).any(-1).nonzero().view(-1).tolist()
    return f"{what}: rows {rows[:12]} ({len(rows)}) differ by up to {float((lg.float() - ref.float()).abs().max()):.2e}"


@pytest.fixture(scope="module")
def taming_engine():
    from wmar_amd.models.engine import GPTEngine
    cfg = synth.TAMING_GPT
    return cfg, GPTEngine(cfg, synth.synth_gpt_state(cfg, seed=0, logit_scale=10.0), max_batch=64)


def test_taming_decode_step_is_bit_reproducible_over_thirty_passes(taming_engine):
    cfg, eng = taming_engine
    seq = torch.randint(0, cfg.vocab_size, (64, 256), generator=torch.Generator().manual_seed(5)).cuda()
    ref = torch.empty(256, 64, cfg.vocab_size, device="cuda")
    for t in range(256):
        ref[t].copy_(eng.decode_step(seq[:, t], t))
    for p in range(29):
        for t in range(256):
            lg = eng.decode_step(seq[:, t], t)
            if not torch.equal(lg, ref[t]):
                raise AssertionError(_diff_msg(f"pass {p + 1}, position {t}", lg, ref[t]))


def test_taming_generation_graph_replays_are_bit_reproducible(taming_engine):
    """Ten replays of the captured 256-step graph (watermark + top-k + top-p sampler inside): tokens AND the traced logits of every step."""
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    cfg, eng = taming_engine
    import os
    ids = []
    for line in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wmar_amd", "assets", "vqgan_alive_ids.txt")):
        ids.extend(int(t) for t in line.replace(",", " ").split())
    alive = torch.tensor(ids)
    dead = torch.tensor(sorted(set(range(cfg.vocab_size)) - set(ids)))
    wm = GentimeWatermark({"alive_ids": alive, "dead_ids": dead, "embedding": None}, cfg.vocab_size,
                          SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
    cond = torch.tensor([(i * 37) % 1000 for i in range(64)], dtype=torch.int64, device="cuda")
    torch.manual_seed(1)
    q = torch.empty(256, 64, cfg.vocab_size, device="cuda").exponential_(1)
    ctx = wm.wm_ctx()
    tok0, tr0 = eng.generate(cond, 256, q, temperature=1.0, top_k=250, top_p=0.92, wm_ctx=ctx, use_graph=True, trace_logits=True)
    tok0, tr0 = tok0.clone(), tr0.clone()
    for rep in range(9):
        tok, tr = eng.generate(cond, 256, q, temperature=1.0, top_k=250, top_p=0.92, wm_ctx=ctx, use_graph=True, trace_logits=True)
        if not torch.equal(tr, tr0):
            t = int((tr != tr0).flatten(1).any(1).nonzero()[0])
            raise AssertionError(_diff_msg(f"graph replay {rep + 1}, step {t}", tr[t], tr0[t]))
        assert torch.equal(tok, tok0), f"graph replay {rep + 1}: tokens differ although the traced logits agree"


@pytest.mark.parametrize("B", [1, 5, 10])
def test_taming_small_batch_plan_is_bit_reproducible(taming_engine, B):
    """the weight-streaming plan of 1..12 rows (decode_small.h: wave butterflies, LDS segment sums in fixed order, no atomics): twelve
    teacher-forced passes over 256 positions and five replays of the captured 256-step loop return the first pass's bits"""
    cfg, eng = taming_engine
    assert "k_sgemv" in eng.plan_info(B)["qkv"]
    seq = torch.randint(0, cfg.vocab_size, (B, 256), generator=torch.Generator().manual_seed(50 + B)).cuda()
    ref = torch.empty(256, B, cfg.vocab_size, device="cuda")
    for t in range(256):
        ref[t].copy_(eng.decode_step(seq[:, t], t))
    for p in range(11):
        for t in range(256):
            lg = eng.decode_step(seq[:, t], t)
            if not torch.equal(lg, ref[t]):
                raise AssertionError(_diff_msg(f"batch {B}, pass {p + 1}, position {t}", lg, ref[t]))
    torch.manual_seed(2)
    q = torch.empty(256, B, cfg.vocab_size, device="cuda").exponential_(1)
    cond = torch.tensor([1, 9, 232, 340, 568, 656, 703, 814, 937, 975][:B], device="cuda")
    tok0, tr0 = eng.generate(cond, 256, q, temperature=1.0, top_k=250, top_p=0.92, use_graph=True, trace_logits=True)
    tok0, tr0 = tok0.clone(), tr0.clone()
    for rep in range(4):
        tok, tr = eng.generate(cond, 256, q, temperature=1.0, top_k=250, top_p=0.92, use_graph=True, trace_logits=True)
        assert torch.equal(tr, tr0) and torch.equal(tok, tok0), f"batch {B}, graph replay {rep + 1}"


def test_taming_persistent_step_is_bit_reproducible():
    """the opt-in persistent step (decode_persist.h): its rows cross workgroups INSIDE a launch (agent-scope atomics behind a
    device-wide barrier) -- a stale read would show as a run-to-run difference.  48 layers, batch 5: six teacher-forced passes over 128
    positions and three replays of the captured loop, bit for bit; zero barrier fallbacks."""
    import os
    from wmar_amd.models.engine import GPTEngine
    cfg = synth.TAMING_GPT
    os.environ["WMAR_PERSIST"] = "1"
    try:
        eng = GPTEngine(cfg, synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=10.0), max_batch=5)
    finally:
        del os.environ["WMAR_PERSIST"]
    if "persistent" not in eng.plan_info(5).get("path", ""):
        pytest.skip("the persistent step is not available on this device")
    seq = torch.randint(0, cfg.vocab_size, (5, 128), generator=torch.Generator().manual_seed(77)).cuda()
    ref = torch.empty(128, 5, cfg.vocab_size, device="cuda")
    for t in range(128):
        ref[t].copy_(eng.decode_step(seq[:, t], t))
    for p in range(5):
        for t in range(128):
            lg = eng.decode_step(seq[:, t], t)
            if not torch.equal(lg, ref[t]):
                raise AssertionError(_diff_msg(f"persistent step, pass {p + 1}, position {t}", lg, ref[t]))
    torch.manual_seed(3)
    q = torch.empty(256, 5, cfg.vocab_size, device="cuda").exponential_(1)
    cond = torch.tensor([1, 9, 232, 340, 568], device="cuda")
    tok0 = eng.generate(cond, 256, q, temperature=1.0, top_k=250, top_p=0.92, use_graph=True).clone()
    for rep in range(3):
        assert torch.equal(eng.generate(cond, 256, q, temperature=1.0, top_k=250, top_p=0.92, use_graph=True), tok0), rep
    assert eng.plan_info(5)["barrier_fallbacks"] == "0"


def test_rar_xl_step_is_bit_reproducible():
    from wmar_amd.models.engine import RAREngine
    cfg = synth.RAR_XL
    eng = RAREngine(cfg, synth.synth_rar_state(cfg, seed=12, logit_scale=8.0), max_batch=64)
    g = torch.Generator().manual_seed(3)
    cond = torch.randint(0, 1000, (64,), generator=g) + cfg.codebook_size + 1
    both = torch.cat([cond, torch.full_like(cond, cfg.none_condition_id)]).cuda()
    toks = torch.randint(0, cfg.codebook_size, (64, 256), generator=g)
    ref = []
    for p in range(3):
        eng.forward_position(torch.full((128,), -1, dtype=torch.int64).cuda(), both, 0)
        tok = both
        for n in range(256):
            lg = eng.forward_position(tok, both, n + 1)
            if p == 0:
                ref.append(lg.clone())
            elif not torch.equal(lg, ref[n]):
                raise AssertionError(_diff_msg(f"RAR-XL pass {p}, step {n}", lg, ref[n]))
            t = toks[:, n]
            tok = torch.cat([t, t]).cuda()


def test_chameleon_7b_step_is_bit_reproducible():
    from wmar_amd.models.engine import ChameleonEngine
    cfg = synth.CHAMELEON_7B
    sd = synth.synth_chameleon_state(cfg, 0, "cuda", 8.0, gen_device="cuda")
    eng = ChameleonEngine(cfg, sd, max_batch=16, max_seq_len=192)
    del sd
    M, T = 48, 128
    seq = torch.randint(0, 65536, (M, T), generator=torch.Generator().manual_seed(4)).cuda()
    ref = []
    for p in range(3):
        for t in range(T):
            pos = torch.full((M,), t, dtype=torch.int32, device="cuda")
            lg = eng.forward_tokens(seq[:, t].contiguous(), pos)
            if p == 0:
                ref.append(lg.clone())
            elif not torch.equal(lg, ref[t]):
                raise AssertionError(_diff_msg(f"Chameleon-7B pass {p}, position {t}", lg, ref[t]))
