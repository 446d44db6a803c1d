"""The CPU oracle (oracle/) against vectors produced by running the reference
(tests/golden/make_golden.py) -- pins the oracle before it is used as a checker."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import model_oracle as M
from oracle import wm_oracle as W
from wmar_amd.utils import synth


def test_randperm_kat(kat):
    for e in kat["randperm"]:
        p = W.randperm(e["seed"], e["n"])
        assert p[:8].tolist() == e["first"][: len(p[:8])]
        assert p[-4:].tolist() == e["last"]
        assert hashlib.sha256(p.astype("<i8").tobytes()).hexdigest() == e["sha256"]


def test_randperm_vs_torch_live():
    for n, seed in [(17, 1), (971, 99), (4096, 2**40 + 3)]:
        g = torch.Generator().manual_seed(seed)
        assert torch.randperm(n, generator=g).numpy().tolist() == W.randperm(seed, n).tolist()
    # two consecutive randperms on one stream (stratified split)
    g = torch.Generator().manual_seed(5)
    a, b = torch.randperm(10, generator=g), torch.randperm(20, generator=g)
    k = W.KeyParams(list(range(10)), list(range(10, 30)), 30, 1.0)
    gl = W.greenlist_for_seed(k, 5)
    assert gl[:10].tolist() == a.tolist() and (gl[10:] - 10).tolist() == b.tolist()


def test_survey_appendix_b_values():
    # SURVEY.md Appendix B, captured from the reference
    assert W.randperm(0, 971, 5).tolist() == [530, 180, 439, 611, 435]
    assert W.randperm(12345, 15413, 5).tolist() == [10105, 7602, 3995, 12666, 13721]
    assert W.randperm(7, 57344, 5).tolist() == [20655, 34632, 35451, 41458, 22367]


def test_greenlists_kat(kat, key_factory):
    for name, e in kat["keys"].items():
        assert e["dead_is_ascending"]
        key = key_factory(e)
        assert hashlib.sha256(key.dead.astype("<i8").tobytes()).hexdigest() == e["dead_sha256"]
        for c in e["contexts"]:
            gl = W.greenlist(key, sum(c["ctx"]))
            assert len(gl) == c["len"], name
            assert gl[:8].tolist() == c["first"]
            assert gl[-6:].tolist() == c["last"]
            assert hashlib.sha256(gl.astype("<i8").tobytes()).hexdigest() == c["sha256"]
        if "sha256_sorted_ctx0_15" in e:
            h = hashlib.sha256()
            for s in range(16):
                h.update(np.sort(W.greenlist(key, s)).astype("<i4").tobytes())
            assert h.hexdigest() == e["sha256_sorted_ctx0_15"]


def test_survey_key_hashes(kat):
    assert kat["keys"]["taming"]["sha256_sorted_ctx0_15"] == \
        "280a8fbc2493cd3e592b787b9fd13e714e87839aaf1d21b80b71ae59ee76af75"
    assert kat["keys"]["rar"]["sha256_sorted_ctx0_15"] == \
        "7e914db4a775cde787965f3b80dd651a8619e1024946fd20f624a4422ba6c8ff"


def test_key_table_rows_match_greenlists(kat, key_factory):
    key = key_factory(kat["keys"]["rar"])
    tab = W.key_table(key, 3, 5)
    for r in range(5):
        bits = np.unpackbits(tab[r].view(np.uint8), bitorder="little")[: key.vocab]
        assert sorted(np.nonzero(bits)[0].tolist()) == sorted(W.greenlist(key, 3 + r).tolist())


def test_betainc(kat):
    for e in kat["betainc"]:
        v = W.betainc_int(e["a"], e["b"], e["x"])
        if e["v"] != e["v"]:
            assert v != v
        else:
            assert abs(np.log(v) - np.log(e["v"])) < 1e-10, e


def test_process_logits(golden, kat, key_factory):
    key = key_factory(kat["keys"]["taming"])
    got = W.process_logits(key, golden["proc_taming_past"], golden["proc_taming_logits"], 2.0)
    assert np.array_equal(got, golden["proc_taming_out"])
    key2 = key_factory(kat["keys"]["taming_h2_g50"])
    lg = golden["proc_taming_logits"][:2]
    assert np.array_equal(W.process_logits(key2, golden["proc_h2_short_past"], lg, 1.5), golden["proc_h2_short_out"])
    assert np.array_equal(golden["proc_h2_short_out"], lg)  # too-short context: untouched
    assert np.array_equal(W.process_logits(key2, golden["proc_h2_past"], lg, 1.5), golden["proc_h2_out"])
    k3 = key_factory(kat["keys"]["taming"], seed="spatial", context_size=3, spatial_dim=4)
    assert np.array_equal(W.process_logits(k3, golden["proc_sp3_past"], lg, 2.0), golden["proc_sp3_out"])
    k1 = key_factory(kat["keys"]["taming"], seed="spatial", context_size=1, spatial_dim=4)
    assert np.array_equal(W.process_logits(k1, golden["proc_sp1a_past"], lg, 2.0), golden["proc_sp1a_out"])
    assert np.array_equal(W.process_logits(k1, golden["proc_sp1b_past"], lg, 2.0), golden["proc_sp1b_out"])


SMALL_GPT = synth.GPTConfig(vocab_size=16384, block_size=16, n_layer=2, n_head=4, n_embd=128)
LOOPS = {"k250p92": (250, 0.92, 1.0), "k100p80T13": (100, 0.8, 1.3), "nok_p95": (None, 0.95, 0.9),
         "k50nop": (50, None, 1.0), "plain": (None, None, 1.0)}


def test_gpt_step_logits(golden):
    sd = synth.synth_gpt_state(SMALL_GPT, seed=3, logit_scale=40.0)
    seq = torch.from_numpy(golden["gpt_seq"])
    pk = pv = None
    for t in range(seq.shape[1]):
        lg, nk, nv = M.gpt_step(sd, SMALL_GPT.n_head, seq[:, t:t + 1], pk, pv, t)
        pk = nk if pk is None else [torch.cat((a, b), -2) for a, b in zip(pk, nk)]
        pv = nv if pv is None else [torch.cat((a, b), -2) for a, b in zip(pv, nv)]
        np.testing.assert_allclose(lg.numpy()[:, ::16], golden["gpt_logits"][t], rtol=0, atol=2e-4)
        assert np.array_equal(lg.numpy().argmax(-1), golden["gpt_logits_argmax"][t])


def test_gpt_prefix_pass_equals_the_reference_logits(golden):
    """the one-pass masked prefix (GPT.forward_with_past with past=None, what tests/test_gpu_depth.py checks the 48-layer engine
    against) gives the reference's incremental logits at every position."""
    sd = synth.synth_gpt_state(SMALL_GPT, seed=3, logit_scale=40.0)
    seq = torch.from_numpy(golden["gpt_seq"])
    full = M.gpt_prefix(sd, SMALL_GPT.n_head, seq).numpy()
    for t in range(seq.shape[1]):
        np.testing.assert_allclose(full[:, t, ::16], golden["gpt_logits"][t], rtol=0, atol=2e-4)
        assert np.array_equal(full[:, t].argmax(-1), golden["gpt_logits_argmax"][t])
    sel = M.gpt_prefix(sd, SMALL_GPT.n_head, seq, [0, seq.shape[1] - 1]).numpy()
    assert np.array_equal(sel, full[:, [0, seq.shape[1] - 1]])


def test_sampling_stage_on_reference_logits(golden, kat, key_factory):
    """Stage-level: reference model logits + the noise multinomial drew -> the reference's tokens."""
    key = key_factory(kat["keys"]["taming"])
    toks = golden["loop_k250p92_tokens"]
    cond = golden["loop_cond"]
    for n in range(golden["loop_logits"].shape[0]):
        past = np.concatenate([cond, toks[:, :n]], axis=1)
        lg = W.process_logits(key, past, golden["loop_logits"][n], 2.0)
        got = W.sample_rows(lg, golden["loop_q"][n], 1.0, 250, 0.92)
        assert got.tolist() == toks[:, n].tolist(), n


@pytest.mark.parametrize("tag", list(LOOPS))
def test_sampling_loop_tokens(golden, kat, key_factory, tag):
    """End to end: oracle GPT + oracle sampler reproduce the reference's token ids."""
    tk, tp, T = LOOPS[tag]
    key = key_factory(kat["keys"]["taming"])
    sd = synth.synth_gpt_state(SMALL_GPT, seed=3, logit_scale=40.0)
    torch.manual_seed(11)
    toks = M.sample_with_past(sd, SMALL_GPT.n_head, torch.from_numpy(golden["loop_cond"]), 16, T, tk, tp, key, 2.0)
    assert np.array_equal(toks.numpy(), golden[f"loop_{tag}_tokens"])


def test_sampling_loop_static_cache_equals_the_cat_form(golden, kat, key_factory):
    """model_oracle's preallocated-cache form (used by the full-size end-to-end GPU test) against the reference's own
    tokens, and its logits against the torch.cat form step by step."""
    tk, tp, T = LOOPS["wm"] if "wm" in LOOPS else list(LOOPS.values())[0]
    tag = "wm" if "wm" in LOOPS else list(LOOPS)[0]
    key = key_factory(kat["keys"]["taming"])
    sd = synth.synth_gpt_state(SMALL_GPT, seed=3, logit_scale=40.0)
    ra, rb = [], []
    torch.manual_seed(11)
    a = M.sample_with_past(sd, SMALL_GPT.n_head, torch.from_numpy(golden["loop_cond"]), 16, T, tk, tp, key, 2.0, record=ra)
    torch.manual_seed(11)
    b = M.sample_with_past(sd, SMALL_GPT.n_head, torch.from_numpy(golden["loop_cond"]), 16, T, tk, tp, key, 2.0, record=rb, static_cache=True)
    assert np.array_equal(b.numpy(), golden[f"loop_{tag}_tokens"]) and np.array_equal(a.numpy(), b.numpy())
    assert max(float(np.abs(x["logits"] - y["logits"]).max()) for x, y in zip(ra, rb)) < 1e-5


def test_sampling_loop_unwatermarked(golden):
    sd = synth.synth_gpt_state(SMALL_GPT, seed=3, logit_scale=40.0)
    torch.manual_seed(11)
    toks = M.sample_with_past(sd, SMALL_GPT.n_head, torch.from_numpy(golden["loop_cond"]), 16, 1.0, 250, 0.92)
    assert np.array_equal(toks.numpy(), golden["loop_nowm_tokens"])


def _cmp_p(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.all(np.abs(np.log10(a[m]) - np.log10(b[m])) < 1e-9)
    assert np.all(np.abs(a[m] - b[m]) < 1e-5)  # the north_star tolerance


def test_detector(golden, kat, key_factory):
    key = key_factory(kat["keys"]["taming"])
    pv, ns, ng = W.detect(key, golden["det_codes_rand"])
    _cmp_p(pv, golden["det_pvals_rand"])
    _cmp_p(pv, [0.0808584840378873, 0.45210172026709927])
    pv, ns, ng, masks = W.detect(key, golden["loop_k250p92_tokens"], return_masks=True)
    _cmp_p(pv, golden["det_pvals_gen"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_gen"])
    pv, ns, ng, masks = W.detect(key, golden["det_codes_rep"], return_masks=True)
    _cmp_p(pv, golden["det_pvals_rep"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_rep"])
    assert ns[0] == (golden["det_masks_rep"][0] >= 0).sum()
    k2 = key_factory(kat["keys"]["taming_h2_g50"])
    pv, ns, ng, masks = W.detect(k2, golden["det_codes_rand"][:, :40], return_masks=True)
    _cmp_p(pv, golden["det_pvals_h2"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_h2"])
    kr = key_factory(kat["keys"]["rar"])
    _cmp_p(W.detect(kr, (np.arange(256) % 1024)[None])[0], golden["det_pvals_rar"])
    _cmp_p(W.detect(kr, (np.arange(256) % 1024)[None])[0], [0.39578004086406304])
    kc = key_factory(kat["keys"]["chameleon_fixed"])
    _cmp_p(W.detect(kc, golden["det_codes_cham"])[0], golden["det_pvals_cham"])
    k3 = key_factory(kat["keys"]["taming"], seed="spatial", context_size=3, spatial_dim=4)
    pv, ns, ng, masks = W.detect(k3, golden["det_codes_sp"], return_masks=True)
    _cmp_p(pv, golden["det_pvals_sp3"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_sp3"])
    k1 = key_factory(kat["keys"]["taming"], seed="spatial", context_size=1, spatial_dim=4)
    pv, ns, ng, masks = W.detect(k1, golden["det_codes_sp"], return_masks=True)
    _cmp_p(pv, golden["det_pvals_sp1"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_sp1"])


def test_detector_short_input_raises(kat, key_factory):
    key = key_factory(kat["keys"]["taming"])
    with pytest.raises(ValueError):
        W.detect(key, np.array([[3]]))


SMALL_VQ = synth.VQConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32,
                          z_channels=16, embed_dim=8, n_embed=512)


def test_vqgan(golden):
    sd = synth.synth_vq_state(SMALL_VQ, seed=5)
    img = M.codes_to_images(sd, SMALL_VQ, torch.from_numpy(golden["vq_codes"]))
    np.testing.assert_allclose(img.numpy(), golden["vq_images"], rtol=0, atol=1e-5)
    z = M.encode_prequant(sd, SMALL_VQ, torch.from_numpy(golden["vq_images"]))
    np.testing.assert_allclose(z.numpy(), golden["vq_prequant"], rtol=0, atol=1e-5)
    codes = M.images_to_codes(sd, SMALL_VQ, torch.from_numpy(golden["vq_images"]))
    assert np.array_equal(codes.numpy(), golden["vq_codes_roundtrip"])


def test_oracle_spatial_loop_reproduces_reference_tokens(kat, key_factory):
    """the oracle's sampling loop under SPATIAL seeding equals the reference's (tests/golden/spatial_vectors.npz)."""
    import os
    import numpy as np
    import torch
    from oracle import model_oracle as M
    from tests.conftest import REPO
    from wmar_amd.utils import synth
    sv = np.load(os.path.join(REPO, "tests", "golden", "spatial_vectors.npz"))
    cfg = synth.GPTConfig(vocab_size=16384, block_size=16, n_layer=2, n_head=4, n_embd=128)
    sd = synth.synth_gpt_state(cfg, seed=3, logit_scale=40.0)
    for h in (1, 3):
        key = key_factory(kat["keys"]["taming"], seed="spatial", context_size=h, spatial_dim=4)
        torch.manual_seed(11)
        q = [torch.empty(4, 16384).exponential_(1) for _ in range(16)]
        toks = M.sample_with_past(sd, cfg.n_head, torch.from_numpy(sv["cond"]), 16, 1.0, 250, 0.92, key, 2.0, q_source=lambda n, b, v: q[n])
        assert np.array_equal(toks.numpy(), sv[f"tokens_h{h}"]), h


@pytest.mark.parametrize("h", [4, 5])
def test_oracle_linear_context_beyond_3_equals_the_reference(kat, key_factory, h):
    """LINEAR seeding with h = 4 / 5 (the reference takes any context size, gentime_watermark.py:236-241): the oracle's loop,
    detector and logit processor against tests/golden/linear_h_vectors.npz (make_golden.py linear_h_vectors)."""
    import os
    from tests.conftest import REPO
    lv = np.load(os.path.join(REPO, "tests", "golden", "linear_h_vectors.npz"))
    cfg = synth.GPTConfig(vocab_size=16384, block_size=24, n_layer=2, n_head=4, n_embd=128)
    sd = synth.synth_gpt_state(cfg, seed=3, logit_scale=40.0)
    key = key_factory(kat["keys"]["taming"], context_size=h)
    torch.manual_seed(11)
    q = [torch.empty(4, 16384).exponential_(1) for _ in range(24)]
    toks = M.sample_with_past(sd, cfg.n_head, torch.from_numpy(lv["cond"]), 24, 1.0, 250, 0.92, key, 2.0, q_source=lambda n, b, v: q[n])
    assert np.array_equal(toks.numpy(), lv[f"tokens_h{h}"])
    pv, ns, ng, masks = W.detect(key, lv[f"tokens_h{h}"], return_masks=True)
    assert np.allclose(pv, lv[f"pvals_h{h}"], rtol=1e-9, atol=0, equal_nan=True)
    assert np.array_equal(np.array(masks, dtype=np.int8), lv[f"masks_h{h}"])
    past = lv[f"proc_past_h{h}"]
    assert np.array_equal(W.process_logits(key, past, lv["proc_logits"], 2.0), lv[f"proc_out_h{h}"])
    assert np.array_equal(W.process_logits(key, past[:, :h - 1], lv["proc_logits"], 2.0), lv[f"proc_short_out_h{h}"])
    assert np.array_equal(lv[f"proc_short_out_h{h}"], lv["proc_logits"])          # too-short context: every row skipped
