"""HIP watermark kernels (through the C ABI) against the CPU oracle and the golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import wm_oracle as W  # noqa: E402
from tests.conftest import load_ids  # noqa: E402


def _wm(cfg, delta=2.0, **over):
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    alive = load_ids(cfg["alive"])
    dead = sorted(set(range(cfg["vocab"])) - set(alive))
    vq = {"alive_ids": torch.tensor(alive), "dead_ids": torch.tensor(dead), "embedding": None}
    kw = dict(seed=cfg["seed"], h=cfg["h"], spatial_dim=16)
    kw.update(over)
    return GentimeWatermark(vq, cfg["vocab"], SeedStrategy(kw["seed"]), SplitStrategy(cfg["split"]), kw["h"], delta,
                            cfg["gamma"], device="cuda", spatial_dim=kw["spatial_dim"])


def test_single_hip_runtime_loaded():
    import wmar_amd._lib as l
    l.load()
    libs = {ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}
    assert len(libs) == 1, libs


def test_key_table_on_device_matches_oracle(kat, key_factory):
    for name in ("taming", "rar", "chameleon_fixed", "taming_rand"):
        wm = _wm(kat["keys"][name])
        tab = wm.key_table().cpu().numpy().view(np.uint32)
        key = key_factory(kat["keys"][name])
        rows = [0, 5, tab.shape[0] - 1] if tab.shape[0] > 1 else [0]
        for r in rows:
            assert np.array_equal(tab[r], W.key_table(key, r, 1)[0]), (name, r)


def test_process_logits_golden(golden, kat):
    wm = _wm(kat["keys"]["taming"])
    proc = wm.spawn_logit_processor()
    lg = torch.from_numpy(golden["proc_taming_logits"]).cuda()
    out = proc(past_ids=torch.from_numpy(golden["proc_taming_past"]).cuda(), logits=lg)
    assert out.data_ptr() == lg.data_ptr()  # in place, like the reference
    assert np.array_equal(out.cpu().numpy(), golden["proc_taming_out"])
    lg2 = torch.from_numpy(golden["proc_taming_logits"][:2])
    wm2 = _wm(kat["keys"]["taming_h2_g50"], delta=1.5)
    assert np.array_equal(wm2._process_logits(torch.from_numpy(golden["proc_h2_short_past"]).cuda(), lg2.clone().cuda()).cpu().numpy(),
                          golden["proc_h2_short_out"])
    assert np.array_equal(wm2._process_logits(torch.from_numpy(golden["proc_h2_past"]).cuda(), lg2.clone().cuda()).cpu().numpy(),
                          golden["proc_h2_out"])
    w3 = _wm(kat["keys"]["taming"], seed="spatial", h=3, spatial_dim=4)
    assert np.array_equal(w3._process_logits(torch.from_numpy(golden["proc_sp3_past"]).cuda(), lg2.clone().cuda()).cpu().numpy(),
                          golden["proc_sp3_out"])
    w1 = _wm(kat["keys"]["taming"], seed="spatial", h=1, spatial_dim=4)
    assert np.array_equal(w1._process_logits(torch.from_numpy(golden["proc_sp1a_past"]).cuda(), lg2.clone().cuda()).cpu().numpy(),
                          golden["proc_sp1a_out"])
    assert np.array_equal(w1._process_logits(torch.from_numpy(golden["proc_sp1b_past"]).cuda(), lg2.clone().cuda()).cpu().numpy(),
                          golden["proc_sp1b_out"])


def _sample_fused(wm, logits, past, q, T, tk, tp):
    import ctypes as C
    import wmar_amd._lib as l
    L = l.load()
    B, V = logits.shape
    lg = torch.from_numpy(np.ascontiguousarray(logits)).cuda()
    qq = torch.from_numpy(np.ascontiguousarray(q)).cuda()
    scratch = torch.empty_like(lg)
    tok = torch.empty(B, dtype=torch.int64, device="cuda")
    ctx = wm.wm_ctx() if wm is not None else None
    p = torch.from_numpy(np.ascontiguousarray(past)).cuda() if past is not None else None
    l.check(L.wmar_sample_fused(C.byref(ctx) if ctx is not None else None, lg.data_ptr(), B, V,
                                p.data_ptr() if p is not None else None, p.shape[1] if p is not None else 0,
                                p.shape[1] if p is not None else 0, float(T), int(tk) if tk else 0,
                                float(tp) if tp is not None else -1.0, qq.data_ptr(), scratch.data_ptr(), tok.data_ptr(),
                                l.stream_ptr()))
    return tok.cpu().numpy(), scratch.cpu().numpy()


def test_fused_sampler_on_reference_logits(golden, kat):
    """reference logits + reference noise -> the reference's token ids (bit-exact)."""
    wm = _wm(kat["keys"]["taming"])
    toks, cond = golden["loop_k250p92_tokens"], golden["loop_cond"]
    for n in range(golden["loop_logits"].shape[0]):
        past = np.concatenate([cond, toks[:, :n]], axis=1)
        got, _ = _sample_fused(wm, golden["loop_logits"][n], past, golden["loop_q"][n], 1.0, 250, 0.92)
        assert got.tolist() == toks[:, n].tolist(), n


@pytest.mark.parametrize("T,tk,tp", [(1.0, 250, 0.92), (1.3, 100, 0.8), (0.9, None, 0.95), (1.0, 50, None),
                                     (1.0, None, None), (0.7, 1, 0.5), (1.0, 16384, 0.0), (1.0, 300, 1.0)])
def test_fused_sampler_vs_oracle_random(kat, key_factory, T, tk, tp):
    """Seeded random rows, incl. heavy ties: tokens AND the kept set must equal the oracle's."""
    rs = np.random.RandomState(7)
    B, V = 12, 16384
    logits = (rs.randn(B, V) * 3).astype(np.float32)
    logits[1] = np.round(logits[1])            # many exact ties (top-k boundary inside a tie group)
    logits[2] = 0.0                            # all equal
    logits[3, :200] = 9.0                      # tie group straddling top-k / top-p boundaries
    logits[4, 5] = 60.0                        # one dominant token
    logits[5, 100:] = -np.inf                  # -inf entries present
    q = rs.exponential(size=(B, V)).astype(np.float32)
    past = rs.randint(0, 16384, size=(B, 3)).astype(np.int64)
    wm = _wm(kat["keys"]["taming"])
    key = key_factory(kat["keys"]["taming"])
    got, x = _sample_fused(wm, logits, past, q, T, tk, tp)
    biased = W.process_logits(key, past, logits, 2.0)
    exp, xs, ps = W.sample_rows(biased, q, T, tk, tp, return_all=True)
    assert got.tolist() == exp.tolist()
    # temperature-scaled biased logits are bit-identical
    assert np.array_equal(x, biased / np.float32(T))


def test_fused_sampler_no_watermark(golden):
    rs = np.random.RandomState(3)
    logits = (rs.randn(5, 1024) * 2).astype(np.float32)
    q = rs.exponential(size=(5, 1024)).astype(np.float32)
    got, _ = _sample_fused(None, logits, None, q, 1.0, None, None)
    assert got.tolist() == W.sample_rows(logits, q).tolist()


def test_detector_golden(golden, kat, key_factory):
    def cmp(pv, ref):
        pv = pv.cpu().numpy()
        assert np.array_equal(np.isnan(pv), np.isnan(ref))
        m = ~np.isnan(ref)
        assert np.all(np.abs(np.log10(pv[m]) - np.log10(ref[m])) < 1e-9)
        assert np.all(np.abs(pv[m] - ref[m]) < 1e-5)

    wm = _wm(kat["keys"]["taming"])
    cmp(wm.detect(torch.from_numpy(golden["det_codes_rand"])), golden["det_pvals_rand"])
    pv, masks = wm.detect(torch.from_numpy(golden["loop_k250p92_tokens"]), return_masks=True)
    cmp(pv, golden["det_pvals_gen"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_gen"])
    pv, masks = wm.detect(torch.from_numpy(golden["det_codes_rep"]), return_masks=True)
    cmp(pv, golden["det_pvals_rep"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_rep"])
    w2 = _wm(kat["keys"]["taming_h2_g50"])
    pv, masks = w2.detect(torch.from_numpy(golden["det_codes_rand"][:, :40]), return_masks=True)
    cmp(pv, golden["det_pvals_h2"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_h2"])
    cmp(_wm(kat["keys"]["rar"]).detect((torch.arange(256) % 1024).view(1, -1)), golden["det_pvals_rar"])
    cmp(_wm(kat["keys"]["chameleon_fixed"]).detect(torch.from_numpy(golden["det_codes_cham"])), golden["det_pvals_cham"])
    w3 = _wm(kat["keys"]["taming"], seed="spatial", h=3, spatial_dim=4)
    pv, masks = w3.detect(torch.from_numpy(golden["det_codes_sp"]), return_masks=True)
    cmp(pv, golden["det_pvals_sp3"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_sp3"])
    w1 = _wm(kat["keys"]["taming"], seed="spatial", h=1, spatial_dim=4)
    pv, masks = w1.detect(torch.from_numpy(golden["det_codes_sp"]), return_masks=True)
    cmp(pv, golden["det_pvals_sp1"])
    assert np.array_equal(np.array(masks, dtype=np.int8), golden["det_masks_sp1"])


def test_detector_vs_oracle_counts(kat, key_factory):
    rs = np.random.RandomState(5)
    codes = rs.randint(0, 16384, size=(16, 256)).astype(np.int64)
    codes[3, 100:] = codes[3, :156]       # many repeated bigrams
    codes[4] = 77                         # a single distinct bigram
    wm = _wm(kat["keys"]["taming"])
    pv, ns, ng = wm.detect_counts(torch.from_numpy(codes))
    epv, ens, eng = W.detect(key_factory(kat["keys"]["taming"]), codes)
    assert ns.cpu().numpy().tolist() == ens.tolist()
    assert ng.cpu().numpy().tolist() == eng.tolist()
    a, b = pv.cpu().numpy(), epv
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(b)
    assert np.all(np.abs(np.log10(a[m]) - np.log10(b[m])) < 1e-9)


def test_detector_short_raises(kat):
    wm = _wm(kat["keys"]["taming"])
    with pytest.raises(ValueError):
        wm.detect(torch.tensor([[3]]))


def test_process_logits_empty_past_is_a_no_op(kat):
    """RAR.generate calls the processor at step 0 with ids of shape [B, 0] (rar.py:420,451): the reference catches the
    per-row ValueError and leaves the logits alone (gentime_watermark.py:266-269); FIXED seeding needs no context and still biases."""
    rs = np.random.RandomState(0)
    lg = torch.from_numpy(rs.randn(3, 1024).astype(np.float32)).cuda()
    empty = torch.zeros(3, 0, dtype=torch.int64, device="cuda")
    for seed, h in (("linear", 1), ("linear", 2), ("spatial", 1)):
        wm = _wm(kat["keys"]["rar"], seed=seed, h=h)
        out = wm.spawn_logit_processor()(past_ids=empty, logits=lg.clone())
        assert torch.equal(out, lg), (seed, h)
    wm = _wm(kat["keys"]["rar"], seed="fixed", h=0)
    out = wm.spawn_logit_processor()(empty, lg.clone())        # positional call, as HF's LogitsProcessorList does
    moved = out != lg
    assert moved.sum(1).tolist() == [256, 256, 256] and torch.equal(out[moved], (lg + 2.0)[moved])
    # short-but-not-empty context under h = 2 is skipped too
    wm = _wm(kat["keys"]["rar"], seed="linear", h=2)
    one = torch.zeros(3, 1, dtype=torch.int64, device="cuda")
    assert torch.equal(wm.spawn_logit_processor()(past_ids=one, logits=lg.clone()), lg)


def test_linear_context_16_small_vocabulary_against_the_oracle():
    """LINEAR seeding at the largest context size the build takes (WMAR_MAX_CONTEXT 16; the reference takes any,
    gentime_watermark.py:236-241) on a 512-entry vocabulary (8177 key rows): logit processor incl. rows whose context is too short,
    detector counts / p-values / masks, token for token against the oracle."""
    from oracle import wm_oracle as W
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    V, h = 512, 16
    alive = list(range(0, V, 2)) + list(range(1, V, 4))
    dead = sorted(set(range(V)) - set(alive))
    vq = {"alive_ids": torch.tensor(alive), "dead_ids": torch.tensor(dead), "embedding": None}
    wm = GentimeWatermark(vq, V, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, h, 1.5, 0.25, device="cuda")
    key = W.KeyParams(alive, dead, V, 0.25, split="stratifiedrand", seed="linear", context_size=h)
    assert wm.key_table().shape[0] == h * (V - 1) + 1 and wm.key_table_bytes == wm.key_table().numel() * 4
    rs = np.random.RandomState(16)
    logits = rs.randn(6, V).astype(np.float32)
    for t in (3, 16, 40):                      # shorter than the context (rows skipped), exactly h, longer
        past = rs.randint(0, V, size=(6, t)).astype(np.int64)
        got = wm._process_logits(torch.from_numpy(past).cuda(), torch.from_numpy(logits).clone().cuda()).cpu().numpy()
        assert np.array_equal(got, W.process_logits(key, past, logits, 1.5)), t
    codes = rs.randint(0, V, size=(5, 256)).astype(np.int64)
    codes[1, 100:140] = codes[1, 40:80]        # repeated n-grams are scored once
    pv, ns, ng, masks = wm.detect_counts(torch.from_numpy(codes).cuda(), return_masks=True)
    rpv, rns, rng, rmasks = W.detect(key, codes, return_masks=True)
    assert np.array_equal(ns.cpu().numpy(), rns) and np.array_equal(ng.cpu().numpy(), rng)
    assert np.abs(np.log10(pv.cpu().numpy()) - np.log10(rpv)).max() < 1e-9
    assert [m[:len(r)] for m, r in zip(masks.cpu().tolist(), rmasks)] == rmasks
