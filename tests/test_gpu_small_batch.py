"""The small-batch decode path (1..12 rows: wmar_amd/csrc/decode_small.h -- weight-streaming kernels on row-major weights, five
launches per layer) against the REFERENCE's outputs at production width (tests/golden/prod_vectors.npz, made by
tests/golden/make_golden.py from mingpt.py:125-214) and against the matrix-core plan of the same engine build.

  * every batch size 1..12, teacher-forced through all 256 positions: logits at the fixture's 12 positions within 5e-4 of the
    reference's, the arg-max of every row at every position equal (rows are independent: rows [:B] of the 64-row fixture);
  * the reference's batch size (configs/taming_generate.json: 5): a 256-step watermarked sampling loop, graph and eager, gives
    the same tokens as the matrix-core plan (WMAR_NO_SMALL=1 engine) on the same noise -- and the 4-row fixture loop of
    test_gpu_prod_shapes.py (the reference's own tokens) runs on this path too;
  * the engine says which plan it runs (plan_info)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.conftest import REPO  # noqa: E402
from tests.test_gpu_watermark import _wm  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402

GCFG = synth.GPTConfig(vocab_size=16384, block_size=256, n_layer=2, n_head=24, n_embd=1536)
ATOL = 5e-4


@pytest.fixture(scope="module")
def pv():
    return np.load(os.path.join(REPO, "tests", "golden", "prod_vectors.npz"))


@pytest.fixture(scope="module")
def gpt():
    from wmar_amd.models.engine import GPTEngine
    return GPTEngine(GCFG, synth.synth_gpt_state(GCFG, seed=9, logit_scale=10.0), max_batch=12)


@pytest.fixture(scope="module")
def gpt_mfma():
    from wmar_amd.models.engine import GPTEngine
    os.environ["WMAR_NO_SMALL"] = "1"
    try:
        return GPTEngine(GCFG, synth.synth_gpt_state(GCFG, seed=9, logit_scale=10.0), max_batch=12)
    finally:
        del os.environ["WMAR_NO_SMALL"]


def test_plan_info_names_the_small_path(gpt, gpt_mfma):
    assert "k_sgemv" in gpt.plan_info(5)["qkv"] and gpt.plan_info(5)["resid"] == "none"
    assert "k_sgemv" not in gpt_mfma.plan_info(5)["qkv"]


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_small_batch_teacher_forced_256_positions(pv, gpt, B):
    seq = torch.from_numpy(pv["gpt_seq"].astype(np.int64))[:B].cuda()
    want = {int(p): i for i, p in enumerate(pv["gpt_pos"])}
    worst, flips = 0.0, 0
    for t in range(256):
        lg = gpt.decode_step(seq[:, t], t)
        am = lg.argmax(-1).cpu().numpy()
        ref_am = pv["gpt_argmax"][t].astype(np.int64)[:B]
        if not np.array_equal(am, ref_am):      # an arg-max may only move between two logits closer than the tolerance
            l = lg.cpu().numpy()
            for b in np.nonzero(am != ref_am)[0]:
                assert abs(l[b, am[b]] - l[b, ref_am[b]]) < 2 * ATOL, (t, b)
                flips += 1
        if t in want:
            d = np.abs(lg[:, ::64].cpu().numpy() - pv["gpt_logits"][want[t]][:B]).max()
            worst = max(worst, float(d))
            assert d < ATOL, (B, t, d)
    assert flips <= 1
    print(f"{B} rows: max |dlogit| over {len(want)} positions: {worst:.2e}; arg-max near-tie flips: {flips}")


@pytest.mark.parametrize("graph", [True, False])
def test_batch_5_watermarked_loop_equals_matrix_core_plan(kat, gpt, gpt_mfma, graph):
    wm = _wm(kat["keys"]["taming"])
    B, steps = 5, 256
    g = torch.Generator(device="cuda").manual_seed(55)
    q = torch.empty(steps, B, 16384, device="cuda").exponential_(1, generator=g)
    cond = torch.tensor([1, 9, 232, 340, 568]).cuda()          # the first classes of configs/taming_generate.json
    a, la = gpt.generate(cond, steps, q, 1.0, 250, 0.92, wm.wm_ctx(), use_graph=graph, trace_logits=True)
    b, lb = gpt_mfma.generate(cond, steps, q, 1.0, 250, 0.92, wm.wm_ctx(), use_graph=graph, trace_logits=True)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    la, lb = la.cpu().numpy(), lb.cpu().numpy()
    # the two plans sum in different orders: tokens equal unless a race is closer than the logit tolerance
    for r in range(B):
        bad = np.nonzero(a[r] != b[r])[0]
        n = int(bad[0]) if bad.size else steps
        assert np.abs(la[:n + 1, r] - lb[:n + 1, r]).max() < ATOL
        assert bad.size == 0, f"row {r} diverges at step {n}"
    pa, pb = wm.detect(torch.from_numpy(a).cuda()).cpu().numpy(), wm.detect(torch.from_numpy(b).cuda()).cpu().numpy()
    assert np.array_equal(pa, pb)


@pytest.fixture(scope="module")
def gpt_persist():
    """the opt-in persistent step (wmar_amd/csrc/decode_persist.h: one launch per decode step, device-wide barriers between its
    phases; measured slower than the five-launch plan and therefore not the default -- DESIGN section 6a)"""
    from wmar_amd.models.engine import GPTEngine
    os.environ["WMAR_PERSIST"] = "1"
    try:
        return GPTEngine(GCFG, synth.synth_gpt_state(GCFG, seed=9, logit_scale=10.0), max_batch=12)
    finally:
        del os.environ["WMAR_PERSIST"]


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5])
def test_persistent_step_matches_the_reference_fixture_and_the_launch_plan(pv, kat, gpt, gpt_persist, B):
    info = gpt_persist.plan_info(B)
    if "persistent" not in info.get("path", ""):
        pytest.skip("the persistent step is not available on this device (workgroups not co-resident / XCD grouping): " + info.get("path", ""))
    assert "persistent" not in gpt.plan_info(B).get("path", "") and "k_sgemv" in gpt_persist.plan_info(6)["qkv"] and "k_sgemv" in gpt.plan_info(12)["qkv"]
    seq = torch.from_numpy(pv["gpt_seq"].astype(np.int64))[:B].cuda()
    want = {int(p): i for i, p in enumerate(pv["gpt_pos"])}
    worst = 0.0
    for t in range(256):
        lg = gpt_persist.decode_step(seq[:, t], t)
        assert np.array_equal(lg.argmax(-1).cpu().numpy(), pv["gpt_argmax"][t].astype(np.int64)[:B]), t
        if t in want:
            d = float(np.abs(lg[:, ::64].cpu().numpy() - pv["gpt_logits"][want[t]][:B]).max())
            worst = max(worst, d)
            assert d < ATOL, (B, t, d)
    # the captured 256-step loop: same tokens as the five-launch plan on the same noise
    wm = _wm(kat["keys"]["taming"])
    g = torch.Generator(device="cuda").manual_seed(77 + B)
    q = torch.empty(256, B, 16384, device="cuda").exponential_(1, generator=g)
    cond = torch.tensor([1, 9, 232, 340, 568][:B]).cuda()
    a = gpt_persist.generate(cond, 256, q, 1.0, 250, 0.92, wm.wm_ctx(), use_graph=True)
    b = gpt.generate(cond, 256, q, 1.0, 250, 0.92, wm.wm_ctx(), use_graph=True)
    assert torch.equal(a, b)
    assert gpt_persist.plan_info(B)["barrier_fallbacks"] == "0"
    print(f"persistent step, {B} rows: max |dlogit| {worst:.2e} over {len(want)} positions; 256-step loop equal to the launch plan")
