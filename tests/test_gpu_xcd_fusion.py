"""The fused output projection + residual fold + LN2 statistics launch (k_bx_xr: split-K reduction behind an XCD-local L2 barrier) against
the two-launch path (k_bx + k_resid_stats) it replaces, at production width (1536 x 24 heads, 64 rows).

Both paths fold the same slabs in the same order.  With four waves x 6 steps in phase 1 (WMAR_XR_NW4=1: the arithmetic of k_bx) they
differ only in how the LayerNorm statistics are chunked (16 chunks of 96 columns against 12 of 128), i.e. in the last bits of an fp64 sum:
logits agree to 2e-5 and the sampled tokens are identical.  The shipped launch splits a K slice over EIGHT waves x 3 steps (round 5): a
different association of the same fp32 partial sums -- logits agree to 1e-4 (the depth tests' measure of fp32 rounding noise; logit scale
10 here) and the sampled tokens can differ only where a race is closer than that (rows are independent: at most 2 of 64 rows may leave the
other path's sequence).  The engine's status entry point reports clean flags (no barrier timeout, no block on a foreign XCD)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from wmar_amd.utils import synth  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
cfg = synth.GPTConfig(vocab_size=16384, block_size=256, n_layer=4, n_head=24, n_embd=1536)
eng = GPTEngine(cfg, synth.synth_gpt_state(cfg, seed=3, logit_scale=10.0), max_batch=64)
print("PLAN", eng.plan_info(64)["proj"])
seq = torch.randint(0, cfg.vocab_size, (64, 40), generator=torch.Generator().manual_seed(5)).cuda()
out = [eng.decode_step(seq[:, t], t).cpu().numpy() for t in range(40)]
q = torch.empty(64, 64, cfg.vocab_size).exponential_(1, generator=torch.Generator().manual_seed(7)).cuda()
tok = eng.generate(torch.arange(64).cuda(), 64, q, 1.0, 250, 0.92, None, use_graph=True)
np.savez(sys.argv[1], logits=np.stack(out), tokens=tok.cpu().numpy())
"""


def _run(tmp_path, name, env_extra):
    out = tmp_path / f"{name}.npz"
    env = dict(os.environ, **env_extra)
    res = subprocess.run([sys.executable, "-c", _CHILD % REPO, str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    plan = [l for l in res.stdout.splitlines() if l.startswith("PLAN")][0]
    return np.load(out), plan


def test_fused_projection_matches_the_two_launch_path(tmp_path):
    fused, plan_f = _run(tmp_path, "fused", {})
    plain, plan_p = _run(tmp_path, "plain", {"WMAR_NO_XR": "1"})
    assert "k_bx<1" in plan_p
    # gfx950 in SPX mode: the probe must enable the fused launch -- a regression that silently keeps the two-launch path fails here
    assert "k_bx_xr" in plan_f, "the block -> XCD grouping probe did not enable the fused launch on this device: " + plan_f
    d = np.abs(fused["logits"] - plain["logits"]).max()
    assert d <= 1e-4, d
    rows_off = int((fused["tokens"] != plain["tokens"]).any(axis=1).sum())
    assert rows_off <= 2, rows_off
    # the four-wave variant of the fused launch has k_bx's arithmetic: the round-4 bound holds for it
    fused4, plan_4 = _run(tmp_path, "fused4", {"WMAR_XR_NW4": "1"})
    assert "k_bx_xr" in plan_4
    d4 = np.abs(fused4["logits"] - plain["logits"]).max()
    assert d4 <= 2e-5, d4
    assert np.array_equal(fused4["tokens"], plain["tokens"])


def test_status_entry_point_and_phase_reset():
    from wmar_amd.models.engine import GPTEngine
    from wmar_amd import _lib
    cfg = synth.GPTConfig(vocab_size=16384, block_size=64, n_layer=2, n_head=24, n_embd=1536)
    eng = GPTEngine(cfg, synth.synth_gpt_state(cfg, seed=1, logit_scale=5.0), max_batch=64)
    tok = torch.zeros(64, dtype=torch.int64, device="cuda")
    for t in range(8):
        eng.decode_step(tok, t)
    _lib.check(eng._L.wmar_gpt_check(eng._h, _lib.stream_ptr(eng.device)))        # no timeout, no foreign XCD
    # (-1, -1) returns the attention schedule to the automatic one after a manual setting
    # (16 rows: the automatic schedule of the matrix-core plan is batch-dependent; 1..8 rows run the streaming plan, one attention kernel)
    auto = eng.plan_info(16)["attn"]
    eng.set_attention_phases(0, 0)
    assert eng.plan_info(16)["attn"] != auto
    eng.set_attention_phases(-1, -1)
    assert eng.plan_info(16)["attn"] == auto


_CHILD_FALLBACK = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
mode = sys.argv[2]
cfg = synth.GPTConfig(vocab_size=16384, block_size=256, n_layer=3, n_head=24, n_embd=1536)
eng = GPTEngine(cfg, synth.synth_gpt_state(cfg, seed=3, logit_scale=10.0), max_batch=64)
print("PLAN0", eng.plan_info(64)["proj"])
seq = torch.randint(0, cfg.vocab_size, (64, 6), generator=torch.Generator().manual_seed(5)).cuda()
q = torch.empty(48, 64, cfg.vocab_size).exponential_(1, generator=torch.Generator().manual_seed(7)).cuda()
if mode == "step_first":
    out = [eng.decode_step(seq[:, t], t).cpu().numpy() for t in range(6)]
    tok = eng.generate(torch.arange(64).cuda(), 48, q, 1.0, 250, 0.92, None, use_graph=True)
else:
    tok = eng.generate(torch.arange(64).cuda(), 48, q, 1.0, 250, 0.92, None, use_graph=True)
    out = [eng.decode_step(seq[:, t], t).cpu().numpy() for t in range(6)]
info = eng.plan_info(64)
print("PLAN1", info["proj"])
print("FALLBACKS", info["barrier_fallbacks"])
np.savez(sys.argv[1], logits=np.stack(out), tokens=tok.cpu().numpy())
"""


def _run_fb(tmp_path, name, mode, env_extra):
    out = tmp_path / f"{name}.npz"
    res = subprocess.run([sys.executable, "-c", _CHILD_FALLBACK % REPO, str(out), mode], env=dict(os.environ, **env_extra),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    tags = {l.split()[0]: l.split(None, 1)[1] for l in res.stdout.splitlines() if l.split() and l.split()[0] in ("PLAN0", "PLAN1", "FALLBACKS")}
    return np.load(out), tags


@pytest.mark.parametrize("mode", ["generate_first", "step_first"])
def test_failed_barrier_falls_back_to_the_two_launch_path_and_reruns(tmp_path, mode):
    """WMAR_INJECT_SYNC_FAIL=1 raises the barrier-timeout flag in front of the first fused call: its waits leave at once (garbage
    results), the call notices, switches the engine to k_bx + k_resid_stats and re-runs -- the caller gets the two-launch path's
    tokens / logits, no exception, and the engine stays on that path (plan_info: barrier_fallbacks = 1)."""
    plain, _ = _run_fb(tmp_path, "plain", mode, {"WMAR_NO_XR": "1"})
    hurt, tags = _run_fb(tmp_path, "hurt", mode, {"WMAR_INJECT_SYNC_FAIL": "1"})
    assert "k_bx_xr" in tags["PLAN0"] and "k_bx<1" in tags["PLAN1"] and tags["FALLBACKS"] == "1", tags
    assert np.array_equal(hurt["tokens"], plain["tokens"])
    assert np.array_equal(hurt["logits"], plain["logits"])          # after the fallback the very same launches run
