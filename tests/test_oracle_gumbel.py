"""The oracle's Gumbel-key restatement against the reference's own outputs (wmar_audio/watermark/engine.py,
imported by tests/golden/make_golden.py --only gumbel).  CPU only."""
import os

import numpy as np
import pytest

from oracle import wm_oracle as W
from tests.conftest import REPO


@pytest.fixture(scope="module")
def gv():
    return np.load(os.path.join(REPO, "tests", "golden", "gumbel_vectors.npz"))


def test_key_is_torch_rand(gv):
    import torch
    assert np.array_equal(W.gumbel_key(42, 1024), gv["gum_rs_42"])
    for seed in (0, 7, 2 ** 31 - 2, 2 ** 32 + 5):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        assert np.array_equal(W.gumbel_key(seed, 300), torch.rand(300, generator=g).numpy())


@pytest.mark.parametrize("hname", ["same", "rows"])
def test_sample_tokens_equal_reference(gv, hname):
    lg, h = gv["gum_logits"], gv["gum_hash_" + hname]
    for (t, p, k), name in zip(gv["gum_cases"], gv["gum_case_names"]):
        assert np.array_equal(W.gumbel_sample(lg, h, True, t, p, int(k)), gv[f"gum_tok_{hname}_{name}"]), name
    assert np.array_equal(W.gumbel_sample(lg, h, False), gv[f"gum_tok_{hname}_greedy"])


@pytest.mark.parametrize("hname", ["same", "rows"])
def test_score_equal_reference(gv, hname):
    got = W.gumbel_score_tok(gv["gum_score_tokens"], gv["gum_hash_" + hname], 1024)
    assert np.array_equal(got, gv["gum_score_" + hname])


def test_window_hash_host_mirror():
    import torch
    from wmar_amd.watermarking.gumbel_watermark import get_wm_window_hash
    assert get_wm_window_hash(torch.zeros(5, 0, dtype=torch.int64), seed=42).tolist() == [42] * 5
    g = torch.Generator(device="cpu")
    g.manual_seed(9)
    h0 = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g).item())
    ng = torch.tensor([[1, 2], [7, 7], [1023, 0]])
    assert get_wm_window_hash(ng, seed=9).tolist() == [h0 ^ 1 ^ 2, h0, h0 ^ 1023]
