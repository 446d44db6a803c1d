"""Chameleon host pieces against the reference's own outputs (tests/golden/chameleon_vectors.npz, made by importing
deps/chameleon/inference/{logits_processor,token_selector,vocab}.py and the HF warpers).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cham_oracle as CO
from oracle import wm_oracle as W
from tests.conftest import REPO
from tests.golden.make_golden import synth_vocab_map


@pytest.fixture(scope="module")
def cv():
    return np.load(os.path.join(REPO, "tests", "golden", "chameleon_vectors.npz"))


def test_vocab_mirrors(cv):
    from wmar_amd.models.chameleon import VocabInfo, VocabTranslation
    vi = VocabInfo(synth_vocab_map())
    assert vi.image_tokens == cv["cham_image_tokens"].tolist()
    assert vi.text_tokens == cv["cham_text_tokens"].tolist()
    assert vi.special_tokens == cv["cham_special_tokens"].tolist()
    assert [vi.bos_id, vi.eos_id, vi.boi_id, vi.eoi_id, vi.pad_id, vi.eot_id] == cv["cham_ids"].tolist()
    assert (vi.begin_image, vi.end_image, vi.begin_sequence) == (vi.boi_id, vi.eoi_id, vi.bos_id)
    vt = VocabTranslation(vi)
    assert np.array_equal(vt.convert_bpe2img(torch.from_numpy(cv["cham_bpe_batch"])).numpy(), cv["cham_bpe2img"])
    assert np.array_equal(vt.convert_img2bp2(torch.from_numpy(cv["cham_img_batch"])).numpy(), cv["cham_img2bpe"])


@pytest.mark.parametrize("name,seed", [("fixed", "fixed"), ("linear", "linear"), ("nowm", None)])
def test_sampling_chain_equals_reference(cv, name, seed):
    h, delta, temp, top_p = cv[f"cham_{name}_params"]
    alive = cv["cham_image_tokens"]
    V = cv["cham_logits3"].shape[1]
    dead = np.array(sorted(set(range(V)) - set(alive.tolist())), dtype=np.int64)
    key = W.KeyParams(alive, dead, V, 0.25, seed=seed) if seed else None
    tok, lg = CO.sample_step(torch.from_numpy(cv["cham_logits3"]), cv[f"cham_{name}_q"], float(temp), float(top_p), 3.0, 1.2,
                             allow_ids=alive, key=key, past_ids=cv["cham_input_ids"], delta=float(delta))
    ref = cv[f"cham_{name}_processed"] * float(temp)       # the fixture holds logits after temperature and top-p
    kept = np.isfinite(ref)
    np.testing.assert_allclose(lg[kept], ref[kept], rtol=2e-6, atol=2e-6)
    assert np.array_equal(tok, cv[f"cham_{name}_tok"])
