"""The module-shaped handles `get_image_tokenizer()` returns and `update_weights` with the reference's signature
(wmar/utils/utils.py:47-66, call site generate.py:327-332): the same call applied to a real nn.Module and to a handle
leaves the same weights; torch's load_state_dict rules (non-strict reporting, shape errors) hold.  Host logic only."""
import os

import pytest
import torch
import torch.nn as nn

from wmar_amd.models.tokenizer_handles import ImageTokenizerHandle
from wmar_amd.utils.utils import update_weights


class _Tok(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = nn.Sequential(nn.Conv2d(3, 4, 3), nn.Conv2d(4, 2, 1))
        self.decoder = nn.Sequential(nn.Conv2d(2, 4, 3), nn.Conv2d(4, 3, 1))
        self.quantize = nn.Module()
        self.quantize.embedding = nn.Embedding(16, 2)
        self.quant_conv = nn.Conv2d(2, 2, 1)
        self.post_quant_conv = nn.Conv2d(2, 2, 1)


def _pair():
    torch.manual_seed(0)
    m = _Tok()
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    drops = []
    return m, state, ImageTokenizerHandle(state, lambda: drops.append(1)), drops


@pytest.mark.parametrize("nested", [False, True])
def test_delta_update_matches_a_real_module(tmp_path, nested):
    m, state, h, drops = _pair()
    torch.manual_seed(1)
    delta = {"0.weight": 0.1 * torch.randn(4, 3, 3, 3), "1.bias": torch.randn(2), "not.there": torch.ones(1)}
    path = str(tmp_path / "enc_delta.pth")
    torch.save({"state_dict": delta} if nested else delta, path)
    r_mod = update_weights(m.encoder, path)                    # the reference's call, on a torch module
    r_h = update_weights(h.encoder, path)                      # ... and on the handle
    assert list(r_mod.unexpected_keys) == list(r_h.unexpected_keys) == ["not.there"]
    assert list(r_mod.missing_keys) == list(r_h.missing_keys) == []
    for k, v in m.state_dict().items():
        assert torch.equal(v, state[k]), k
    assert "encoder.not.there" not in state and len(drops) == 1   # unknown keys are ignored; the packed engine is dropped once


def test_full_checkpoint_and_legacy_form(tmp_path):
    m, state, h, drops = _pair()
    new = {k: torch.full_like(v, 0.25) for k, v in m.decoder.state_dict().items()}
    path = str(tmp_path / "dec.pth")
    torch.save(new, path)
    update_weights(m.decoder, path, delta=False)
    update_weights(h.decoder, path, False)
    for k, v in m.state_dict().items():
        assert torch.equal(v, state[k]), k

    class Wrapper:
        def get_image_tokenizer(self):
            return h

    before = state["decoder.0.bias"].clone()
    torch.save({"0.bias": torch.ones(4)}, path)
    update_weights(Wrapper(), "decoder", path)                 # rounds 1-4 form
    assert torch.equal(state["decoder.0.bias"], before + 1)


def test_load_state_dict_rules():
    m, state, h, _ = _pair()
    sd = m.encoder.state_dict()
    sd.pop("1.bias")
    r = h.encoder.load_state_dict(sd, strict=False)
    assert r.missing_keys == ["1.bias"] and r.unexpected_keys == []
    with pytest.raises(RuntimeError, match="Missing key"):
        h.encoder.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError, match="size mismatch"):
        h.encoder.load_state_dict({"0.weight": torch.zeros(1)}, strict=False)
    assert set(h.state_dict()) == set(m.state_dict())
    assert h.quantize.n_e == h.quantize.num_embeddings == 16 and h.quantize.e_dim == 2
    assert h.quantize.embedding.weight is state["quantize.embedding.weight"]
    assert hasattr(h, "quant_conv") and list(h.post_quant_conv.state_dict()) == ["weight", "bias"]


def test_load_state_dict_updates_in_place_like_nn_module():
    """tensors taken before a load keep aliasing the weights (nn.Module.load_state_dict copies into the parameters): state_dict()
    values, parameters(), a codebook a watermarker holds"""
    import torch
    from wmar_amd.models.tokenizer_handles import ModuleHandle
    state = {"decoder.conv.weight": torch.zeros(2, 3), "decoder.conv.bias": torch.zeros(2), "quantize.embedding.weight": torch.zeros(4, 2)}
    calls = []
    h = ModuleHandle(state, "decoder.", lambda: calls.append(1))
    before = h.state_dict()["conv.weight"]
    held = next(iter(h.parameters()))
    h.load_state_dict({"conv.weight": torch.ones(2, 3), "conv.bias": torch.full((2,), 2.0)})
    assert calls == [1]
    assert torch.equal(before, torch.ones(2, 3)) and before.data_ptr() == state["decoder.conv.weight"].data_ptr()
    assert torch.equal(held, torch.ones(2, 3))
    assert torch.equal(state["decoder.conv.bias"], torch.full((2,), 2.0)) and torch.equal(state["quantize.embedding.weight"], torch.zeros(4, 2))
