"""The 8-bit PIL round trip inside the reference's Chameleon ``images_to_codes`` (wmar/models/chameleon_wrapper.py:177-181 ->
deps/chameleon/inference/image_tokenizer.py:100-122 `_pil_from_chw_tensor`, :74-93 `_vqgan_input_from`) against this build's
tensor-only form.  image_tokenizer.py cannot be imported under this Python (it annotates with the PIL.Image MODULE inside
typing.Union), so the two functions' published steps are carried out here with PIL itself."""
import numpy as np
import pytest
import torch
from PIL import Image


def _pil_path(chw: torch.Tensor) -> torch.Tensor:
    # _pil_from_chw_tensor
    n = (torch.clamp(chw.detach().cpu(), -1.0, 1.0) + 1.0) / 2.0
    u8 = (n.permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    img = Image.fromarray(u8)
    if img.mode != "RGB":
        img = img.convert("RGB")
    # _vqgan_input_from(target 512)
    s = min(img.size)
    scale = chw.shape[-1] / s
    new_size = (round(scale * img.size[0]), round(scale * img.size[1]))
    img = img.resize(new_size, Image.LANCZOS)
    T = chw.shape[-1]
    x0, y0 = (img.width - T) // 2, (img.height - T) // 2
    img = img.crop((x0, y0, x0 + T, y0 + T))
    a = np.array(img) / 255.0
    a = a * 2 - 1
    return torch.from_numpy(a).permute(2, 0, 1).float()


@pytest.mark.parametrize("size", [64, 512])
def test_eight_bit_round_trip_equals_pil_path(size):
    from wmar_amd.models.chameleon_wrapper import eight_bit_round_trip
    g = torch.Generator().manual_seed(size)
    x = torch.rand(2, 3, size, size, generator=g) * 2.6 - 1.3            # beyond [-1, 1]: the clamp matters
    ticks = torch.arange(256, dtype=torch.float64) / 255.0 * 2 - 1       # values that sit exactly on an 8-bit level
    x[0, 0, 0, :min(256, size)] = ticks[:min(256, size)].float()
    x[0, 1, 1, :min(256, size)] = (ticks[:min(256, size)] + 1e-7).float()
    got = eight_bit_round_trip(x)
    for b in range(2):
        assert torch.equal(got[b], _pil_path(x[b]))


@pytest.mark.gpu
def test_eight_bit_round_trip_on_device_equals_host():
    from wmar_amd.models.chameleon_wrapper import eight_bit_round_trip
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 128, 128, generator=g) * 2.6 - 1.3
    assert torch.equal(eight_bit_round_trip(x.cuda()).cpu(), eight_bit_round_trip(x))
