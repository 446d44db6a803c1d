"""Mirror of the pieces of ``wmar.utils.utils`` the generation harness uses
(wmar/utils/utils.py:47-80): uint8 conversion and delta-checkpoint patching."""
from __future__ import annotations

from typing import Union

import numpy as np
import torch
from PIL import Image


def simple_rescale(x):
    return (x + 1.0) / 2.0


# Rescale to [0, 1], clip (!), transpose, rescale to [0, 255], convert to uint8 -> PIL image
def chw_to_pillow(x: Union[torch.Tensor, np.ndarray]) -> Image.Image:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    x = (255 * simple_rescale(x.transpose(1, 2, 0))).clip(0, 255)
    x = np.round(x).astype(np.uint8)  # round-half-even, like the reference
    return Image.fromarray(x)


def update_weights(wrapper, which: str, ckpt_path: str, delta: bool = True):
    """update_weights(model.get_image_tokenizer().encoder|decoder, ckpt) of generate.py:327-332.
    `which` is "encoder" or "decoder"; the deltas are added to the base VQGAN weights and the
    native engine is repacked on next use."""
    assert which in ("encoder", "decoder")
    assert delta, "only delta checkpoints are used on the generation path"
    wrapper.model.apply_delta(which + ".", ckpt_path)
