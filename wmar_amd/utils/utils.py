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


def tolerant_torch_load(path):
    """``torch.load(path, map_location="cpu", weights_only=False)`` of the reference, without its installed packages: Lightning
    checkpoints pickle objects of pytorch_lightning / omegaconf; only ``state_dict`` matters, so unknown classes are replaced
    by inert stand-ins while unpickling."""
    import pickle
    from types import SimpleNamespace
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        pass

    class _Stub:
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, s):
            pass

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except Exception:
                return _Stub

    pm = SimpleNamespace(Unpickler=_Unpickler, load=lambda f, **k: _Unpickler(f, **k).load(), __name__="tolerant_pickle")
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=pm)


def update_weights(model, ckpt_path, delta=True, *legacy):  # Deltas!
    """wmar/utils/utils.py:47-66, same signature and semantics: ``model`` is anything with ``state_dict()`` and
    ``load_state_dict(sd, strict=False)`` -- the handles ``get_image_tokenizer().encoder`` / ``.decoder`` return
    (generate.py:327-332) or a torch module.  ``delta=True`` ADDS the checkpoint's tensors to the current ones key by key
    (keys the model lacks are carried along and then ignored by the non-strict load, as in the reference).

    Earlier rounds' form ``update_weights(wrapper, "encoder" | "decoder", ckpt_path[, delta])`` is still accepted."""
    if isinstance(ckpt_path, str) and ckpt_path in ("encoder", "decoder") and hasattr(model, "get_image_tokenizer"):
        which, ckpt_path = ckpt_path, delta
        delta = legacy[0] if legacy else True
        model = getattr(model.get_image_tokenizer(), which)
    state_dict = tolerant_torch_load(ckpt_path)
    if "state_dict" in state_dict:
        state_dict = state_dict["state_dict"]

    if delta:
        state_dict_to_apply = dict(model.state_dict())
        for key in state_dict:
            if key in state_dict_to_apply:
                cur = state_dict_to_apply[key]
                state_dict_to_apply[key] = cur + state_dict[key].to(cur.device)
            else:
                state_dict_to_apply[key] = state_dict[key]
    else:
        state_dict_to_apply = state_dict

    return model.load_state_dict(state_dict_to_apply, strict=False)
