"""Module-shaped handles on the image tokenizer's weights.

The reference patches finetuned encoders / decoders into a loaded model with
``update_weights(model.get_image_tokenizer().encoder, ckpt)`` (generate.py:327-332, wmar/utils/utils.py:47-66), i.e. it
expects ``get_image_tokenizer()`` to return an object with ``.encoder`` / ``.decoder`` / ``.quantize`` sub-modules that
have ``state_dict()`` and ``load_state_dict(sd, strict=False)``.  Here the tokenizer runs inside a native engine that packs
its weights once, so the sub-modules are handles on one shared ``{checkpoint key: tensor}`` dict: ``state_dict()`` returns
the tensors under the handle's prefix (keys relative to it, like ``nn.Module.state_dict``), ``load_state_dict`` replaces
them with torch's rules (shape mismatch raises; ``strict=False`` reports missing / unexpected keys and ignores the
unexpected ones) and drops the packed engine so that it is rebuilt from the new weights on next use."""
from __future__ import annotations

from collections import OrderedDict, namedtuple
from typing import Callable, Dict, Optional

import torch

IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])


class ModuleHandle:
    def __init__(self, state: Dict[str, torch.Tensor], prefix: str, on_change: Callable[[], None]):
        self._state, self._prefix, self._on_change = state, prefix, on_change

    def _keys(self):
        return [k for k in self._state if k.startswith(self._prefix)]

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        n = len(self._prefix)
        return OrderedDict((k[n:], self._state[k]) for k in self._keys())

    def load_state_dict(self, state_dict, strict: bool = True):
        n = len(self._prefix)
        own = self._keys()
        missing = [k[n:] for k in own if k[n:] not in state_dict]
        unexpected = [k for k in state_dict if self._prefix + k not in self._state]
        errors = []
        for k in own:
            if k[n:] in state_dict and tuple(state_dict[k[n:]].shape) != tuple(self._state[k].shape):
                errors.append(f"size mismatch for {k[n:]}: copying a param with shape {tuple(state_dict[k[n:]].shape)} from checkpoint, "
                              f"the shape in current model is {tuple(self._state[k].shape)}.")
        if strict and (missing or unexpected):
            errors.append(f"Missing key(s) in state_dict: {missing}. Unexpected key(s) in state_dict: {unexpected}.")
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(type(self).__name__, "\n\t".join(errors)))
        for k in own:
            if k[n:] in state_dict:
                # in place, as nn.Module.load_state_dict does (param.copy_): tensors taken earlier -- state_dict() values,
                # parameters(), a codebook a watermarker holds -- keep aliasing the updated weights
                cur = self._state[k]
                with torch.no_grad():
                    cur.copy_(state_dict[k[n:]].detach().to(device=cur.device, dtype=cur.dtype))
        self._on_change()
        return IncompatibleKeys(missing, unexpected)

    def parameters(self):
        return iter(self.state_dict().values())

    def named_parameters(self):
        return iter(self.state_dict().items())

    def eval(self):
        return self

    def __repr__(self):
        return f"ModuleHandle({self._prefix!r}, {len(self._keys())} tensors)"


class Quantize:
    """What the watermarker and init_alivecodes need of the quantizer module (taming quantize.py:213-331,
    maskgit_vqgan.py:248-362): the codebook, its size, alive / dead id lists."""

    def __init__(self, state: Dict[str, torch.Tensor], key: str = "quantize.embedding.weight"):
        self._state, self._key = state, key
        self.embedding = _Embedding(state, key)
        self.alive_ids: Optional[torch.Tensor] = None
        self.dead_ids: Optional[torch.Tensor] = None

    @property
    def n_e(self):                       # taming VectorQuantizer2
        return self._state[self._key].shape[0]

    @property
    def e_dim(self):
        return self._state[self._key].shape[1]

    num_embeddings = n_e                 # MaskGIT-VQGAN VectorQuantizer
    embedding_dim = e_dim


class _Embedding:
    def __init__(self, state, key):
        self._state, self._key = state, key

    @property
    def weight(self):
        return self._state[self._key]


class ImageTokenizerHandle(ModuleHandle):
    """Stands where VQModel (taming), PretrainedTokenizer (RAR) and Chameleon's VQModel stand: the whole tokenizer as a
    handle (prefix ''), with ``encoder`` / ``decoder`` / ``quantize`` (+ the 1x1 quant convolutions where the model has them)."""

    def __init__(self, state: Dict[str, torch.Tensor], on_change: Callable[[], None]):
        super().__init__(state, "", on_change)
        self.encoder = ModuleHandle(state, "encoder.", on_change)
        self.decoder = ModuleHandle(state, "decoder.", on_change)
        self.quantize = Quantize(state)
        if any(k.startswith("quant_conv.") for k in state):
            self.quant_conv = ModuleHandle(state, "quant_conv.", on_change)
            self.post_quant_conv = ModuleHandle(state, "post_quant_conv.", on_change)
