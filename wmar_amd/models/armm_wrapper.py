"""The wrapper contract the harness and notebooks talk to -- same method names, arguments and shape rules as
``wmar.models.armm_wrapper.AutoregressiveMultimodalModelWrapper`` (wmar/models/armm_wrapper.py:22-89).  ``load_model``
(a VQGAN loader for finetuning) is not part of the generation/detection path."""
from __future__ import annotations

import torch


def _read_id_list(path):
    """Comma-separated ids, possibly over several lines, in FILE order."""
    ids = []
    with open(path, "r") as f:
        for line in f:
            ids += [int(tok) for tok in line.split(",") if tok.strip()]
    return ids


class AutoregressiveMultimodalModelWrapper:
    """Subclasses set ``model`` (with a ``device``), ``codes_size``, ``image_size`` and implement the abstract methods."""

    _ABSTRACT = "Subclass should implement this"

    def __init__(self):
        pass

    # ---- what a model wrapper must provide
    def set_watermarker(self, watermarker=None):
        raise NotImplementedError(self._ABSTRACT + ", after init")

    def get_image_tokenizer(self):
        raise NotImplementedError(self._ABSTRACT)

    def get_vq(self):
        raise NotImplementedError(self._ABSTRACT)

    def get_total_vocab_size(self):
        raise NotImplementedError(self._ABSTRACT)

    def sample(self, conditioning, gen_params, apply_watermark=False):
        raise NotImplementedError(self._ABSTRACT)

    def codes_to_images(self, codes):
        raise NotImplementedError(self._ABSTRACT)

    def images_to_codes(self, images):
        raise NotImplementedError(self._ABSTRACT)

    @property
    def device(self):
        return self.model.device

    # Where the Exp(1) noise of ``torch.multinomial`` is drawn.  None: the model's device (what the reference does when the
    # model lives on the GPU).  "cpu": the CPU default generator, copied to the device -- reproduces a reference run whose
    # model (and therefore whose ``multinomial`` call) was on the CPU, e.g. the committed golden fixtures.
    noise_device = None

    def _noise_draw(self, fn, shape, generator=None):
        """`fn` fills a fresh float32 tensor of `shape` in place on the noise device; returns it on the model's device."""
        dev = torch.device(self.noise_device) if self.noise_device is not None else torch.device(self.model.device)
        t = torch.empty(shape, dtype=torch.float32, device=dev)
        fn(t, generator)
        return t if dev == torch.device(self.model.device) else t.to(self.model.device)

    def init_alivecodes(self, alive_ids_path):
        """Attach ``alive_ids`` (file order) and ``dead_ids`` to the quantizer.  The dead list is
        ``list(set(range(V)) - set(alive))`` exactly as armm_wrapper.py:42-55 builds it: its ORDER feeds the key derivation."""
        vq = self.get_image_tokenizer().quantize
        vocab = getattr(vq, "n_e", None)
        if vocab is None:
            vocab = vq.num_embeddings
        alive = _read_id_list(alive_ids_path)
        dead = list(set(range(vocab)) - set(alive))
        vq.alive_ids = torch.tensor(alive, dtype=torch.long)
        vq.dead_ids = torch.tensor(dead, dtype=torch.long)

    # ---- shape rules
    def is_codes_shaped(self, codes):
        n = self.codes_size * self.codes_size
        return isinstance(codes, torch.Tensor) and codes.ndim == 2 and codes.shape[1] == n

    def is_images_shaped(self, images):
        if not isinstance(images, torch.Tensor) or images.ndim != 4:
            return False
        return tuple(images.shape[1:]) == (3, self.image_size, self.image_size)
