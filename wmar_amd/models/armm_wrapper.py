"""Mirror of ``wmar.models.armm_wrapper`` (wmar/models/armm_wrapper.py:22-89): the abstract
wrapper API the harness and notebooks talk to.  Only the parts on the generation/detection
path are kept; ``load_model`` (VQGAN loader for finetuning) is out of scope."""
from __future__ import annotations

import torch


class AutoregressiveMultimodalModelWrapper:
    def __init__(self):
        pass

    def set_watermarker(self, watermarker=None):
        raise NotImplementedError("Subclass should implement this, after init")

    def get_image_tokenizer(self):
        raise NotImplementedError("Subclass should implement this")

    def get_vq(self):
        raise NotImplementedError("Subclass should implement this")

    def get_total_vocab_size(self):
        raise NotImplementedError("Subclass should implement this")

    @property
    def device(self):
        return self.model.device

    def init_alivecodes(self, alive_ids_path):
        """armm_wrapper.py:42-55 -- alive ids in FILE order, dead = list(set(range(V)) - set(alive))."""
        vq = self.get_image_tokenizer().quantize
        vocab_sz = vq.n_e if hasattr(vq, "n_e") else vq.num_embeddings
        alive_ids = []
        with open(alive_ids_path, "r") as f:
            for line in f:
                alive_ids.extend(list(map(int, line.split(","))))
        dead_ids = list(set(range(vocab_sz)) - set(alive_ids))
        vq.alive_ids = torch.tensor(alive_ids, dtype=torch.long)
        vq.dead_ids = torch.tensor(dead_ids, dtype=torch.long)

    def sample(self, conditioning, gen_params, apply_watermark=False):
        raise NotImplementedError("Subclass should implement this")

    def codes_to_images(self, codes):
        raise NotImplementedError("Subclass should implement this")

    def images_to_codes(self, images):
        raise NotImplementedError("Subclass should implement this")

    # Shape checkers
    def is_codes_shaped(self, codes):
        return (
            isinstance(codes, torch.Tensor) and codes.ndim == 2 and codes.shape[1] == self.codes_size * self.codes_size
        )

    def is_images_shaped(self, images):
        return (
            isinstance(images, torch.Tensor)
            and images.ndim == 4
            and images.shape[1] == 3
            and images.shape[2] == self.image_size
            and images.shape[3] == self.image_size
        )
