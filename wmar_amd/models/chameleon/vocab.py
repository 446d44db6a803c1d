"""Vocabulary bookkeeping of Chameleon: mirrors ``deps/chameleon/inference/vocab.py`` (VocabInfo :11-74,
VocabTranslation :77-122) -- same attribute names and results; the translations are device gathers."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


class VocabInfo:
    """Token classes of the BPE vocabulary (``vocab_map``: the tokenizer json's name -> id table)."""

    _SPECIAL = {"bos_id": "<s>", "eos_id": "</s>", "boi_id": "<racm3:break>", "eoi_id": "<eoss>", "pad_id": "<pad>",
                "eot_id": "<reserved08706>"}

    def __init__(self, vocab_map: Dict[str, int]):
        self.name2val = vocab_map
        for attr, name in self._SPECIAL.items():
            setattr(self, attr, vocab_map.get(name))
        self.val2name = {v: k for k, v in vocab_map.items()}
        self.all_tokens: List[int] = sorted(vocab_map.values())
        self.image_tokens: List[int] = sorted(v for k, v in vocab_map.items() if k.startswith("IMGIMG"))
        self.special_tokens: List[int] = sorted(v for k, v in vocab_map.items() if k.startswith("<") and k != "<")
        img, spc = set(self.image_tokens), set(self.special_tokens)
        self.text_tokens: List[int] = [t for t in self.all_tokens if t not in img and t not in spc]

    begin_sequence = property(lambda self: self.bos_id)
    end_sequence = property(lambda self: self.eos_id)
    begin_image = property(lambda self: self.boi_id)
    end_image = property(lambda self: self.eoi_id)
    padding = property(lambda self: self.pad_id)
    end_turn = property(lambda self: self.eot_id)


class VocabTranslation:
    """BPE id <-> VQGAN code.  An image token is named ``IMGIMG<letters>Z`` where the letters A..J spell the code's decimal
    digits (vocab.py:82-93)."""

    def __init__(self, vocab_info: VocabInfo, device: Optional[str] = None):
        self._vocab = vocab_info
        self._device = device
        self.bpe2img: Dict[int, int] = {}
        for tok in vocab_info.image_tokens:
            name = vocab_info.val2name[tok][len("IMGIMG"):-1]
            self.bpe2img[tok] = int("".join(str(ord(c) - ord("A")) if "A" <= c <= "J" else c for c in name))
        self.img2bpe: Dict[int, int] = {v: k for k, v in self.bpe2img.items()}
        # the reference pairs sorted(bpe ids) with sorted(codes) (vocab.py:100-104): rank-to-rank, which coincides with the
        # name-derived mapping whenever codes grow with ids (true for the released tokenizer)
        self.bpe2img_search_tensors = (torch.tensor(sorted(self.bpe2img.keys()), device=device),
                                       torch.tensor(sorted(self.bpe2img.values()), device=device))
        table = torch.zeros(max(self.img2bpe.keys()) + 1, dtype=torch.int, device=device)
        for k, v in self.img2bpe.items():
            table[k] = v
        self.img2bpe_mapping_tensor = table

    def convert_bpe2img(self, bpe_batch: torch.Tensor) -> torch.Tensor:
        bpe_tok, img_tok = self.bpe2img_search_tensors
        bpe_tok, img_tok = bpe_tok.to(bpe_batch.device), img_tok.to(bpe_batch.device)
        return img_tok[torch.searchsorted(bpe_tok, bpe_batch)]

    def convert_img2bp2(self, img_batch: torch.Tensor) -> torch.Tensor:
        return self.img2bpe_mapping_tensor.to(img_batch.device)[img_batch]
