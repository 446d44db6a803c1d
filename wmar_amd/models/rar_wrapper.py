"""Drop-in for ``wmar.models.rar_wrapper.RarARMMWrapper`` on MI355X (wmar/models/rar_wrapper.py:17-128).

``sample`` keeps the reference's fixed generation settings (guidance 4.0, ``guidance_scale_pow`` 0,
temperature 1.0; ``gen_params`` are ignored exactly as the reference ignores them, :89-105)."""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from ..utils.synth import MASKGIT_VQ, RAR_XL, MaskgitVQConfig, RARConfig, synth_maskgit_state, synth_rar_state
from .armm_wrapper import AutoregressiveMultimodalModelWrapper
from ..watermarking.gumbel_watermark import GumbelWatermark
from .engine import MaskgitVQEngine, RAREngine
from .tokenizer_handles import ImageTokenizerHandle

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
_RAR_SIZES = {"rar_b": (768, 24, 3072), "rar_l": (1024, 24, 4096), "rar_xl": (1280, 32, 5120), "rar_xxl": (1408, 40, 6144)}


def cfg_scales(steps: int, guidance_scale: float, guidance_scale_pow: float) -> torch.Tensor:
    """Per-step cfg_scale evaluated with the reference's own fp32 tensor arithmetic (rar.py:430-434)."""
    out = []
    for step in range(steps):
        scale_pow = torch.ones((1)) * guidance_scale_pow
        scale_step = (1 - torch.cos(((step / steps) ** scale_pow) * torch.pi)) * 1 / 2
        out.append((guidance_scale - 1) * scale_step + 1)
    return torch.cat(out)


class RarARMMWrapper(AutoregressiveMultimodalModelWrapper):
    def __init__(self, modelpath=None, rar_size="rar_xl", *, rar_cfg: Optional[RARConfig] = None,
                 vq_cfg: Optional[MaskgitVQConfig] = None, rar_state: Optional[Dict[str, torch.Tensor]] = None,
                 vq_state: Optional[Dict[str, torch.Tensor]] = None, device="cuda", max_batch=64):
        super().__init__()
        if modelpath is not None:
            d, L, F = _RAR_SIZES[rar_size]
            rar_cfg = RARConfig(hidden_size=d, num_hidden_layers=L, num_attention_heads=16, intermediate_size=F)
            vq_cfg = MASKGIT_VQ
            vq_state = torch.load(os.path.join(modelpath, "maskgit-vqgan-imagenet-f16-256.bin"), map_location="cpu")
            rar_state = torch.load(os.path.join(modelpath, f"{rar_size}.bin"), map_location="cpu")
        assert rar_cfg is not None and vq_cfg is not None and rar_state is not None and vq_state is not None
        self.rar_size = rar_size
        dev = torch.device(device)
        self.model = SimpleNamespace(device=dev, cfg=rar_cfg, max_batch=max_batch,
                                     engine=RAREngine(rar_cfg, rar_state, max_batch=max_batch, device=dev))
        self._vq_cfg = vq_cfg
        self._vq_state = {k: v.detach().to(dev, torch.float32) for k, v in vq_state.items()}
        self._vq_engine = None
        # PretrainedTokenizer's place (titok.py:24-89): .encoder / .decoder / .quantize as handles on _vq_state
        self.tokenizer = ImageTokenizerHandle(self._vq_state, self._drop_vq_engine)
        ids = os.path.join(_ASSETS, "rar_all_ids.txt")
        if vq_cfg.num_embeddings == 1024 and os.path.exists(ids):
            self.init_alivecodes(ids)
        else:
            vq = self.get_vq()
            vq.alive_ids = torch.arange(vq_cfg.num_embeddings, dtype=torch.long)
            vq.dead_ids = torch.zeros(0, dtype=torch.long)
        self.codes_size = int(math.sqrt(rar_cfg.image_seq_len))
        self.image_size = vq_cfg.resolution
        self.dim_z = vq_cfg.z_channels
        self.watermarker = None
        self.use_graph = True

    @classmethod
    def synthetic(cls, rar_cfg=RAR_XL, vq_cfg=MASKGIT_VQ, seed=0, device="cuda", max_batch=64, logit_scale=20.0):
        rs = synth_rar_state(rar_cfg, seed, device, logit_scale, gen_device=device)
        vs = synth_maskgit_state(vq_cfg, seed, device, gen_device=device)
        return cls(None, rar_cfg=rar_cfg, vq_cfg=vq_cfg, rar_state=rs, vq_state=vs, device=device, max_batch=max_batch)

    def __repr__(self):
        return "RarARMMWrapper"

    def _drop_vq_engine(self):
        self._vq_engine = None  # repacked from _vq_state on next use

    @property
    def vq_engine(self) -> MaskgitVQEngine:
        if self._vq_engine is None:
            self._vq_engine = MaskgitVQEngine(self._vq_cfg, self._vq_state, max_batch=self.model.max_batch,
                                              device=self.model.device)
        return self._vq_engine

    def set_watermarker(self, watermarker=None):
        self.watermarker = watermarker

    def get_image_tokenizer(self):
        return self.tokenizer

    def get_vq(self):
        return self.tokenizer.quantize

    def get_total_vocab_size(self):
        return self.get_vq().num_embeddings

    def draw_noise(self, B: int, generator=None) -> torch.Tensor:
        """The draws of RAR.generate on the default generator, in order: the label-drop mask of
        preprocess_condition (rar.py:305) then one [B,V] Exp(1) tensor per step (multinomial, :454)."""
        cfg = self.model.cfg
        self._noise_draw(lambda t, g: t.uniform_(0, 1, generator=g), (B, 1), generator)   # == torch.rand(B, 1) on that stream
        q = torch.empty(cfg.image_seq_len, B, cfg.codebook_size, dtype=torch.float32, device=self.model.device)
        for n in range(cfg.image_seq_len):
            q[n].copy_(self._noise_draw(lambda t, g: t.exponential_(1, generator=g), (B, cfg.codebook_size), generator))
        return q

    # conditioning: list of size [b] of class indices; gen_params ignored (as in the reference)
    def sample(self, conditioning, gen_params=None, apply_watermark=False, q: Optional[torch.Tensor] = None):
        conditioning = torch.as_tensor(conditioning, device=self.model.device).view(-1)
        cfg = self.model.cfg
        B = conditioning.shape[0]
        scales = cfg_scales(cfg.image_seq_len, 4.0, 0.0)
        out = torch.empty(B, cfg.image_seq_len, dtype=torch.int64, device=self.model.device)
        mb = self.model.max_batch
        if apply_watermark and isinstance(self.watermarker, GumbelWatermark):
            # Gumbel key: the key replaces the sampling noise (extension)
            w = self.watermarker
            for b0 in range(0, B, mb):
                b1 = min(B, b0 + mb)
                out[b0:b1] = self.model.engine.generate_gumbel(conditioning[b0:b1], w.log_rs, scales, w.temperature, w.top_p,
                                                               w.top_k, use_graph=self.use_graph)
            return out.detach()
        wm_ctx = self.watermarker.wm_ctx() if apply_watermark else None
        if q is None and B > mb:
            q = self.draw_noise(B)
        for b0 in range(0, B, mb):
            b1 = min(B, b0 + mb)
            qq = q[:, b0:b1].contiguous() if q is not None else self.draw_noise(b1 - b0)
            out[b0:b1] = self.model.engine.generate(conditioning[b0:b1], qq, scales, 1.0, wm_ctx, use_graph=self.use_graph)
        codes = out.detach()
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        return codes

    def codes_to_images(self, codes):
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        images = self.vq_engine.decode(codes.to(self.model.device))
        assert self.is_images_shaped(images), f"Images shape: {images.shape}"
        return images

    def images_to_codes(self, images):
        assert self.is_images_shaped(images), f"Images shape: {images.shape}"
        codes = self.vq_engine.encode(images.to(self.model.device))
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        return codes
