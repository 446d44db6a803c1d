"""Drop-in for ``wmar.models.taming_wrapper.TamingARMMWrapper`` on MI355X.

Same public surface (wmar/models/taming_wrapper.py:22-92): ``sample``, ``codes_to_images``,
``images_to_codes``, ``set_watermarker``, ``get_vq``, ``get_image_tokenizer``,
``get_total_vocab_size``, ``device``, ``codes_size / image_size / dim_z``.  The bodies call the
native engines of libwmar_hip.so; PyTorch only stores tensors, parses the checkpoint and
draws the Exp(1) noise ``torch.multinomial`` would draw.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from ..utils.utils import tolerant_torch_load as _tolerant_torch_load
from ..utils.synth import GPTConfig, VQConfig, synth_gpt_state, synth_gpt_state_fast, synth_vq_state, synth_vq_state_fast
from .armm_wrapper import AutoregressiveMultimodalModelWrapper
from .engine import GPTEngine, VQGANEngine
from .tokenizer_handles import ImageTokenizerHandle

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def configs_from_yaml(config_path):
    """Model dims from the checkpoint's ``configs/net2net.yaml`` (plain PyYAML, no OmegaConf)."""
    import yaml

    with open(config_path) as f:
        cfg = yaml.safe_load(f)
    p = cfg["model"]["params"]
    t = p["transformer_config"]["params"]
    g = GPTConfig(vocab_size=t["vocab_size"], block_size=t["block_size"], n_layer=t["n_layer"], n_head=t["n_head"],
                  n_embd=t["n_embd"])
    fs = p["first_stage_config"]["params"]
    dd = fs["ddconfig"]
    v = VQConfig(ch=dd["ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
                 attn_resolutions=tuple(dd["attn_resolutions"]), resolution=dd["resolution"],
                 in_channels=dd["in_channels"], out_ch=dd["out_ch"], z_channels=dd["z_channels"],
                 embed_dim=fs["embed_dim"], n_embed=fs["n_embed"])
    return g, v


class _Net2Net:
    """Stands where ``Net2NetTransformer`` stands in the reference wrapper (``self.model``)."""

    def __init__(self, gpt_cfg: GPTConfig, vq_cfg: VQConfig, gpt_state: Dict[str, torch.Tensor],
                 vq_state: Dict[str, torch.Tensor], device, max_batch: int):
        self.device = torch.device(device)
        self.gpt_cfg, self.vq_cfg = gpt_cfg, vq_cfg
        self.max_batch = max_batch
        self.vq_state = {k: v.detach().to(self.device, torch.float32) for k, v in vq_state.items()
                         if not k.startswith("loss.")}
        self.transformer = GPTEngine(gpt_cfg, gpt_state, max_batch=max_batch, device=self.device)
        self._vq_engine: Optional[VQGANEngine] = None
        # VQModel's place: .encoder / .decoder / .quantize (+ quant convs) as state_dict()/load_state_dict() handles on vq_state
        self.first_stage_model = ImageTokenizerHandle(self.vq_state, self._drop_vq_engine)

    def _drop_vq_engine(self):
        self._vq_engine = None  # repacked from vq_state on next use

    @property
    def vq_engine(self) -> VQGANEngine:
        if self._vq_engine is None:
            self._vq_engine = VQGANEngine(self.vq_cfg, self.vq_state, max_batch=self.max_batch, device=self.device)
        return self._vq_engine

    def apply_delta(self, prefix: str, ckpt_path: str):
        """update_weights(model.<prefix>, ckpt) with delta=True (wmar/utils/utils.py:47-66):
        ``*_delta.pth`` tensors are ADDED to the base weights key-wise."""
        from ..utils.utils import update_weights
        update_weights(getattr(self.first_stage_model, prefix.rstrip(".")), ckpt_path, delta=True)


class TamingARMMWrapper(AutoregressiveMultimodalModelWrapper):
    def __init__(self, modelpath=None, *, gpt_cfg=None, vq_cfg=None, gpt_state=None, vq_state=None, device="cuda",
                 max_batch=64):
        super().__init__()
        if modelpath is not None:
            # NOTE: make sure you download the models first (see the reference README)
            config_path = os.path.join(modelpath, "configs/net2net.yaml")
            ckpt_path = os.path.join(modelpath, "checkpoints/net2net.ckpt")
            gpt_cfg, vq_cfg = configs_from_yaml(config_path)
            sd = _tolerant_torch_load(ckpt_path)
            if "state_dict" in sd:
                sd = sd["state_dict"]
            gpt_state = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
            vq_state = {k[len("first_stage_model."):]: v for k, v in sd.items() if k.startswith("first_stage_model.")}
        assert gpt_cfg is not None and vq_cfg is not None and gpt_state is not None and vq_state is not None
        self.model = _Net2Net(gpt_cfg, vq_cfg, gpt_state, vq_state, device, max_batch)
        alive = os.path.join(_ASSETS, "vqgan_alive_ids.txt")
        if vq_cfg.n_embed == 16384:
            self.init_alivecodes(alive if os.path.exists(alive) else "assets/vqgan_alive_ids.txt")
        else:  # reduced test configs: every code alive
            vq = self.get_vq()
            vq.alive_ids = torch.arange(vq_cfg.n_embed, dtype=torch.long)
            vq.dead_ids = torch.zeros(0, dtype=torch.long)
        self.codes_size = vq_cfg.codes_size
        self.image_size = vq_cfg.resolution
        self.dim_z = vq_cfg.embed_dim
        self.watermarker = None
        self.use_graph = True

    @classmethod
    def synthetic(cls, gpt_cfg: GPTConfig, vq_cfg: VQConfig, seed=0, device="cuda", max_batch=64, logit_scale=30.0,
                  fast=True):
        """Random-init weights of the given architecture (no checkpoints exist offline)."""
        if fast:
            gs = synth_gpt_state_fast(gpt_cfg, seed, device, logit_scale)
            vs = synth_vq_state_fast(vq_cfg, seed, device)
        else:
            gs = synth_gpt_state(gpt_cfg, seed, "cpu", logit_scale)
            vs = synth_vq_state(vq_cfg, seed, "cpu")
        return cls(None, gpt_cfg=gpt_cfg, vq_cfg=vq_cfg, gpt_state=gs, vq_state=vs, device=device, max_batch=max_batch)

    def __repr__(self):
        return "TamingARMMWrapper"

    def set_watermarker(self, watermarker=None):
        self.watermarker = watermarker

    def get_image_tokenizer(self):
        return self.model.first_stage_model

    def get_vq(self):
        return self.get_image_tokenizer().quantize

    def get_total_vocab_size(self):
        return self.get_vq().n_e

    def draw_noise(self, steps: int, B: int, generator=None) -> torch.Tensor:
        """The noise of the reference's ``torch.multinomial(probs, 1)`` calls: one
        ``empty(B, V).exponential_(1)`` per decode step, in step order, from the device's
        default generator (mingpt.py:363) -- drawn up front so the loop is one hipGraph."""
        V = self.model.gpt_cfg.vocab_size
        q = torch.empty(steps, B, V, dtype=torch.float32, device=self.model.device)
        for n in range(steps):
            q[n].copy_(self._noise_draw(lambda t, g: t.exponential_(1, generator=g), (B, V), generator))
        return q

    # conditioning: list of size [b]; gen_params: dict; returns detached codes [b, codes_size**2]
    def sample(self, conditioning, gen_params, apply_watermark=False, q: Optional[torch.Tensor] = None):
        conditioning = torch.as_tensor(conditioning, device=self.model.device).view(-1)
        steps = self.codes_size * self.codes_size
        B = conditioning.shape[0]
        wm_ctx = self.watermarker.wm_ctx() if apply_watermark else None
        out = torch.empty(B, steps, dtype=torch.int64, device=self.model.device)
        mb = self.model.max_batch
        if q is None and B > mb and steps * B * self.model.gpt_cfg.vocab_size * 4 <= (32 << 30):
            q = self.draw_noise(steps, B)  # one [B,V] draw per step for the WHOLE batch, as the reference does
        for b0 in range(0, B, mb):
            b1 = min(B, b0 + mb)
            qq = q[:, b0:b1].contiguous() if q is not None else self.draw_noise(steps, b1 - b0)
            out[b0:b1] = self.model.transformer.generate(
                conditioning[b0:b1], steps, qq, temperature=gen_params["temperature"], top_k=gen_params["top_k"],
                top_p=gen_params["top_p"], wm_ctx=wm_ctx, use_graph=self.use_graph)
        codes = out.detach()
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        return codes

    # codes: [b, codes_size**2] tokens -> [b, 3, image_size, image_size] pixels in [-1, 1]
    def codes_to_images(self, codes):
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        images = self.model.vq_engine.decode(codes.to(self.model.device))
        assert self.is_images_shaped(images), f"Images shape: {images.shape}"
        return images

    # images: [b, 3, image_size, image_size] pixels in [-1, 1] -> [b, codes_size**2] tokens
    def images_to_codes(self, images):
        assert self.is_images_shaped(images), f"Images shape: {images.shape}"
        codes = self.model.vq_engine.encode(images.to(self.model.device))
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        return codes
