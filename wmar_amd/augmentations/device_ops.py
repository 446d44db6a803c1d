"""The evaluation transforms as kernels of this build (``wmar_augment``, wmar_amd/csrc/augment.hip): what the transform modules run
for tensors on the MI355X, and the harness's fused form -- one launch that reads the decoder's [-1, 1] batch, transforms it in
[0, 1], clamps and writes [-1, 1] for the encoder (generate.py:146-150 around every (transform, parameter) pair; ~90 pairs per
image).  CPU tensors keep the torch restatements in valuemetric.py / geometric.py (host-side utilities, as in the reference)."""
from __future__ import annotations

import os

import torch

from .. import _lib

IDENTITY, BLUR, NOISE, BRIGHTNESS, ROTATE, FLIP_H, CROP_RESIZE, CROP_PAD = range(8)


def eligible(image: torch.Tensor) -> bool:
    if os.environ.get("WMAR_AUG_TORCH"):        # A/B knob: the torch restatements on the device instead of the kernels
        return False
    return image.is_cuda and image.dtype == torch.float32 and image.dim() in (3, 4)


def run(op: int, image: torch.Tensor, p0: float = 0.0, p1: float = 0.0, noise: torch.Tensor = None, pm1: bool = False) -> torch.Tensor:
    x = image.unsqueeze(0) if image.dim() == 3 else image
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    if noise is not None:
        noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
        assert noise.numel() == x.numel()
    L = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(L.wmar_augment(int(op), x.data_ptr(), out.data_ptr(), noise.data_ptr() if noise is not None else None, B, C, H, W,
                                  1 if pm1 else 0, float(p0), float(p1), _lib.stream_ptr(x.device)))
    return out[0] if image.dim() == 3 else out


def _rotation(angle):
    quarters, rest = divmod(angle, 90)          # floor division: -20 -> (-1, 70), as Rotate.forward
    return quarters % 4, rest


def fused(name: str, imgs_pm1: torch.Tensor, param):
    """`aug(imgs / 2 + 0.5, param).clamp(0, 1) * 2 - 1` of the AugmentationManager table entry `name` as ONE launch, or None when the
    entry has no device form (jpeg: host PIL) or the tensor is not eligible.  Bit-identical to the unfused sequence
    (tests/test_gpu_augment_kernels.py)."""
    if not eligible(imgs_pm1) or imgs_pm1.dim() != 4:
        return None
    H, W = imgs_pm1.shape[-2:]
    if name == "gaussian-blur":
        return run(IDENTITY if param == 0 else BLUR, imgs_pm1, param, pm1=True)
    if name == "gaussian-noise":
        return run(NOISE, imgs_pm1, param, noise=torch.randn_like(imgs_pm1), pm1=True)      # one randn draw, as GaussianNoise.forward
    if name == "brightness":
        return run(BRIGHTNESS, imgs_pm1, param, pm1=True)
    if name == "rotation":
        q, rest = _rotation(param)
        if q % 2 and H != W:
            return None
        return run(ROTATE, imgs_pm1, q, rest, pm1=True)
    if name == "flip-h":
        return run(FLIP_H if param else IDENTITY, imgs_pm1, pm1=True)
    if name == "upperleft-crop":
        oh, ow = int(param * H), int(param * W)
        return run(IDENTITY if (oh, ow) == (H, W) else CROP_RESIZE, imgs_pm1, oh, ow, pm1=True)
    return None
