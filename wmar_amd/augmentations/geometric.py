"""Geometric transforms on [B, 3, H, W] (or [3, H, W]) tensors in [0, 1], any device; API of
``wmar/augmentations/geometric.py`` (:22-117): ``Rotate``, ``UpperLeftCrop``, ``UpperLeftCropWithResizeBack``,
``UpperLeftCropWithPadBack``, ``HorizontalFlip``, ``Identity`` modules called as ``T()(image, parameter)``.

The reference delegates to ``torchvision.transforms.functional`` (absent offline); its tensor code paths are restated with torch
primitives: ``rotate`` = inverse affine map sampled with ``grid_sample`` (nearest, zeros outside, pixel-centre coordinates),
``resize(antialias=True)`` = ``interpolate(mode="bilinear", antialias=True, align_corners=False)``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import device_ops as _dev


def rotate_nearest(image: torch.Tensor, angle: float) -> torch.Tensor:
    """Counter-clockwise rotation by ``angle`` degrees about the image centre: nearest interpolation, same canvas, zero fill
    (torchvision ``rotate(img, angle)`` defaults for tensors)."""
    if angle % 360 == 0:
        return image
    x = image.unsqueeze(0) if image.dim() == 3 else image
    B, _, H, W = x.shape
    rad = math.radians(angle)
    c, s = math.cos(rad), math.sin(rad)
    # offsets of the output pixel centres from the image centre; each samples the input at R^-1 (offset)
    dx = torch.arange(W, dtype=x.dtype, device=x.device) + 0.5 - 0.5 * W
    dy = torch.arange(H, dtype=x.dtype, device=x.device) + 0.5 - 0.5 * H
    dx, dy = dx[None, :].expand(H, W), dy[:, None].expand(H, W)
    src_x = (c * dx - s * dy) / (0.5 * W)          # grid_sample's [-1, 1] coordinates (align_corners=False)
    src_y = (s * dx + c * dy) / (0.5 * H)
    grid = torch.stack([src_x, src_y], dim=-1).expand(B, H, W, 2)
    out = TF.grid_sample(x, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    return out[0] if image.dim() == 3 else out


def upper_left_crop(image: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    return image[..., :out_h, :out_w]


def resize_bilinear(image: torch.Tensor, size) -> torch.Tensor:
    x = image.unsqueeze(0) if image.dim() == 3 else image
    x = TF.interpolate(x, size=tuple(size), mode="bilinear", antialias=True, align_corners=False)
    return x[0] if image.dim() == 3 else x


class _Named(nn.Module):
    def __repr__(self):
        return type(self).__name__


class Identity(_Named):
    def forward(self, image, *args, **kwargs):
        return image


class HorizontalFlip(_Named):
    def forward(self, image, *args, **kwargs):
        if _dev.eligible(image):
            return _dev.run(_dev.FLIP_H, image)
        return image.flip(-1)


class Rotate(_Named):
    """Whole quarter turns are exact (a transpose/flip, like ``rotate(..., expand=True)`` on a square canvas); the remainder
    in [0, 90) is resampled (geometric.py:38-46)."""

    def __init__(self, min_angle=None, max_angle=None, do90=False):
        super().__init__()
        self.min_angle, self.max_angle = min_angle, max_angle
        self.base_angles = torch.tensor([-90, 0, 0, 90] if do90 else [0])

    def get_random_angle(self):
        if self.min_angle is None or self.max_angle is None:
            raise ValueError("min_angle and max_angle must be provided")
        pick = torch.randint(0, len(self.base_angles), size=(1,))
        jitter = torch.randint(self.min_angle, self.max_angle + 1, size=(1,)).item()
        return self.base_angles[pick].item() + jitter

    def forward(self, image, angle=None):
        angle = self.get_random_angle() if angle is None else angle
        quarters, rest = divmod(angle, 90)          # floor division: -20 -> (-1, 70)
        if _dev.eligible(image) and (quarters % 2 == 0 or image.shape[-1] == image.shape[-2]):
            return image if angle % 360 == 0 else _dev.run(_dev.ROTATE, image, quarters % 4, rest)      # MI355X: one launch, both steps
        if quarters % 4:
            image = torch.rot90(image, quarters % 4, dims=(-2, -1))
        return rotate_nearest(image, rest)


class UpperLeftCrop(_Named):
    def __init__(self, min_size=None, max_size=None):
        super().__init__()
        self.min_size, self.max_size = min_size, max_size

    def get_random_size(self, h, w):
        if self.min_size is None or self.max_size is None:
            raise ValueError("min_size and max_size must be provided")
        draw = lambda n: torch.randint(int(self.min_size * n), int(self.max_size * n) + 1, size=(1,)).item()  # noqa: E731
        return draw(h), draw(w)

    def forward(self, image, size=None):
        h, w = image.shape[-2:]
        oh, ow = (int(size * h), int(size * w)) if size is not None else self.get_random_size(h, w)
        return upper_left_crop(image, oh, ow)


class UpperLeftCropWithResizeBack(_Named):
    def __init__(self):
        super().__init__()
        self.crop = UpperLeftCrop()

    def forward(self, image, crop_size=None):
        full = tuple(image.shape[-2:])
        if _dev.eligible(image) and crop_size is not None:
            oh, ow = int(crop_size * full[0]), int(crop_size * full[1])
            return image if (oh, ow) == full else _dev.run(_dev.CROP_RESIZE, image, oh, ow)
        part = self.crop(image, crop_size)
        return part if tuple(part.shape[-2:]) == full else resize_bilinear(part, full)


class UpperLeftCropWithPadBack(_Named):
    def __init__(self):
        super().__init__()
        self.crop = UpperLeftCrop()

    def forward(self, image, crop_size=None):
        full_h = image.shape[-2]
        if _dev.eligible(image) and crop_size is not None and image.shape[-1] == full_h:
            return _dev.run(_dev.CROP_PAD, image, int(crop_size * full_h), int(crop_size * full_h))
        part = self.crop(image, crop_size)
        missing = full_h - part.shape[-2]
        return TF.pad(part, (0, missing, 0, missing), mode="constant", value=0.0)      # right and bottom, as F.pad(img, (0, 0, p, p))
