"""Geometric transforms of ``wmar/augmentations/geometric.py`` (:22-117) on [B, 3, H, W] tensors in [0, 1], any device.

The reference delegates to ``torchvision.transforms.functional`` (absent offline); its tensor code paths are restated with
torch primitives: ``rotate`` = inverse affine map sampled with ``grid_sample`` (nearest, zeros outside, pixel-centre
coordinates), ``resize(antialias=True)`` = ``interpolate(mode="bilinear", antialias=True, align_corners=False)``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as TF


def _rotate_nearest(image: torch.Tensor, angle: float) -> torch.Tensor:
    """torchvision.transforms.functional.rotate(img, angle) for tensors: counter-clockwise by ``angle`` degrees about the image
    centre, nearest interpolation, expand=False, fill 0."""
    if angle % 360 == 0:
        return image
    squeeze = image.dim() == 3
    x = image.unsqueeze(0) if squeeze else image
    B, _, H, W = x.shape
    # output pixel (x_o, y_o) relative to the centre samples input at R(-angle) applied in image coordinates (y down)
    a = math.radians(angle)
    # torchvision builds the inverse matrix of a rotation by -angle: [cos a, -sin a; sin a, cos a] acting on (x, y) offsets
    cos, sin = math.cos(a), math.sin(a)
    theta = torch.tensor([[cos, -sin, 0.0], [sin, cos, 0.0]], dtype=x.dtype, device=x.device)
    # base grid of pixel-centre offsets, normalised the way torchvision's _gen_affine_grid does
    xs = (torch.arange(W, dtype=x.dtype, device=x.device) + 0.5 - W * 0.5)
    ys = (torch.arange(H, dtype=x.dtype, device=x.device) + 0.5 - H * 0.5)
    gx = xs[None, :].expand(H, W)
    gy = ys[:, None].expand(H, W)
    sx = (theta[0, 0] * gx + theta[0, 1] * gy) / (0.5 * W)
    sy = (theta[1, 0] * gx + theta[1, 1] * gy) / (0.5 * H)
    grid = torch.stack([sx, sy], dim=-1)[None].expand(B, H, W, 2)
    out = TF.grid_sample(x, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    return out.squeeze(0) if squeeze else out


class Identity(nn.Module):
    def forward(self, image, *args, **kwargs):
        return image

    def __repr__(self):
        return "Identity"


class Rotate(nn.Module):
    """geometric.py:22-48: the multiple-of-90 part first (expand=True: an exact transpose/flip), then the remainder."""

    def __init__(self, min_angle=None, max_angle=None, do90=False):
        super().__init__()
        self.min_angle, self.max_angle = min_angle, max_angle
        self.base_angles = torch.tensor([-90, 0, 0, 90]) if do90 else torch.tensor([0])

    def get_random_angle(self):
        if self.min_angle is None or self.max_angle is None:
            raise ValueError("min_angle and max_angle must be provided")
        base_angle = self.base_angles[torch.randint(0, len(self.base_angles), size=(1,))].item()
        return base_angle + torch.randint(self.min_angle, self.max_angle + 1, size=(1,)).item()

    def forward(self, image, angle=None):
        if angle is None:
            angle = self.get_random_angle()
        base_angle = angle // 90 * 90
        angle = angle - base_angle
        k = (base_angle // 90) % 4
        if k:
            image = torch.rot90(image, k, dims=(-2, -1))
        return _rotate_nearest(image, angle)

    def __repr__(self):
        return "Rotate"


class UpperLeftCrop(nn.Module):
    def __init__(self, min_size=None, max_size=None):
        super().__init__()
        self.min_size, self.max_size = min_size, max_size

    def get_random_size(self, h, w):
        if self.min_size is None or self.max_size is None:
            raise ValueError("min_size and max_size must be provided")
        return (torch.randint(int(self.min_size * h), int(self.max_size * h) + 1, size=(1,)).item(),
                torch.randint(int(self.min_size * w), int(self.max_size * w) + 1, size=(1,)).item())

    def forward(self, image, size=None):
        h, w = image.shape[-2:]
        oh, ow = self.get_random_size(h, w) if size is None else (int(size * h), int(size * w))
        return image[..., :oh, :ow]


class UpperLeftCropWithResizeBack(nn.Module):
    def __init__(self):
        super().__init__()
        self.crop = UpperLeftCrop()

    def forward(self, image, crop_size=None):
        out_size = (image.shape[-2], image.shape[-1])
        image = self.crop(image, crop_size)
        if tuple(image.shape[-2:]) == out_size:
            return image
        squeeze = image.dim() == 3
        x = image.unsqueeze(0) if squeeze else image
        x = TF.interpolate(x, size=out_size, mode="bilinear", antialias=True, align_corners=False)
        return x.squeeze(0) if squeeze else x


class UpperLeftCropWithPadBack(nn.Module):
    def __init__(self):
        super().__init__()
        self.crop = UpperLeftCrop()

    def forward(self, image, crop_size=None):
        out_h = image.shape[-2]
        image = self.crop(image, crop_size)
        pad = out_h - image.shape[-2]
        return TF.pad(image, (0, pad, 0, pad), mode="constant", value=0.0)   # F.pad(img, (0, 0, pad, pad)): right and bottom


class HorizontalFlip(nn.Module):
    def forward(self, image, *args, **kwargs):
        return torch.flip(image, dims=(-1,))

    def __repr__(self):
        return "HorizontalFlip"
