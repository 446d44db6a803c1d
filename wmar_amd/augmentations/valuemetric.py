"""Value transforms of ``wmar/augmentations/valuemetric.py`` (:41-140) on [B, 3, H, W] tensors in [0, 1], any device.

``torchvision.transforms.functional`` is restated with torch primitives: ``gaussian_blur(img, k)`` = separable kernel
exp(-x^2 / 2 sigma^2) on x = -(k-1)/2 .. (k-1)/2 with sigma = 0.3 ((k-1)/2 - 1) + 0.8, reflect padding, depthwise conv;
``adjust_brightness`` = (factor * img) clamped to [0, 1].  JPEG stays on the host (PIL), as in the reference."""
from __future__ import annotations

import io

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF


def _gaussian_kernel1d(kernel_size: int, sigma: float, dtype, device) -> torch.Tensor:
    lim = (kernel_size - 1) * 0.5
    x = torch.linspace(-lim, lim, steps=kernel_size, dtype=dtype, device=device)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(image: torch.Tensor, kernel_size: int) -> torch.Tensor:
    sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
    k1 = _gaussian_kernel1d(kernel_size, sigma, image.dtype, image.device)
    k2 = k1[:, None] * k1[None, :]
    squeeze = image.dim() == 3
    x = image.unsqueeze(0) if squeeze else image
    C = x.shape[1]
    p = kernel_size // 2
    x = TF.pad(x, (p, p, p, p), mode="reflect")
    x = TF.conv2d(x, k2[None, None].expand(C, 1, kernel_size, kernel_size), groups=C)
    return x.squeeze(0) if squeeze else x


def jpeg_compress(image: torch.Tensor, quality: int) -> torch.Tensor:
    """valuemetric.py:15-38: 3xHxW in [0,1] -> 8-bit PIL (ToPILImage: x*255 truncated to uint8) -> JPEG -> tensor / 255."""
    from PIL import Image
    assert image.min() >= 0 and image.max() <= 1, f"Image pixel values must be in the range [0, 1], got [{image.min()}, {image.max()}]"
    arr = (image.detach().cpu() * 255).to(torch.uint8).permute(1, 2, 0).numpy()
    buffer = io.BytesIO()
    Image.fromarray(arr).save(buffer, format="JPEG", quality=quality)
    buffer.seek(0)
    out = np.asarray(Image.open(buffer).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(out).permute(2, 0, 1)


class JPEG(nn.Module):
    def __init__(self, min_quality=None, max_quality=None, passthrough=True):
        super().__init__()
        self.min_quality, self.max_quality, self.passthrough = min_quality, max_quality, passthrough

    def get_random_quality(self):
        if self.min_quality is None or self.max_quality is None:
            raise ValueError("Quality range must be specified")
        return torch.randint(self.min_quality, self.max_quality + 1, size=(1,)).item()

    def jpeg_single(self, image, quality):
        if self.passthrough:
            return (jpeg_compress(image, quality).to(image.device) - image).detach() + image
        return jpeg_compress(image, quality).to(image.device)

    def forward(self, image, quality=None):
        quality = quality or self.get_random_quality()
        image = torch.clamp(image, 0, 1)
        if image.dim() == 4:
            image = torch.stack([self.jpeg_single(im, quality) for im in image])
        else:
            image = self.jpeg_single(image, quality)
        return image.clamp(0, 1)

    def __repr__(self):
        return "JPEG"


class GaussianBlur(nn.Module):
    def __init__(self, min_kernel_size=None, max_kernel_size=None):
        super().__init__()
        self.min_kernel_size, self.max_kernel_size = min_kernel_size, max_kernel_size

    def get_random_kernel_size(self):
        if self.min_kernel_size is None or self.max_kernel_size is None:
            raise ValueError("Kernel size range must be specified")
        k = torch.randint(self.min_kernel_size, self.max_kernel_size + 1, size=(1,)).item()
        return k + 1 if k % 2 == 0 else k

    def forward(self, image, kernel_size=None):
        if kernel_size == 0:
            return image
        kernel_size = kernel_size or self.get_random_kernel_size()
        return gaussian_blur(image, kernel_size).clamp(0, 1)

    def __repr__(self):
        return "GaussianBlur"


class Brightness(nn.Module):
    def __init__(self, min_factor=None, max_factor=None):
        super().__init__()
        self.min_factor, self.max_factor = min_factor, max_factor

    def get_random_factor(self):
        if self.min_factor is None or self.max_factor is None:
            raise ValueError("min_factor and max_factor must be provided")
        return torch.rand(1).item() * (self.max_factor - self.min_factor) + self.min_factor

    def forward(self, image, factor=None):
        factor = self.get_random_factor() if factor is None else factor
        return (image * factor).clamp(0, 1)

    def __repr__(self):
        return "Brightness"


class GaussianNoise(nn.Module):
    def __init__(self, min_std=None, max_std=None):
        super().__init__()
        self.min_std, self.max_std = min_std, max_std

    def get_random_std(self):
        if self.min_std is None or self.max_std is None:
            raise ValueError("Standard deviation range must be specified")
        return torch.rand(1).item() * (self.max_std - self.min_std) + self.min_std

    def forward(self, image, std=None):
        std = self.get_random_std() if std is None else std
        return (image + torch.randn_like(image) * std).clamp(0, 1)

    def __repr__(self):
        return "GaussianNoise"
