"""Value transforms on [B, 3, H, W] (or [3, H, W]) tensors in [0, 1], any device; API of ``wmar/augmentations/valuemetric.py``
(:41-140): ``JPEG``, ``GaussianBlur``, ``Brightness``, ``GaussianNoise`` modules called as ``T()(image, parameter)``; without a
parameter one is drawn from the range given at construction.

The reference delegates to ``torchvision.transforms.functional`` (absent offline).  Its tensor code paths are restated here:
``gaussian_blur(img, k)`` = separable kernel exp(-x^2 / 2 sigma^2) on x = -(k-1)/2 .. (k-1)/2 with
sigma = 0.3 ((k-1)/2 - 1) + 0.8, reflect padding, depthwise convolution; ``adjust_brightness`` = factor * img clamped to [0, 1].
JPEG stays on the host (PIL), as in the reference.
"""
from __future__ import annotations

import io

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import device_ops as _dev


# ------------------------------------------------------------------ functional forms
def _gaussian_kernel1d(kernel_size: int, sigma: float, dtype, device) -> torch.Tensor:
    half = (kernel_size - 1) * 0.5
    taps = torch.linspace(-half, half, steps=kernel_size, dtype=dtype, device=device)
    w = torch.exp(-0.5 * (taps / sigma).pow(2))
    return w / w.sum()


def _as_batch(image):
    return (image.unsqueeze(0), True) if image.dim() == 3 else (image, False)


def gaussian_blur(image: torch.Tensor, kernel_size: int) -> torch.Tensor:
    sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
    k1 = _gaussian_kernel1d(kernel_size, sigma, image.dtype, image.device)
    x, squeezed = _as_batch(image)
    ch, pad = x.shape[1], kernel_size // 2
    weight = (k1[:, None] * k1[None, :]).expand(ch, 1, kernel_size, kernel_size)
    x = TF.conv2d(TF.pad(x, (pad, pad, pad, pad), mode="reflect"), weight, groups=ch)
    return x[0] if squeezed else x


def adjust_brightness(image: torch.Tensor, factor: float) -> torch.Tensor:
    return (image * factor).clamp(0, 1)


def add_gaussian_noise(image: torch.Tensor, std: float) -> torch.Tensor:
    return image + std * torch.randn_like(image)


def jpeg_compress(image: torch.Tensor, quality: int) -> torch.Tensor:
    """One 3xHxW image in [0, 1] through an 8-bit JPEG file in memory (ToPILImage truncates x*255 to uint8; ToTensor divides by 255)."""
    from PIL import Image
    lo, hi = float(image.min()), float(image.max())
    assert lo >= 0 and hi <= 1, f"Image pixel values must be in the range [0, 1], got [{lo}, {hi}]"
    hwc = (image.detach().cpu() * 255).to(torch.uint8).permute(1, 2, 0).numpy()
    with io.BytesIO() as buf:
        Image.fromarray(hwc).save(buf, format="JPEG", quality=quality)
        buf.seek(0)
        back = np.asarray(Image.open(buf).convert("RGB"), dtype=np.float32)
    return torch.from_numpy(back / 255.0).permute(2, 0, 1)


# ------------------------------------------------------------------ module forms
class _Ranged(nn.Module):
    """A transform with one scalar parameter drawn from [lo, hi] when the caller gives none."""
    what = "parameter"

    def __init__(self, lo=None, hi=None):
        super().__init__()
        self.lo, self.hi = lo, hi

    def _need_range(self):
        if self.lo is None or self.hi is None:
            raise ValueError(f"{self.what} range must be specified")

    def _uniform(self):
        self._need_range()
        return self.lo + torch.rand(1).item() * (self.hi - self.lo)

    def _integer(self):
        self._need_range()
        return torch.randint(self.lo, self.hi + 1, size=(1,)).item()

    def __repr__(self):
        return type(self).__name__


class JPEG(_Ranged):
    what = "Quality"

    def __init__(self, min_quality=None, max_quality=None, passthrough=True):
        super().__init__(min_quality, max_quality)
        self.passthrough = passthrough          # straight-through gradient form (valuemetric.py:54-58)

    get_random_quality = _Ranged._integer

    def jpeg_single(self, image, quality):
        coded = jpeg_compress(image, quality).to(image.device)
        return image + (coded - image).detach() if self.passthrough else coded

    def forward(self, image, quality=None):
        quality = quality or self.get_random_quality()
        image = image.clamp(0, 1)
        out = torch.stack([self.jpeg_single(one, quality) for one in image]) if image.dim() == 4 else self.jpeg_single(image, quality)
        return out.clamp(0, 1)


class GaussianBlur(_Ranged):
    what = "Kernel size"

    def __init__(self, min_kernel_size=None, max_kernel_size=None):
        super().__init__(min_kernel_size, max_kernel_size)

    def get_random_kernel_size(self):
        return self._integer() | 1              # even draws move up to the next odd size

    def forward(self, image, kernel_size=None):
        if kernel_size == 0:                    # the sweep's "no blur" entry
            return image
        k = kernel_size or self.get_random_kernel_size()
        if _dev.eligible(image):                # MI355X: one launch over the batch (csrc/augment.hip), clamp inside
            return _dev.run(_dev.BLUR, image, k)
        return gaussian_blur(image, k).clamp(0, 1)


class Brightness(_Ranged):
    what = "Brightness factor"

    def __init__(self, min_factor=None, max_factor=None):
        super().__init__(min_factor, max_factor)

    get_random_factor = _Ranged._uniform

    def forward(self, image, factor=None):
        factor = self.get_random_factor() if factor is None else factor
        if _dev.eligible(image):
            return _dev.run(_dev.BRIGHTNESS, image, factor)
        return adjust_brightness(image, factor)


class GaussianNoise(_Ranged):
    what = "Standard deviation"

    def __init__(self, min_std=None, max_std=None):
        super().__init__(min_std, max_std)

    get_random_std = _Ranged._uniform

    def forward(self, image, std=None):
        std = self.get_random_std() if std is None else std
        if _dev.eligible(image):                # the draws stay torch.randn_like (the reference's RNG call); the arithmetic is the kernel's
            return _dev.run(_dev.NOISE, image, std, noise=torch.randn_like(image))
        return add_gaussian_noise(image, std).clamp(0, 1)
