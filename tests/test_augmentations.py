"""Evaluation-time transforms: semantics of the torchvision calls the reference makes, restated on torch
primitives.  torchvision is not installed, so the checks are the functions' defining properties and closed forms."""
import math

import numpy as np
import torch

from wmar_amd.augmentations import AugmentationManager
from wmar_amd.augmentations.geometric import HorizontalFlip, Rotate, UpperLeftCropWithPadBack, UpperLeftCropWithResizeBack
from wmar_amd.augmentations.valuemetric import JPEG, Brightness, GaussianBlur, GaussianNoise, _gaussian_kernel1d


def _img(B=2, H=32, W=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 3, H, W, generator=g)


def test_table_matches_reference_parameter_lists():
    m = AugmentationManager(False, False, True)
    assert [a[0] for a in m.augs] == ["gaussian-blur", "gaussian-noise", "jpeg", "brightness", "rotation", "flip-h", "upperleft-crop"]
    assert m.augs[0][2] == [0, 1, 3, 5, 7, 9, 11, 13, 15, 17, 19] and m.augs[4][2] == [-20, -15, -10, -5, 0, 5, 10, 15, 20]
    x = _img()
    for name, fn, params in m.augs:
        for p in params[:3]:
            y = fn(x.clone(), p)
            assert y.shape == x.shape and float(y.min()) >= 0 and float(y.max()) <= 1, (name, p)


def test_identities():
    x = _img()
    assert torch.equal(GaussianBlur()(x, 0), x)
    assert torch.allclose(GaussianBlur()(x, 1), x, atol=1e-6)          # 1-tap kernel
    assert torch.equal(Brightness()(x, 1), x)
    assert torch.equal(GaussianNoise()(x, 0), x)
    assert torch.equal(Rotate()(x, 0), x)
    assert torch.equal(UpperLeftCropWithResizeBack()(x, 1.0), x)
    assert torch.equal(HorizontalFlip()(HorizontalFlip()(x)), x)
    assert torch.equal(HorizontalFlip()(x)[..., 0], x[..., -1])


def test_blur_kernel_closed_form():
    k = _gaussian_kernel1d(5, 0.3 * (2 - 1) + 0.8, torch.float64, "cpu")
    ref = np.exp(-0.5 * (np.arange(-2, 3) / 1.1) ** 2)
    np.testing.assert_allclose(k.numpy(), ref / ref.sum(), rtol=1e-12)
    const = torch.full((1, 3, 16, 16), 0.37)
    assert torch.allclose(GaussianBlur()(const, 7), const, atol=1e-6)   # normalised kernel + reflect padding keep constants
    x = _img(1, 24, 24)
    y = GaussianBlur()(x, 9)
    assert float(y.var()) < float(x.var())


def test_rotate_quarter_turns_exact_and_small_angles():
    x = _img(1, 16, 16)
    assert torch.equal(Rotate()(x, 90), torch.rot90(x, 1, dims=(-2, -1)))
    assert torch.equal(Rotate()(x, -90 + 0), torch.rot90(x, -1, dims=(-2, -1)))
    # a centred blob stays centred; corners are filled with zeros at 20 degrees
    y = Rotate()(torch.ones(1, 3, 64, 64), 20)
    assert float(y[0, 0, 32, 32]) == 1.0 and float(y[0, 0, 0, 0]) == 0.0
    # nearest sampling: every output value is one of the input values or the zero fill
    z = Rotate()(x, 10)
    vals = set(x.flatten().tolist()) | {0.0}
    assert set(z.flatten().tolist()) <= vals
    # counter-clockwise: a mark right of the centre moves up (towards smaller row index)
    m = torch.zeros(1, 3, 65, 65)
    m[..., 32, 60] = 1.0
    r = Rotate()(m, 20)
    yy, xx = np.argwhere(r[0, 0].numpy() == 1.0)[0]
    assert yy < 32 and abs(math.hypot(yy - 32, xx - 32) - 28) < 1.5


def test_crop_variants():
    x = _img(1, 32, 32)
    y = UpperLeftCropWithResizeBack()(x, 0.5)
    ref = torch.nn.functional.interpolate(x[..., :16, :16], size=(32, 32), mode="bilinear", antialias=True, align_corners=False)
    assert torch.equal(y, ref)
    p = UpperLeftCropWithPadBack()(x, 0.75)
    assert p.shape == x.shape and torch.equal(p[..., :24, :24], x[..., :24, :24]) and float(p[..., 24:, :].abs().max()) == 0


def test_jpeg_roundtrip_is_lossy_but_close():
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 32), torch.linspace(0, 1, 32), indexing="ij")
    x = torch.stack([yy, xx, 0.5 * (yy + xx)])[None].repeat(2, 1, 1, 1).contiguous()     # smooth ramps
    y = JPEG()(x.clone(), 95)
    assert y.shape == x.shape and 0 < float((y - x).abs().mean()) < 0.1
    assert float((JPEG()(x.clone(), 5) - x).abs().mean()) > float((y - x).abs().mean())
