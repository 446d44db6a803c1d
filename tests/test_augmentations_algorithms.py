"""f2: the evaluation transforms against INDEPENDENT restatements of the published torchvision algorithms (torchvision is not
installed offline, so the reference's own calls cannot run): plain numpy loops written from the algorithm descriptions --
``torchvision.transforms.functional.gaussian_blur`` (separable kernel exp(-x^2/2s^2), s = 0.3((k-1)/2 - 1) + 0.8, reflect
padding), ``rotate`` on tensors (inverse affine map about the image centre in pixel-centre coordinates, nearest sample with
round-half-even, zero fill; quarter turns first, as wmar/augmentations/geometric.py:38-46 does), ``resize(antialias=True)``
(separable triangle filter widened by the scale factor, normalised weights), ``adjust_brightness`` (blend with black, clamp).
The same checks run on the CPU and, in the GPU suite, on the MI355X (where the harness runs them on whole batches)."""
import math

import numpy as np
import pytest
import torch

from wmar_amd.augmentations.geometric import HorizontalFlip, Rotate, UpperLeftCropWithPadBack, UpperLeftCropWithResizeBack
from wmar_amd.augmentations.valuemetric import Brightness, GaussianBlur

DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _img(B=2, H=18, W=18, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 3, H, W, generator=g)


def _blur_ref(x, k):
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    taps = np.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k)
    w = np.exp(-0.5 * (taps / sigma) ** 2)
    w = w / w.sum()
    p = k // 2
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)), mode="reflect")
    B, C, H, W = x.shape
    out = np.zeros_like(x, dtype=np.float64)
    for i in range(k):
        for j in range(k):
            out += w[i] * w[j] * xp[:, :, i:i + H, j:j + W]
    return np.clip(out, 0, 1)


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("k", [3, 9, 17])
def test_gaussian_blur_equals_published_algorithm(dev, k):
    x = _img(H=20, W=23)
    got = GaussianBlur()(x.to(dev), k).cpu().numpy()
    np.testing.assert_allclose(got, _blur_ref(x.numpy().astype(np.float64), k), rtol=0, atol=2e-6)


def _rotate_ref(x, angle):
    """geometric.py:38-46: rotate(base, expand=True) for the quarter turns (square canvas: a pure permutation), then rotate(rest)."""
    base = angle // 90 * 90
    rest = angle - base
    q = (base // 90) % 4
    x = np.rot90(x, k=q, axes=(-2, -1)).copy() if q else x
    if rest % 360 == 0:
        return x
    B, C, H, W = x.shape
    th = math.radians(rest)
    c, s = math.cos(th), math.sin(th)
    out = np.zeros_like(x)
    for yo in range(H):
        for xo in range(W):
            dx, dy = xo + 0.5 - 0.5 * W, yo + 0.5 - 0.5 * H
            sx, sy = c * dx - s * dy, s * dx + c * dy                 # inverse map of a counter-clockwise rotation
            xi = int(np.rint(np.float32(sx + 0.5 * W - 0.5)))
            yi = int(np.rint(np.float32(sy + 0.5 * H - 0.5)))
            if 0 <= xi < W and 0 <= yi < H:
                out[:, :, yo, xo] = x[:, :, yi, xi]
    return out


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("angle", [-20, -5, 5, 10, 20, 90])
def test_rotate_equals_published_algorithm(dev, angle):
    x = _img(H=16, W=16, seed=3)
    got = Rotate()(x.to(dev), angle).cpu().numpy()
    ref = _rotate_ref(x.numpy(), angle)
    # nearest sampling: identical pixels, except where a source coordinate sits within float rounding of a .5 boundary
    assert (got != ref).mean() < 0.01, (angle, float((got != ref).mean()))
    assert got.shape == ref.shape


def _aa_weights(n_in, n_out):
    scale = n_in / n_out
    support = max(scale, 1.0)
    W = np.zeros((n_out, n_in))
    for i in range(n_out):
        center = scale * (i + 0.5)
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), n_in)
        ws = [max(0.0, 1.0 - abs((j - center + 0.5) / max(scale, 1.0))) for j in range(lo, hi)]
        W[i, lo:hi] = np.array(ws) / sum(ws)
    return W


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("factor", [0.5, 0.75, 0.9])
def test_crop_resize_back_equals_published_algorithm(dev, factor):
    x = _img(H=20, W=20, seed=5)
    got = UpperLeftCropWithResizeBack()(x.to(dev), factor).cpu().numpy()
    n = int(factor * 20)
    crop = x.numpy().astype(np.float64)[:, :, :n, :n]
    Wm = _aa_weights(n, 20)
    ref = np.einsum("oi,bcij,pj->bcop", Wm, crop, Wm)
    np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6)


@pytest.mark.parametrize("dev", DEVICES)
def test_brightness_flip_padback(dev):
    x = _img(seed=7).to(dev)
    for f in (1.25, 2.0, 3.0):
        np.testing.assert_array_equal(Brightness()(x, f).cpu().numpy(), np.clip(x.cpu().numpy() * np.float32(f), 0, 1))
    np.testing.assert_array_equal(HorizontalFlip()(x).cpu().numpy(), x.cpu().numpy()[..., ::-1])
    y = UpperLeftCropWithPadBack()(x, 0.5).cpu().numpy()
    assert y.shape == tuple(x.shape) and np.array_equal(y[..., :9, :9], x.cpu().numpy()[..., :9, :9]) and not y[..., 9:, :].any()


@pytest.mark.gpu
def test_sweep_runs_batched_on_device_through_the_detector(kat):
    """the harness's use (generate.py:142-164): a batch of images through a transform, re-encoded and scored, all on the GPU."""
    from tests.test_gpu_watermark import _wm
    from wmar_amd.augmentations import AugmentationManager
    from wmar_amd.models.taming_wrapper import TamingARMMWrapper
    from wmar_amd.utils import synth
    gcfg, vcfg = synth.GPTConfig(**synth.HARNESS_GPT), synth.VQConfig(**synth.HARNESS_VQ)
    m = TamingARMMWrapper(None, gpt_cfg=gcfg, vq_cfg=vcfg, gpt_state=synth.synth_gpt_state(gcfg, 21, "cpu", 40.0),
                          vq_state=synth.synth_vq_state(vcfg, 21, "cpu"), max_batch=4)
    wm = _wm(kat["keys"]["taming"])
    imgs = m.codes_to_images(torch.randint(0, 16384, (4, 64), device="cuda")) / 2 + 0.5
    for name, fn, params in AugmentationManager(False, False, True).augs:
        for p in (params[1], params[-1]):
            y = fn(imgs.clone(), p).clamp(0, 1)
            assert y.is_cuda and y.shape == imgs.shape
            pv = wm.detect(m.images_to_codes(y * 2 - 1))
            assert pv.shape == (4,) and bool(((pv >= 0) & (pv <= 1)).all())
