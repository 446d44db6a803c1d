"""The evaluation transforms as kernels of this build (wmar_augment, wmar_amd/csrc/augment.hip) on the MI355X:

  * every (transform, parameter) of the AugmentationManager table (generate.py:142-164; jpeg stays on the host) in the harness's
    fused form -- [-1, 1] in, transform in [0, 1], clamp, [-1, 1] out, one launch -- against the torch restatements of the published
    torchvision algorithms run on the CPU (wmar_amd/augmentations/{valuemetric,geometric}.py; themselves checked against independent
    numpy restatements by tests/test_augmentations_algorithms.py): pointwise transforms bit-equal, blur / resize within 3e-6 (summation
    order), rotation identical outside float ties at .5 sample boundaries;
  * the fused launch is bit-identical to the unfused device sequence (range change, module, clamp, range change) it replaces;
  * sizes that are not multiples of the 16 x 16 blur tile, 3-D inputs, in-place rejection."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from wmar_amd.augmentations import AugmentationManager  # noqa: E402
from wmar_amd.augmentations import device_ops as D  # noqa: E402


def _table():
    return [(n, f, p) for n, f, p in AugmentationManager(False, False, True).augs if n != "jpeg"]


@pytest.mark.parametrize("size", [64, 50])
def test_fused_launch_equals_the_torch_restatements_on_cpu(size):
    g = torch.Generator().manual_seed(size)
    imgs = (torch.rand(3, 3, size, size, generator=g) * 2.4 - 1.2).clamp(-1, 1)          # saturated pixels included
    dev = imgs.cuda()
    for name, fn, params in _table():
        for p in params:
            if name == "gaussian-noise":
                n = torch.randn(imgs.shape, generator=g)
                got = D.run(D.NOISE, dev, p, noise=n.cuda(), pm1=True).cpu()
                ref = ((imgs / 2.0 + 0.5) + p * n).clamp(0, 1) * 2.0 - 1.0
                assert torch.equal(got, ref), (name, p)
                continue
            got = D.fused(name, dev, p)
            assert got is not None and got.is_cuda and got.shape == dev.shape, (name, p)
            ref = fn(imgs / 2.0 + 0.5, p).clamp(0, 1) * 2.0 - 1.0                           # CPU tensors: the torch restatements
            d = (got.cpu() - ref).abs()
            if name in ("brightness", "flip-h"):
                assert torch.equal(got.cpu(), ref), (name, p)
            elif name == "rotation":
                assert float((d > 0).float().mean()) < 0.01, (name, p, float((d > 0).float().mean()))
            else:
                assert float(d.max()) <= 6e-6, (name, p, float(d.max()))                     # 3e-6 in [0, 1] = 6e-6 in [-1, 1]


def test_fused_launch_is_bit_identical_to_the_unfused_device_sequence():
    g = torch.Generator(device="cuda").manual_seed(7)
    imgs = (torch.rand(4, 3, 96, 96, generator=g, device="cuda") * 2.2 - 1.1).clamp(-1, 1)
    for name, fn, params in _table():
        if name == "gaussian-noise":
            continue                                                                        # (its draws differ call to call; covered above)
        for p in params:
            a = D.fused(name, imgs, p)
            b = fn(imgs / 2.0 + 0.5, p).clamp(0, 1) * 2.0 - 1.0
            assert torch.equal(a, b), (name, p)


def test_shapes_and_errors():
    from wmar_amd import _lib
    from wmar_amd.augmentations.geometric import Rotate, UpperLeftCropWithPadBack
    from wmar_amd.augmentations.valuemetric import GaussianBlur
    x = torch.rand(3, 40, 40, device="cuda")
    y = GaussianBlur()(x, 5)
    assert y.shape == x.shape and torch.allclose(y, GaussianBlur()(x.cpu(), 5).cuda(), atol=3e-6)
    assert torch.equal(Rotate()(x, 90), torch.rot90(x, 1, dims=(-2, -1))) and torch.equal(Rotate()(x, -90), torch.rot90(x, -1, dims=(-2, -1)))
    assert torch.equal(Rotate()(x, 180), torch.rot90(x, 2, dims=(-2, -1)))
    z = UpperLeftCropWithPadBack()(x, 0.5)
    assert torch.equal(z[:, :20, :20], x[:, :20, :20]) and not bool(z[:, 20:].any()) and not bool(z[:, :, 20:].any())
    with pytest.raises(_lib.WmarError):
        D.run(D.BLUR, torch.rand(1, 3, 4, 4, device="cuda"), 19)                           # reflect padding needs k // 2 < size
    rect = torch.rand(2, 3, 24, 40, device="cuda")                                          # odd quarter turns of a non-square image: torch path
    assert Rotate()(rect, 90).shape == (2, 3, 40, 24)


def test_harness_fuses_only_the_managers_own_callables():
    """fill_batch_log replaces the AugmentationManager's entries by the fused launch (same pixels) and leaves a caller's own callable
    registered under one of those names alone"""
    from wmar_amd import harness

    class M:
        def codes_to_images(self, codes): return (codes.float().view(-1, 1, 8, 8).repeat(1, 3, 4, 4) / 32.0 - 1.0).cuda()
        def images_to_codes(self, imgs): return ((imgs[:, 0, ::4, ::4] + 1.0) * 32.0).round().long().view(imgs.shape[0], -1)

    codes = torch.randint(0, 64, (3, 64), device="cuda")
    mine = lambda x, p: x * 0.0 + 0.25                                           # noqa: E731
    table = [a for a in AugmentationManager(False, False, True).augs if a[0] == "brightness"]
    logs = {}
    for key, augs in (("default", table), ("custom", [("brightness", mine, [2.0])]), ("unfused", table)):
        ev = {"metric_names": [], "augmentations": augs, "max_roundtrips": 0, "orig_only": False, "fuse_augmentations": key != "unfused"}
        log = {}
        harness.fill_batch_log(log, "m", M(), codes, ev)
        logs[key] = log["m"]["brightness"]
    assert np.allclose(logs["custom"][0][2], 0.25 * 2 - 1)                       # the caller's transform ran
    for a, b in zip(logs["default"], logs["unfused"]):
        assert a[0] == b[0] and np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])
