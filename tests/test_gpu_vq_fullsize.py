"""MI355X: the HIP image tokenizers against the reference's own outputs AT THE SHAPES THE BENCHMARKS RUN
(tests/golden/fullsize_vq_vectors.npz): k_conv_bx / k_conv at ch 128 x (1,1,2,2,4), attention over 256 tokens as 1x1 convolutions,
128 -> 3 at 256 x 256, the stride-2 convs at 256 -> 128, k_vq_argmin over 16384 x 256 -- plus a live oracle comparison on one more
image.  Tolerances: pixels 2e-4, pre-quantisation vectors 5e-4, codes equal except at the reference's own near ties, p-values 1e-5
(reference: deps/taming/modules/diffusionmodules/model.py:407-434, 507-538,
deps/taming/modules/vqvae/quantize.py:272-314, deps/rar/modeling/modules/maskgit_vqgan.py:157-362,
deps/chameleon/inference/vqgan.py:330-570)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model_oracle as M  # noqa: E402
from oracle import rar_oracle as R  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402

from test_oracle_vq_fullsize import _decode_unclamped  # noqa: E402
from vq_fullsize_common import check_codes, load, sub  # noqa: E402


def test_taming_vqgan_bench_shape_vs_reference():
    from wmar_amd.models.engine import VQGANEngine
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    fx = load()
    cfg = synth.TAMING_VQ
    sd = synth.synth_vq_state(cfg, seed=31)
    eng = VQGANEngine(cfg, sd, max_batch=4)
    codes = torch.from_numpy(fx["tam_codes"])
    img = eng.decode(codes.cuda())
    assert img.shape == (2, 3, 256, 256)
    np.testing.assert_allclose(sub(img.cpu().numpy()), fx["tam_pixels"], rtol=0, atol=2e-4)
    # re-encode the REFERENCE's image (the oracle reproduces it to 2e-5: tests/test_oracle_vq_fullsize.py)
    ref_img = M.codes_to_images(sd, cfg, codes)
    np.testing.assert_allclose(img.cpu().numpy(), ref_img.numpy(), rtol=0, atol=2e-4)          # every pixel, not only the stored ones
    got, pre = eng.encode(ref_img.cuda(), return_prequant=True)
    np.testing.assert_allclose(pre.cpu().numpy()[::8], fx["tam_prequant"], rtol=0, atol=5e-4)
    n_bad = check_codes(got.cpu().numpy(), fx["tam_codes_roundtrip"], fx["tam_margin"], "taming")
    # the detector on both code sets (the round trip decides codes' and hence the p-value)
    class _VQ:
        pass
    from conftest import load_ids
    alive = load_ids("vqgan_alive_ids.txt")
    wm = GentimeWatermark({"alive_ids": torch.tensor(alive), "dead_ids": torch.tensor(sorted(set(range(16384)) - set(alive))),
                           "embedding": torch.zeros(16384, 4)}, 16384, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25,
                          device="cuda")
    np.testing.assert_allclose(wm.detect(codes.cuda()).cpu().numpy(), fx["tam_pvals"], rtol=0, atol=1e-5)
    if n_bad == 0:
        np.testing.assert_allclose(wm.detect(got).cpu().numpy(), fx["tam_pvals_roundtrip"], rtol=0, atol=1e-5)
    # one more image, live against the oracle (a different code pattern: a constant code and a ramp)
    extra = torch.stack([torch.full((256,), 4242, dtype=torch.int64), (torch.arange(256) * 61) % cfg.n_embed])
    ref2 = M.codes_to_images(sd, cfg, extra)
    np.testing.assert_allclose(eng.decode(extra.cuda()).cpu().numpy(), ref2.numpy(), rtol=0, atol=2e-4)
    z2 = M.encode_prequant(sd, cfg, ref2)
    got2, pre2 = eng.encode(ref2.cuda(), return_prequant=True)
    np.testing.assert_allclose(pre2.cpu().numpy(), z2.numpy(), rtol=0, atol=5e-4)
    d = torch.cdist(z2, sd["quantize.embedding.weight"]) ** 2
    two = torch.topk(d, 2, dim=1, largest=False).values
    check_codes(got2.cpu().numpy(), M.quantize_argmin(sd["quantize.embedding.weight"], z2).numpy(), (two[:, 1] - two[:, 0]).numpy(), "taming live")


def test_maskgit_vqgan_bench_shape_vs_reference():
    from wmar_amd.models.engine import MaskgitVQEngine
    fx = load()
    cfg = synth.MASKGIT_VQ
    sd = synth.synth_maskgit_state(cfg, seed=33)
    eng = MaskgitVQEngine(cfg, sd, max_batch=4)
    codes = torch.from_numpy(fx["mg_codes"])
    img = eng.decode(codes.cuda())
    np.testing.assert_allclose(sub(img.cpu().numpy()), fx["mg_pixels"], rtol=0, atol=2e-4)
    ref_img = R.maskgit_decode(sd, cfg, codes)
    np.testing.assert_allclose(img.cpu().numpy(), ref_img.numpy(), rtol=0, atol=2e-4)
    got, pre = eng.encode(ref_img.cuda(), return_prequant=True)
    np.testing.assert_allclose(pre.cpu().numpy()[::8], fx["mg_prequant"], rtol=0, atol=5e-4)
    check_codes(got.cpu().numpy(), fx["mg_codes_roundtrip"], fx["mg_margin"], "maskgit")


def test_chameleon_vqgan_512_vs_reference():
    from wmar_amd.models.engine import VQGANEngine
    fx = load()
    cfg = synth.CHAMELEON_VQ
    sd = synth.synth_vq_state(cfg, seed=35)
    eng = VQGANEngine(cfg, sd, max_batch=2)
    codes = torch.from_numpy(fx["ch_codes"])
    img = eng.decode(codes.cuda())
    assert img.shape == (1, 3, 512, 512)
    np.testing.assert_allclose(sub(img.cpu().numpy()), np.clip(fx["ch_pixels"], -1, 1), rtol=0, atol=3e-4)
    with torch.no_grad():
        ref_img = _decode_unclamped(sd, cfg, codes)             # what the reference's tokenizer re-encodes (no clamp in the model)
    got, pre = eng.encode(ref_img.cuda(), return_prequant=True)
    np.testing.assert_allclose(pre.cpu().numpy()[::16], fx["ch_prequant"], rtol=0, atol=8e-4)
    check_codes(got.cpu().numpy(), fx["ch_codes_roundtrip"], fx["ch_margin"], "chameleon")
