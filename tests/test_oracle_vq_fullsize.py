"""CPU: the torch restatements of the three image tokenizers (oracle/model_oracle.py, oracle/rar_oracle.py) against the reference's
own modules AT THE BENCHMARK SHAPES (tests/golden/fullsize_vq_vectors.npz): Taming VQGAN 256 x 256 / 16384 x 256 codebook,
MaskGIT-VQGAN 256 x 256 / 1024 codes, Chameleon's VQGAN 512 x 512 / 8192 codes."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import model_oracle as M
from oracle import rar_oracle as R
from wmar_amd.utils import synth

from vq_fullsize_common import check_codes, load, sub


def _decode_unclamped(sd, cfg, codes):
    B, S = codes.shape[0], cfg.codes_size
    zq = sd["quantize.embedding.weight"][codes.reshape(-1)].view(B, S, S, cfg.embed_dim).permute(0, 3, 1, 2).contiguous()
    z = F.conv2d(zq, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    return M.decoder_forward(M._sub(sd, "decoder."), cfg.num_resolutions, cfg.num_res_blocks, z)


def test_taming_vqgan_oracle_at_bench_shape():
    fx = load()
    cfg = synth.TAMING_VQ
    sd = synth.synth_vq_state(cfg, seed=31)
    codes = torch.from_numpy(fx["tam_codes"])
    with torch.no_grad():
        img = M.codes_to_images(sd, cfg, codes)
        np.testing.assert_allclose(sub(img.numpy()), fx["tam_pixels"], rtol=0, atol=2e-5)
        z = M.encode_prequant(sd, cfg, img)
        np.testing.assert_allclose(z.numpy()[::8], fx["tam_prequant"], rtol=0, atol=1e-4)
        c2 = M.quantize_argmin(sd["quantize.embedding.weight"], z).view(2, -1)
    check_codes(c2.numpy(), fx["tam_codes_roundtrip"], fx["tam_margin"], "taming oracle")


def test_maskgit_vqgan_oracle_at_bench_shape():
    fx = load()
    cfg = synth.MASKGIT_VQ
    sd = synth.synth_maskgit_state(cfg, seed=33)
    codes = torch.from_numpy(fx["mg_codes"])
    img = R.maskgit_decode(sd, cfg, codes)
    np.testing.assert_allclose(sub(img.numpy()), fx["mg_pixels"], rtol=0, atol=1e-4)      # fp32 summation order of the two torch graphs
    z = R.maskgit_prequant(sd, cfg, img)
    np.testing.assert_allclose(z.numpy()[::8], fx["mg_prequant"], rtol=0, atol=1e-4)
    c2 = R.maskgit_argmin(sd["quantize.embedding.weight"], z).view(2, -1)
    check_codes(c2.numpy(), fx["mg_codes_roundtrip"], fx["mg_margin"], "maskgit oracle")


def test_chameleon_vqgan_oracle_at_512():
    fx = load()
    cfg = synth.CHAMELEON_VQ
    sd = synth.synth_vq_state(cfg, seed=35)
    codes = torch.from_numpy(fx["ch_codes"])
    with torch.no_grad():
        img = _decode_unclamped(sd, cfg, codes)                 # the reference's VQModel.decode does not clamp
        np.testing.assert_allclose(sub(img.numpy()), fx["ch_pixels"], rtol=0, atol=5e-5)
        z = M.encode_prequant(sd, cfg, img)
        np.testing.assert_allclose(z.numpy()[::16], fx["ch_prequant"], rtol=0, atol=2e-4)
        c2 = M.quantize_argmin(sd["quantize.embedding.weight"], z).view(1, -1)
    check_codes(c2.numpy(), fx["ch_codes_roundtrip"], fx["ch_margin"], "chameleon oracle")
