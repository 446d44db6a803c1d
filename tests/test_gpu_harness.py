"""A14 / f1: the harness (wmar_amd.harness.generate, the repo's generate.py) and the delta-checkpoint loader against outputs of
the REFERENCE's own generate.generate / update_weights on a reduced Taming model (tests/golden/make_golden.py harness_vectors:
class ids + seed -> codes, p-values, l0, psnr, file names)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from tests.conftest import REPO  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402

GPT, VQ = synth.HARNESS_GPT, synth.HARNESS_VQ
INPUTS = [c for c in (1, 9) for _ in range(3)]
GEN = {"batch_size": 4, "temperature": 1.0, "top_k": 250, "top_p": 0.92}
EVAL = {"metric_names": ["pvalue", "l0", "psnr"], "augmentations": [], "max_roundtrips": 1, "orig_only": False}


@pytest.fixture(scope="module")
def hv():
    return np.load(os.path.join(REPO, "tests", "golden", "harness_vectors.npz"))


def _model():
    from wmar_amd.models.taming_wrapper import TamingARMMWrapper
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    gcfg, vcfg = synth.GPTConfig(**GPT), synth.VQConfig(**VQ)
    m = TamingARMMWrapper(None, gpt_cfg=gcfg, vq_cfg=vcfg, gpt_state=synth.synth_gpt_state(gcfg, seed=21, logit_scale=40.0),
                          vq_state=synth.synth_vq_state(vcfg, seed=21), max_batch=4)
    m.noise_device = "cpu"       # the fixture run had the model (hence torch.multinomial's generator) on the CPU
    wm = GentimeWatermark(m.get_vq(), m.get_total_vocab_size(), SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25,
                          device="cuda")
    m.set_watermarker(wm)
    return m, wm


def _collect(d):
    files = sorted(os.path.relpath(os.path.join(r, f), d) for r, _, fs in os.walk(d) for f in fs)
    codes, met, png = [], [], []
    for f in files:
        if f.endswith(".npy"):
            codes.append(np.load(os.path.join(d, f)))
            met.append(json.load(open(os.path.join(d, f[:-4] + ".json"))))
            png.append(np.array(Image.open(os.path.join(d, f[:-4] + ".png"))))
    return files, np.stack(codes), met, np.stack(png)


def _check(hv, tag, files, codes, met, png):
    assert files == hv[f"{tag}_files"].tolist()                                   # directory layout and names (generate.py:79-108)
    ref_codes = hv[f"{tag}_codes"]
    gen = np.array(["_roundtrips_0" in f for f in files if f.endswith(".npy")])
    assert np.array_equal(codes[gen], ref_codes[gen])                             # sampled codes: bit-exact token ids
    # re-encoded codes: argmin over 16384 codes of fp32 pixels that agree to 2e-4 -- identical up to near-ties
    assert (codes[~gen] == ref_codes[~gen]).mean() >= 0.99
    pv = np.array([m["pvalue"] for m in met])
    same = (codes == ref_codes).all(axis=1)
    assert same[gen].all()
    assert np.allclose(np.log10(pv[same]), np.log10(hv[f"{tag}_pvalue"][same]), rtol=0, atol=1e-9)
    assert np.abs(pv[same] - hv[f"{tag}_pvalue"][same]).max() < 1e-5              # north_star: p-values within 1e-5
    l0 = np.array([m["l0"] for m in met])
    assert np.array_equal(l0[same], hv[f"{tag}_l0"][same])
    psnr = np.array([m["psnr"] for m in met])
    fin = np.isfinite(hv[f"{tag}_psnr"])
    assert np.array_equal(np.isinf(psnr), ~fin)
    assert np.abs(psnr[fin & same] - hv[f"{tag}_psnr"][fin & same]).max() < 0.05   # 8-bit images that differ in a few +-1 pixels
    d8 = np.abs(png[same].astype(np.int16) - hv[f"{tag}_png"][same].astype(np.int16))
    assert d8.max() <= 1 and (d8 > 0).mean() < 2e-3


@pytest.mark.parametrize("chunk_id,num_chunks,tag", [(0, 1, "job"), (0, 2, "chunk0of2"), (1, 2, "chunk1of2")])
def test_generate_reproduces_reference_harness(hv, tmp_path, chunk_id, num_chunks, tag):
    from wmar_amd import harness
    m, wm = _model()
    assert str(wm) == str(hv["wm_str"])
    harness.seed_everything(1, chunk_id)
    recs = harness.generate(str(tmp_path), m, INPUTS, wm, EVAL, GEN, chunk_id=chunk_id, num_chunks=num_chunks)
    files, codes, met, png = _collect(str(tmp_path))
    _check(hv, tag, files, codes, met, png)
    assert len(recs) == len(met) and all(r["sample_seconds"] > 0 for r in recs)


def test_delta_checkpoints_equal_reference_update_weights(hv, tmp_path):
    """update_weights(encoder|decoder, *_delta.pth) (wmar/utils/utils.py:47-66): decode before / after the decoder patch and
    re-encode after the encoder patch equal the reference's."""
    from wmar_amd.utils.utils import update_weights
    m, _ = _model()
    vs = synth.synth_vq_state(synth.VQConfig(**VQ), seed=21)
    codes = torch.from_numpy(hv["delta_codes"]).cuda()
    img0 = m.codes_to_images(codes)
    np.testing.assert_allclose(img0.cpu().numpy(), hv["delta_img_before"], rtol=0, atol=2e-4)
    assert (m.images_to_codes(img0).cpu().numpy() == hv["delta_codes_before"]).mean() >= 0.99
    torch.save(synth.synth_delta(vs, "decoder.", seed=1), tmp_path / "dec_delta.pth")
    torch.save({"state_dict": synth.synth_delta(vs, "encoder.", seed=2)}, tmp_path / "enc_delta.pth")   # Lightning-style nesting
    r = update_weights(m.get_image_tokenizer().decoder, str(tmp_path / "dec_delta.pth"))      # the reference's call (generate.py:330-332)
    assert not r.missing_keys and not r.unexpected_keys
    img1 = m.codes_to_images(codes)
    np.testing.assert_allclose(img1.cpu().numpy(), hv["delta_img_after"], rtol=0, atol=2e-4)
    assert float(np.abs(hv["delta_img_after"] - hv["delta_img_before"]).max()) > 1e-2       # the patch is not a no-op
    update_weights(m, "encoder", str(tmp_path / "enc_delta.pth"))                              # rounds 1-4 form, still accepted
    c1 = m.images_to_codes(torch.from_numpy(hv["delta_img_after"]).cuda())
    assert (c1.cpu().numpy() == hv["delta_codes_after"]).mean() >= 0.99
    assert (hv["delta_codes_after"] != hv["delta_codes_before"]).mean() > 0.05               # the encoder patch moves codes


@pytest.mark.parametrize("chunk_id,num_chunks,tag", [(0, 1, "job"), (1, 2, "chunk1of2")])
def test_cli_reproduces_reference_harness(hv, tmp_path, chunk_id, num_chunks, tag):
    """the repo's generate.py (the reference's argparse flags, generate.py:246-287) as a subprocess on the same job."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "generate.py"), "--outdir", str(tmp_path), "--model", "taming",
                        "--synthetic", "true", "--synthetic_config", "harness", "--noise_device", "cpu", "--augmentations", "false",
                        "--num_samples_per_conditioning", "3", "--conditioning", "1,9", "--batch_size", "4", "--top_k", "250",
                        "--top_p", "0.92", "--temperature", "1.0", "--wm_method", "gentime", "--wm_seed_strategy", "linear",
                        "--wm_split_strategy", "stratifiedrand", "--wm_context_size", "1", "--wm_delta", "2.0", "--wm_gamma", "0.25",
                        "--seed", "1", "--chunk_id", str(chunk_id), "--num_chunks", str(num_chunks),
                        "--include_neural_compress", "false", "--include_diffpure", "false"],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files, codes, met, png = _collect(str(tmp_path))
    per_image = [f for f in files if f.startswith("c=")]
    _check(hv, tag, per_image, codes, met, png)
    res = json.load(open(tmp_path / "results.json"))
    assert len(res) == len(met) and {"summary.json", "results.json"} <= set(files)
