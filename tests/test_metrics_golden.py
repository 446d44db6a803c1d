"""A13: ``compute_metric`` / ``chw_to_pillow`` / ``compute_psnr`` against the reference's own outputs
(wmar/utils/metrics.py:20-45, wmar/utils/utils.py:74-80), from tests/golden/harness_vectors.npz (made by running the reference's
functions in the build container: tests/golden/make_golden.py harness_vectors)."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import REPO


@pytest.fixture(scope="module")
def hv():
    return np.load(os.path.join(REPO, "tests", "golden", "harness_vectors.npz"))


def test_chw_to_pillow_rounding_and_clip(hv):
    """uint8 conversion: clip to [0,255] BEFORE rounding, round-half-even on the k + 0.5 ticks, HWC layout."""
    from wmar_amd.utils.utils import chw_to_pillow
    a = np.array(chw_to_pillow(hv["met_img_a"]))
    b = np.array(chw_to_pillow(torch.from_numpy(hv["met_img_b"])))       # tensor input path
    assert a.dtype == np.uint8 and a.shape == (32, 32, 3)
    assert np.array_equal(a, hv["met_u8_a"]) and np.array_equal(b, hv["met_u8_b"])


def test_l0_psnr_bpp_equal_reference(hv):
    from wmar_amd.utils.metrics import compute_metric, compute_psnr
    from wmar_amd.utils.utils import chw_to_pillow
    pa, pb = chw_to_pillow(hv["met_img_a"]), chw_to_pillow(hv["met_img_b"])
    a, b = hv["met_code_a"], hv["met_code_b"]
    assert compute_metric("l0", b, a, pb, pa, None, "roundtrips", 0) == float(hv["met_l0"])
    assert compute_metric("psnr", b, a, pb, pa, None, "roundtrips", 0) == pytest.approx(float(hv["met_psnr"]), rel=1e-12)
    assert np.isinf(compute_psnr(pa, pa)) and np.isinf(float(hv["met_psnr_same"]))
    assert compute_metric("bpp", b, a, pb, pa, None, "roundtrips", 0) is None and float(hv["met_bpp"]) == -1.0
    assert compute_metric("pvalue", b, a, pb, pa, None, "roundtrips", 0) is None       # no watermarker: None, not an error
    with pytest.raises(ValueError):
        compute_metric("nope", b, a, pb, pa, object(), "roundtrips", 0)


def test_oracle_pvalue_equals_reference_metric(hv, kat, key_factory):
    """the 'pvalue' metric is detect() on one code row (metrics.py:43): the CPU oracle reproduces the reference's value."""
    from oracle import wm_oracle as W
    key = key_factory(kat["keys"]["taming"])
    pv, _, _ = W.detect(key, hv["met_code_b"].reshape(1, -1))
    assert pv[0] == pytest.approx(float(hv["met_pvalue"]), rel=1e-9)
    pv, _, _ = W.detect(key, hv["job_codes"])
    assert np.allclose(pv, hv["job_pvalue"], rtol=1e-9, atol=0)


@pytest.mark.gpu
def test_pvalue_metric_on_device(hv, kat):
    from tests.test_gpu_watermark import _wm
    from wmar_amd.utils.metrics import compute_metric
    wm = _wm(kat["keys"]["taming"])
    got = compute_metric("pvalue", hv["met_code_b"], hv["met_code_a"], None, None, wm, "roundtrips", 0)
    assert got == pytest.approx(float(hv["met_pvalue"]), rel=1e-9)
