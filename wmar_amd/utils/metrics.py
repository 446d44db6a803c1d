"""Per-image metrics of the evaluation harness; same call signature and values as ``wmar.utils.metrics.compute_metric``
(wmar/utils/metrics.py:20-45): ``pvalue`` (detector on one code row), ``l0`` (fraction of changed codes), ``psnr`` (on 8-bit images).
``bpp`` belongs to the neural-compression attacks, which this build does not carry: it evaluates to None."""
from __future__ import annotations

import numpy as np
import torch


def compute_psnr(a, b, M=255.0):
    """PSNR in dB between two PIL images (or uint8 arrays); identical images give +inf like the reference."""
    diff = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    mse = np.mean(diff * diff)          # stays a numpy scalar: 0 divides to +inf (a Python float would raise)
    with np.errstate(divide="ignore"):
        return 10 * np.log10(M ** 2 / mse)


def _l0(code, orig_code):
    changed = int((np.asarray(orig_code) != np.asarray(code)).sum())
    return changed / orig_code.shape[0]


def _pvalue(code, watermarker):
    row = torch.as_tensor(np.asarray(code).reshape(1, -1), dtype=torch.long, device=watermarker.device)
    return watermarker.detect(row).item()


def compute_metric(metric_name, code, orig_code, img, orig_img, watermarker, transform, param, compressors=None):
    if metric_name == "bpp":
        return None
    if metric_name == "l0":
        return _l0(code, orig_code)
    if metric_name == "psnr":
        return compute_psnr(img, orig_img)
    if watermarker is None:          # detector metrics without a watermarker are undefined, not errors
        return None
    if metric_name == "pvalue":
        return _pvalue(code, watermarker)
    raise ValueError(f"Metric {metric_name} not found")
