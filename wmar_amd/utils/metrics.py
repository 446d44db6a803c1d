"""Mirror of ``wmar.utils.metrics`` (wmar/utils/metrics.py:20-45): pvalue / l0 / psnr.
``bpp`` belongs to the neural-compression attacks, which are outside the hot path."""
from __future__ import annotations

import numpy as np
import torch


# compute psnr between two PIL images
def compute_psnr(a, b, M=255.0):
    mse = np.mean((np.array(a) * 1.0 - np.array(b) * 1.0) ** 2)
    return 10 * np.log10(M**2 / mse)


def compute_metric(metric_name, code, orig_code, img, orig_img, watermarker, transform, param, compressors=None):
    if metric_name == "bpp":
        return None
    elif metric_name == "l0":
        return (orig_code != code).sum().item() / orig_code.shape[0]
    elif metric_name == "psnr":
        return compute_psnr(img, orig_img)
    else:
        if watermarker is None:
            return None
        if metric_name == "pvalue":
            return watermarker.detect(torch.LongTensor(code.reshape(1, -1)).to(watermarker.device)).item()
        raise ValueError(f"Metric {metric_name} not found")
