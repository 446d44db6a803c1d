"""Aggregation of per-image result records into the reference's headline numbers:
TPR at a fixed false-positive rate -- the fraction of images whose detector p-value is below the rate, what
``Analyzer.plot_robustness`` computes per (method, transform, parameter) as ``np.sum(pvals < 0.01) / len(pvals)``
(wmar/utils/analyzer.py:376-381, 419-424) -- the ROC points of ``plot_auc`` (:247-262), mean l0 / PSNR.
Works on the records ``wmar_amd.harness.generate`` returns (and gathers across ranks), so no json round trip is needed;
plotting is not part of this build."""
from __future__ import annotations

import math
from collections import defaultdict
from typing import Dict, Iterable, List, Tuple

import numpy as np


def tpr_at_fpr(pvalues: Iterable, fpr: float = 0.01) -> float:
    """Fraction of p-values strictly below ``fpr``; 0 when the run carried no detector (p-values are None), NaNs never count."""
    pv = list(pvalues)
    if not pv or pv[0] is None:
        return 0.0
    arr = np.asarray(pv, dtype=np.float64)
    return float(np.sum(arr < fpr) / len(arr))


def roc_points(pvalues: Iterable) -> Tuple[List[float], List[float]]:
    """(false-positive rate, true-positive rate) steps of the empirical ROC: the i-th smallest p-value detects (i+1)/N images."""
    pv = sorted(float(p) for p in pvalues)
    xs = pv + [1.0]
    ys = [(i + 1) / len(pv) for i in range(len(pv))]
    return xs, ys + [ys[-1] if ys else 0.0]


def summarize(records: List[dict], fpr: float = 0.01) -> Dict[str, dict]:
    """{"<method>|<transform>_<param>": {n, tpr, l0, psnr, log10_p_median}} over result records
    (keys: method, transform, param, metrics{pvalue, l0, psnr})."""
    groups = defaultdict(list)
    for r in records:
        groups[(r["method"], r["transform"], str(r["param"]))].append(r["metrics"])
    out = {}
    for (method, transform, param), ms in sorted(groups.items()):
        pv = [m.get("pvalue") for m in ms]
        have_p = bool(pv) and pv[0] is not None
        finite = [p for p in pv if p is not None and not (isinstance(p, float) and math.isnan(p))] if have_p else []
        psnr = [m["psnr"] for m in ms if m.get("psnr") is not None and math.isfinite(m["psnr"])]
        l0 = [m["l0"] for m in ms if m.get("l0") is not None]
        out[f"{method}|{transform}_{param}"] = {
            "n": len(ms),
            "tpr": tpr_at_fpr(pv, fpr),
            "l0": float(np.mean(l0)) if l0 else None,
            "psnr": float(np.mean(psnr)) if psnr else None,
            "log10_p_median": float(np.median(np.log10(np.maximum(finite, 1e-300)))) if finite else None,
        }
    return out


def load_results_dir(root: str, resultdir_prefix: str, method_id: str) -> Tuple[Dict[str, list], Dict[str, list], int]:
    """``Analyzer.get_metrics_imagepaths_N`` (wmar/utils/analyzer.py:186-238) for one method: walk
    ``root/<dirs starting with resultdir_prefix>/c=<class>,idx=<n>/`` as generate.py writes them (generate.py:79-108), collect the
    metric jsons named ``<idx>_<method>_<transform>_<param>.json`` into ``{"<transform>_<param>": [metrics, ...]}`` and the
    ``roundtrips_0`` PNG paths per class.  Returns (all_metrics, all_orig_image_paths, N = number of result directories)."""
    import json
    import os

    dirs = [d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)) and d.startswith(resultdir_prefix)]
    result_dirs = []
    for d in dirs:
        result_dirs += [os.path.join(root, d, s) for s in os.listdir(os.path.join(root, d)) if os.path.isdir(os.path.join(root, d, s))]
    all_metrics: Dict[str, list] = {}
    all_orig: Dict[str, list] = {}
    for rd in result_dirs:
        toks = [t.split("=")[-1] for t in os.path.basename(rd).split(",")]
        cls = toks[0]
        for base, _, files in os.walk(rd):
            for f in files:
                stem, ext = os.path.splitext(f)
                parts = stem.split("_")
                if len(parts) != 4 or parts[1] != method_id:
                    continue
                _, _, aug, param = parts
                if ext == ".json":
                    with open(os.path.join(base, f)) as fh:
                        all_metrics.setdefault(f"{aug}_{param}", []).append(json.load(fh))
                elif ext == ".png" and aug == "roundtrips" and param == "0":
                    all_orig.setdefault(cls, []).append(os.path.join(base, f))
    return all_metrics, all_orig, len(result_dirs)


def tpr_table(all_metrics: Dict[str, list], fpr: float = 0.01) -> Dict[str, float]:
    """TPR at `fpr` per "<transform>_<param>" of a ``load_results_dir`` dictionary (the rule of plot_robustness)."""
    return {k: tpr_at_fpr([m.get("pvalue") for m in v], fpr) for k, v in sorted(all_metrics.items())}
