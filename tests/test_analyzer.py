"""TPR@FPR aggregation against the reference's formula evaluated inline."""
import numpy as np

from wmar_amd.utils.analyzer import roc_points, summarize, tpr_at_fpr


def test_tpr_is_the_reference_formula():
    rs = np.random.RandomState(0)
    pv = (10.0 ** rs.uniform(-8, 0, size=200)).tolist()
    assert tpr_at_fpr(pv) == np.sum(np.array(pv) < 0.01) / len(pv)      # analyzer.py:380, :424
    assert tpr_at_fpr([None, None]) == 0.0 and tpr_at_fpr([]) == 0.0
    assert tpr_at_fpr([float("nan"), 1e-9]) == 0.5                        # n_green = 0 gives NaN: not a detection
    xs, ys = roc_points([0.5, 1e-3, 0.02])
    assert xs == [1e-3, 0.02, 0.5, 1.0] and ys == [1 / 3, 2 / 3, 1.0, 1.0]


def test_summarize_groups_by_method_transform_param():
    recs = []
    for i in range(10):
        recs.append(dict(method="wm", transform="roundtrips", param=1, metrics=dict(pvalue=1e-5 if i < 7 else 0.3, l0=0.25, psnr=30.0)))
        recs.append(dict(method="wm", transform="jpeg", param=25, metrics=dict(pvalue=0.2, l0=0.5, psnr=float("inf"))))
        recs.append(dict(method="None", transform="roundtrips", param=1, metrics=dict(pvalue=None, l0=0.0, psnr=31.0)))
    s = summarize(recs)
    assert s["wm|roundtrips_1"]["tpr"] == 0.7 and s["wm|roundtrips_1"]["n"] == 10 and s["wm|roundtrips_1"]["l0"] == 0.25
    assert s["wm|jpeg_25"]["tpr"] == 0.0 and s["wm|jpeg_25"]["psnr"] is None
    assert s["None|roundtrips_1"]["tpr"] == 0.0 and s["None|roundtrips_1"]["log10_p_median"] is None
