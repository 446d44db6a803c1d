"""f3: the result-directory reader and the TPR rule against the reference's own ``Analyzer.get_metrics_imagepaths_N`` run on the
reference harness's output directory (wmar/utils/analyzer.py:186-238; TPR rule :376-381; fixture: tests/golden/harness_vectors.npz
keys ``an_*``, made by make_golden.py harness_vectors)."""
import json
import os

import numpy as np
import pytest

from tests.conftest import REPO


@pytest.fixture(scope="module")
def hv():
    return np.load(os.path.join(REPO, "tests", "golden", "harness_vectors.npz"))


def _rebuild(hv, root):
    """the reference harness's directory, rebuilt from the fixture (file names + the metrics each json held)."""
    files = hv["job_files"].tolist()
    jsons = [f for f in files if f.endswith(".json")]
    npys = [f[:-4] for f in files if f.endswith(".npy")]
    for f in files:
        os.makedirs(os.path.join(root, "run_a", os.path.dirname(f)), exist_ok=True)
        open(os.path.join(root, "run_a", f), "wb").close()
    for i, stem in enumerate(npys):
        m = {"pvalue": float(hv["job_pvalue"][i]), "l0": float(hv["job_l0"][i]), "psnr": float(hv["job_psnr"][i])}
        json.dump(m, open(os.path.join(root, "run_a", stem + ".json"), "w"))
    assert len(jsons) == len(npys)
    os.makedirs(os.path.join(root, "other_run", "c=1,idx=1"))          # a directory the prefix must filter out


def test_results_dir_reader_and_tpr_equal_reference(hv, tmp_path):
    from wmar_amd.utils.analyzer import load_results_dir, tpr_table
    _rebuild(hv, str(tmp_path))
    metrics, orig, N = load_results_dir(str(tmp_path), "run_", str(hv["wm_str"]))
    assert N == int(hv["an_N"]) and sorted(metrics) == hv["an_keys"].tolist()
    for k in metrics:
        assert np.array_equal(np.sort([m["pvalue"] for m in metrics[k]]), hv[f"an_{k}_pvalues_sorted"])
        assert np.array_equal(np.sort([m["l0"] for m in metrics[k]]), hv[f"an_{k}_l0_sorted"])
        for thr in (0.01, 0.25):
            assert tpr_table(metrics, thr)[k] == float(hv[f"an_{k}_tpr_at_{thr}"])
    assert sorted(orig) == hv["an_orig_image_classes"].tolist()
    assert [len(orig[c]) for c in sorted(orig)] == hv["an_orig_images_per_class"].tolist()
    assert load_results_dir(str(tmp_path), "run_", "some-other-method")[0] == {}


def test_summarize_matches_the_same_rule(hv):
    from wmar_amd.utils.analyzer import summarize
    files = [f[:-4] for f in hv["job_files"].tolist() if f.endswith(".npy")]
    recs = []
    for i, stem in enumerate(files):
        _, method, transform, param = os.path.basename(stem).split("_")
        recs.append(dict(method=method, transform=transform, param=int(param),
                         metrics={"pvalue": float(hv["job_pvalue"][i]), "l0": float(hv["job_l0"][i]), "psnr": float(hv["job_psnr"][i])}))
    s = summarize(recs, 0.25)
    for k in hv["an_keys"].tolist():
        assert s[f"{hv['wm_str']}|{k}"]["tpr"] == float(hv[f"an_{k}_tpr_at_0.25"])
        assert s[f"{hv['wm_str']}|{k}"]["n"] == int(hv["an_N"])
