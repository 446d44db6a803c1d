import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def kat():
    import json
    return json.load(open(os.path.join(REPO, "tests", "golden", "key_kat.json")))


def load_ids(name):
    ids = []
    for line in open(os.path.join(REPO, "wmar_amd", "assets", name)):
        ids.extend(int(t) for t in line.split(","))
    return ids


@pytest.fixture(scope="session")
def key_factory():
    """Builds oracle KeyParams for the named golden key configs."""
    from oracle import wm_oracle as W

    def make(cfg, **over):
        alive = load_ids(cfg["alive"])
        dead = sorted(set(range(cfg["vocab"])) - set(alive))
        kw = dict(split=cfg["split"], seed=cfg["seed"], context_size=cfg["h"])
        kw.update(over)
        return W.KeyParams(alive, dead, cfg["vocab"], cfg["gamma"], **kw)

    return make
