import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def usable_cores() -> int:
    """Host threads for the CPU oracle: scheduler affinity capped by the cgroup quota and by 32 (bench.py's rule -- torch's default
    of one thread per visible core oversubscribes a quota-limited container: measured 200x slower on the GPU box's host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(usable_cores())
    except Exception:
        pass


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
