"""bench.py's world_size > 1 control flow executed for real on CPU: two processes under torch.distributed.run with the gloo
backend and a stand-in engine (tests/bench_stub.py) -- process group, key-table broadcast from rank 0, barriers, the per-step
all_gather of codes / counts / p-values, max-over-ranks timing, one JSON line from rank 0.  On the GPU box the same code runs
with backend "nccl" (RCCL over xGMI) and the HIP engines."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_bench_two_ranks_gloo():
    env = dict(os.environ, WMAR_BENCH_BACKEND="gloo", WMAR_BENCH_ENGINE="tests.bench_stub:make", PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=240, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "images/s" and out["value"] > 0 and out["higher_is_better"] is True
    assert abs(out["value"] - 2 * 64 * 2 / (out["ms_per_step"] * 2 / 1e3)) / out["value"] < 0.01     # whole-job aggregate
    assert out["gathered"] == {"codes": [128, 256], "pvalues": 128}                                 # both ranks' results on rank 0
    assert out["detector"]["n_green_mean"] == 100.0      # rank 1 scored with rank 0's key table: the broadcast happened
    assert out["config"]["backend"] == "gloo"


def test_bench_refuses_to_run_the_product_path_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=240, cwd=REPO, env=dict(os.environ, WMAR_BENCH_BACKEND="gloo"))
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout)
