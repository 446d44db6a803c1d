"""wmar_comm_* (include/wmar_hip.h): the RCCL wrappers of the sharded job's exchange step, on the GPU box's one GPU -- a
communicator of world size 1 runs the real ncclCommInitRank / ncclBroadcast / ncclAllGather of the RCCL the process carries (the
broadcast and the gather of one rank are copies), through the ctypes binding a host would use (wmar_amd/comm.py).  The N > 1
behaviour of the SAME harness code is covered on CPU by tests/test_distributed_cpu.py (gloo, 2 and 8 ranks); a two-GPU run is
scripts/smoke_2gpu.sh (WMAR_COMM=rccl switches the harness to this ABI)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_wrappers_world_1_roundtrip():
    from wmar_amd.comm import ID_BYTES, RcclComm
    uid = RcclComm.unique_id()
    assert len(uid) == ID_BYTES and any(uid)
    c = RcclComm(uid, 0, 1, "cuda")
    assert (c.rank, c.world) == (0, 1)
    table = torch.arange(16384 * 512, dtype=torch.int32, device="cuda").view(16384, 512)       # the Taming key table's size: 32 MiB
    want = table.clone()
    c.broadcast_(table, 0)
    codes = torch.randint(0, 16384, (64, 256), device="cuda")
    pv = torch.rand(64, dtype=torch.float64, device="cuda")
    (g_codes,), (g_pv,) = c.all_gather(codes), c.all_gather(pv)
    torch.cuda.synchronize()
    assert torch.equal(table, want) and torch.equal(g_codes, codes) and torch.equal(g_pv, pv)


def test_harness_exchange_through_the_abi(monkeypatch):
    """WMAR_COMM=rccl: harness.broadcast_key_table / gather_records run on the wmar_comm_* entry points (a 1-rank "nccl" group)."""
    import os
    import socket

    import torch.distributed as dist
    from wmar_amd import comm, harness
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    monkeypatch.setenv("WMAR_COMM", "rccl")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert comm.active("cuda") is not None

        class WM:
            def __init__(self): self.t = None
            def key_table(self): return torch.arange(4096, dtype=torch.int32, device="cuda").view(64, 64)
            def set_key_table(self, t): self.t = t
        wm = WM()
        harness.broadcast_key_table(wm, "cuda")
        assert torch.equal(wm.t, wm.key_table())
        ev = {"metric_names": ["pvalue", "l0"], "augmentations": [], "max_roundtrips": 1, "orig_only": False}
        recs = [dict(conditioning=7, idx=i, method="m", transform="roundtrips", param=i % 2, metrics={"pvalue": 0.25 * i, "l0": 0.5},
                     codes=np.arange(16, dtype=np.int64) + i, batch_idx=0, sample_seconds=0.1) for i in range(4)]
        out = harness.gather_records(recs, ev, torch.device("cuda"))
        assert [r["idx"] for r in out] == [0, 1, 2, 3] and all(np.array_equal(r["codes"], np.arange(16) + r["idx"]) for r in out)
        assert [r["metrics"]["pvalue"] for r in out] == [0.0, 0.25, 0.5, 0.75]
    finally:
        dist.destroy_process_group()
        comm._COMM = None
