"""RCCL exchange step through the C ABI (include/wmar_hip.h: wmar_comm_*): the key-table broadcast and the per-step gather of the
sharded job (the reference shards by chunk processes and exchanges nothing, generate.py:204, :304).

The default host path reaches the same RCCL through ``torch.distributed`` (backend "nccl" IS RCCL on ROCm); ``WMAR_COMM=rccl``
makes ``harness.broadcast_key_table`` / ``harness.gather_records`` use this binding instead -- the form a non-Python host binds.
RCCL's 128-byte unique id is made on rank 0 and handed out by the host: here over the already-initialised torch process group's
store (any channel works: a file, the environment)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib

ID_BYTES = 128


class RcclComm:
    """One communicator per process (rank = GPU).  Device tensors in, device tensors out, asynchronous on the current stream."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device="cuda"):
        assert len(unique_id) >= ID_BYTES
        self.device = torch.device(device)
        self._L = _lib.load()
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id[:ID_BYTES]), ID_BYTES)
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_comm_init(buf, ID_BYTES, int(rank), int(world), C.byref(h)))
        self._h = h
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id() -> bytes:
        L = _lib.load()
        buf = C.create_string_buffer(ID_BYTES)
        _lib.check(L.wmar_comm_unique_id(buf, ID_BYTES))
        return buf.raw

    @classmethod
    def from_torch_group(cls, device="cuda") -> "RcclComm":
        """rank / world / id exchange from the initialised torch.distributed group (the id travels as a byte tensor)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        on_gpu = dist.get_backend() == "nccl"
        t = torch.zeros(ID_BYTES, dtype=torch.uint8, device=device if on_gpu else "cpu")
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        dist.broadcast(t, 0)
        return cls(bytes(t.cpu().numpy().tobytes()), rank, world, device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.wmar_comm_destroy(h)
            self._h = None

    def broadcast_(self, t: torch.Tensor, root: int = 0) -> torch.Tensor:
        assert t.is_cuda and t.is_contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_comm_bcast(self._h, t.data_ptr(), t.numel() * t.element_size(), int(root), _lib.stream_ptr(self.device)))
        return t

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        """[world] tensors shaped like t, rank order (every rank sends the same shape)."""
        assert t.is_cuda
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_comm_allgather(self._h, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(),
                                                    _lib.stream_ptr(self.device)))
        return list(out.unbind(0))


_COMM: Optional[RcclComm] = None


def active(device) -> Optional[RcclComm]:
    """The process's communicator when WMAR_COMM=rccl and the job runs on GPUs with an initialised process group; else None."""
    import os

    import torch.distributed as dist
    global _COMM
    if os.environ.get("WMAR_COMM", "") != "rccl" or not dist.is_initialized() or torch.device(device).type != "cuda":
        return None
    if _COMM is None:
        _COMM = RcclComm.from_torch_group(device)
    return _COMM
