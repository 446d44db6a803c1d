"""Drop-in for ``wmar.watermarking.gentime_watermark`` backed by libwmar_hip.so.

Same constructor, ``__str__``, ``spawn_logit_processor``, ``detect`` and
``create_watermarker_from_string`` as the reference
(wmar/watermarking/gentime_watermark.py:95-366).  What changes is HOW:

* the key is derived once, on the host, as a bitmap table (one row per context
  sum) by ``wmar_key_table_build`` and kept resident in HBM -- the reference
  re-seeds an MT19937 and draws two ``randperm``s per row per decode step;
* ``_process_logits`` is one kernel launch (no ``.item()`` sync, no Python loop);
* ``detect`` is one kernel launch for the whole batch.

The device functions need an MI355X (``device="cuda..."``); there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import collections
import os
import hashlib
from enum import Enum
from functools import partial
from typing import Union

import numpy as np
import torch

from .. import _lib


class SeedStrategy(Enum):
    FIXED = "fixed"
    LINEAR = "linear"
    SPATIAL = "spatial"


class SplitStrategy(Enum):
    RANDOM = "rand"
    RANDOM_STRATIFIED = "stratifiedrand"
    CLUSTERING = "clustering"


MAX_CONTEXT = 16      # include/wmar_hip.h WMAR_MAX_CONTEXT (LINEAR / FIXED; SPATIAL: 1 or 3 as in the reference)
_SEED_CODE = {SeedStrategy.FIXED: 0, SeedStrategy.LINEAR: 1, SeedStrategy.SPATIAL: 2}
_SPLIT_CODE = {SplitStrategy.RANDOM: 0, SplitStrategy.RANDOM_STRATIFIED: 1}

# Device key tables shared by watermarkers with the same key (a sweep over delta re-uses one table).  Bounded: a sweep over
# gamma / h / salt would otherwise pin tens to hundreds of MiB of HBM per configuration for the life of the process.  An evicted
# table stays alive as long as a watermarker still references it.
_TABLE_CACHE: "collections.OrderedDict" = collections.OrderedDict()
_TABLE_CACHE_MAX = 4


def clear_key_cache():
    """Drop every cached device key table (they are rebuilt on next use)."""
    _TABLE_CACHE.clear()


def _cache_put(key, table):
    _TABLE_CACHE[key] = table
    _TABLE_CACHE.move_to_end(key)
    while len(_TABLE_CACHE) > _TABLE_CACHE_MAX:
        _TABLE_CACHE.popitem(last=False)


def _is_cuda(device) -> bool:
    return torch.device(device).type == "cuda"


class GentimeWatermark:
    def __init__(
        self,
        vq: Union[object, dict],
        vocab_size: int,
        seed_strategy: SeedStrategy,
        split_strategy: SplitStrategy,
        context_size: int,
        delta: float,
        gamma: float,
        device="cpu",
        spatial_dim=16,
        salt_key=15485863,
    ) -> None:
        self.device = device
        self.vocab_size = vocab_size
        if isinstance(vq, dict):
            alive, dead, emb = vq["alive_ids"], vq["dead_ids"], vq.get("embedding")
        else:
            alive, dead = vq.alive_ids, vq.dead_ids
            emb = vq.embedding.weight if hasattr(vq, "embedding") else None
        self.alive_ids = alive.to(device)
        self.dead_ids = dead.to(device)
        self.embedding = emb
        self.embedding_dim = emb.shape[1] if emb is not None else None
        self._alive_host = np.ascontiguousarray(alive.detach().cpu().numpy().astype(np.int64))
        self._dead_host = np.ascontiguousarray(dead.detach().cpu().numpy().astype(np.int64))

        self.salt_key = salt_key
        self.seed_strategy = seed_strategy
        self.split_strategy = split_strategy
        self.context_size = context_size
        self.delta = delta
        self.gamma = gamma
        self.greenlist_size = int(self.vocab_size * self.gamma)
        self.spatial_dim = spatial_dim
        if split_strategy is SplitStrategy.CLUSTERING:
            raise NotImplementedError("SplitStrategy.CLUSTERING (TSNE+KMeans split) is outside the MI355X hot path")
        if seed_strategy is SeedStrategy.SPATIAL and context_size not in (1, 3):
            raise AssertionError("Spatial seeding only implemented for context size in [1,3]")
        if not 0 <= context_size <= MAX_CONTEXT:
            raise NotImplementedError(f"context sizes 0..{MAX_CONTEXT} are supported (the key table has context_size * (vocab - 1) + 1 rows)")
        # The key is a table of one greenlist bitmap per context SUM: context_size * (vocab - 1) + 1 rows of vocab / 8 bytes, built
        # eagerly on the host and kept in HBM -- 32 MiB for Taming h = 1, 0.5 GiB at h = 16, 8.6 GiB for a 65536-entry vocabulary at
        # h = 16.  Refuse a key beyond the budget here, with the numbers, instead of at the first allocation.
        rows = 1 if seed_strategy is SeedStrategy.FIXED else context_size * (int(vocab_size) - 1) + 1
        self.key_table_bytes = rows * ((int(vocab_size) + 31) // 32) * 4
        budget = float(os.environ.get("WMAR_MAX_KEY_TABLE_GB", "4")) * (1 << 30)
        if self.key_table_bytes > budget:
            raise NotImplementedError(
                f"the key table of {seed_strategy.value} seeding with context_size {context_size} over a vocabulary of {vocab_size} has {rows} rows "
                f"= {self.key_table_bytes / (1 << 30):.1f} GiB, above the budget of {budget / (1 << 30):.1f} GiB (WMAR_MAX_KEY_TABLE_GB); "
                f"use a smaller context_size")
        self._table = None
        if self.seed_strategy == SeedStrategy.FIXED:
            self.fixed_greenlist = self._split_with_seed(0)
        else:
            self.fixed_greenlist = None

    def __str__(self):
        ret = f"{self.seed_strategy.value}-{self.split_strategy.value}-"
        ret += f"h={self.context_size}-d={self.delta:.1f}-g={self.gamma:.2f}"
        return ret

    # ------------------------------------------------------------------ key (host side)
    def _key_params(self) -> _lib.KeyParams:
        kp = _lib.KeyParams()
        kp.salt_key = self.salt_key
        kp.alive_ids = self._alive_host.ctypes.data_as(C.POINTER(C.c_int64))
        kp.n_alive = len(self._alive_host)
        kp.dead_ids = self._dead_host.ctypes.data_as(C.POINTER(C.c_int64))
        kp.n_dead = len(self._dead_host)
        kp.vocab_size = self.vocab_size
        kp.gamma = self.gamma
        kp.split_strategy = _SPLIT_CODE[self.split_strategy]
        kp.seed_strategy = _SEED_CODE[self.seed_strategy]
        return kp

    def _split_with_seed(self, seed: int) -> torch.LongTensor:
        """Greenlist ids in the reference's order (gentime_watermark.py:161-174)."""
        L = _lib.load()
        out = np.zeros(self.vocab_size + 8, dtype=np.int64)
        kp = self._key_params()
        n = L.wmar_key_greenlist(C.byref(kp), C.c_uint64(seed % (2**64)), out.ctypes.data)
        if n < 0:
            _lib.check(int(n))
        return torch.from_numpy(out[:n].copy()).to(self.device)

    def _get_greenlist_ids_for_context(self, context: torch.LongTensor):
        assert context.ndim <= 1, "context must be a non-batched tensor"
        assert len(context) == self.context_size, f"context must be of length {self.context_size}"
        if self.seed_strategy is SeedStrategy.FIXED:
            return self.fixed_greenlist
        seed = (self.salt_key * int(context.sum().item())) % (2**64 - 1)
        return self._split_with_seed(seed)

    def key_table_host(self, n_rows=None) -> np.ndarray:
        """uint32 [n_rows, vocab/32] bitmap table built by the library's host builder."""
        L = _lib.load()
        if n_rows is None:
            n_rows = L.wmar_key_table_rows(_SEED_CODE[self.seed_strategy], self.context_size, self.vocab_size)
        words = L.wmar_key_row_words(self.vocab_size)
        out = np.zeros((n_rows, words), dtype=np.uint32)
        kp = self._key_params()
        _lib.check(L.wmar_key_table_build(C.byref(kp), 0, n_rows, out.ctypes.data, 0))
        return out

    def _cache_key(self):
        h = hashlib.sha1(self._alive_host.tobytes() + b"|" + self._dead_host.tobytes()).hexdigest()
        return (h, self.vocab_size, self.gamma, self.salt_key, self.split_strategy, self.seed_strategy,
                self.context_size, str(torch.device(self.device)))

    def key_table(self) -> torch.Tensor:
        """The device-resident key (built once per key, cached per process)."""
        if self._table is None:
            if not _is_cuda(self.device):
                raise RuntimeError("wmar_amd.GentimeWatermark device functions need device='cuda' (MI355X); "
                                   "there is no CPU implementation")
            ck = self._cache_key()
            if ck not in _TABLE_CACHE:
                host = torch.from_numpy(self.key_table_host().view(np.int32))
                _cache_put(ck, host.to(self.device))
            else:
                _TABLE_CACHE.move_to_end(ck)
            self._table = _TABLE_CACHE[ck]
        return self._table

    def set_key_table(self, table: torch.Tensor):
        """Install a table received from another rank (RCCL broadcast) instead of rebuilding it."""
        self._table = table
        _cache_put(self._cache_key(), table)

    def wm_ctx(self) -> _lib.WmCtx:
        t = self.key_table()
        c = _lib.WmCtx()
        c.table_dev = t.data_ptr()
        c.n_rows = t.shape[0]
        c.vocab_size = self.vocab_size
        c.seed_strategy = _SEED_CODE[self.seed_strategy]
        c.context_size = self.context_size
        c.spatial_dim = self.spatial_dim
        c.delta = float(self.delta)
        return c

    # ---------------------------------------------------------------- logit processor
    # past_ids: [B, len], logits: [B, vocab_size]; mutates and returns logits
    def _process_logits(self, past_ids: torch.LongTensor, logits: torch.Tensor) -> torch.Tensor:
        assert logits.shape[-1] == self.vocab_size, f"Logits shape mismatch: {logits.shape} vs {self.vocab_size}"
        if not logits.is_cuda:
            raise RuntimeError("wmar_amd: logits must live on the MI355X (no CPU implementation)")
        assert logits.dtype == torch.float32 and logits.dim() == 2
        B = past_ids.shape[0]
        work = logits if logits.is_contiguous() else logits.contiguous()
        past = past_ids.to(device=logits.device, dtype=torch.int64)
        if past.dim() == 2 and past.stride(1) != 1:
            past = past.contiguous()
        t = past.shape[1] if past.dim() == 2 else 0
        stride = past.stride(0) if (past.dim() == 2 and B > 0 and t > 0) else max(t, 1)
        ctx = self.wm_ctx()
        with torch.cuda.device(logits.device):
            _lib.check(_lib.load().wmar_wm_process_logits(
                C.byref(ctx), work.data_ptr(), B, past.data_ptr() if t > 0 else None, t, stride,
                _lib.stream_ptr(logits.device)))
        if work is not logits:
            logits.copy_(work)
        return logits

    def spawn_logit_processor(self):
        return partial(self._process_logits)

    # ------------------------------------------------------------------------ detector
    def detect_counts(self, codes: torch.LongTensor, return_masks: bool = False):
        """(pvals f64[B], n_scored i32[B], n_green i32[B][, masks int8[B, h+n_ngrams]]) on the device."""
        L = _lib.load()
        if not _is_cuda(self.device):
            raise RuntimeError("wmar_amd.GentimeWatermark.detect needs device='cuda' (MI355X)")
        codes = codes.to(device=self.device, dtype=torch.int64).contiguous()
        assert codes.dim() == 2
        B, Lc = codes.shape
        dev = codes.device
        pv = torch.empty(B, dtype=torch.float64, device=dev)
        ns = torch.empty(B, dtype=torch.int32, device=dev)
        ng = torch.empty(B, dtype=torch.int32, device=dev)
        masks = None
        mstride = 0
        if return_masks and Lc - self.context_size >= 1:
            nn = L.wmar_detect_num_ngrams(_SEED_CODE[self.seed_strategy], self.context_size, Lc)
            if nn < 0:
                raise AssertionError("Sequence must be a square")
            mstride = self.context_size + nn
            masks = torch.empty(B, mstride, dtype=torch.int8, device=dev)
        ctx = self.wm_ctx()
        with torch.cuda.device(dev):
            _lib.check(L.wmar_detect(C.byref(ctx), float(self.gamma), codes.data_ptr(), B, Lc, ns.data_ptr(),
                                     ng.data_ptr(), pv.data_ptr(), masks.data_ptr() if masks is not None else None,
                                     mstride, _lib.stream_ptr(dev)))
        return (pv, ns, ng, masks) if return_masks else (pv, ns, ng)

    # codes: [B, len] of ids in [0, vocab_size-1]; returns p-values of greenlist hits
    def detect(self, codes: torch.LongTensor, return_masks: bool = False):
        if return_masks:
            pv, _, _, masks = self.detect_counts(codes, True)
            return pv.to(self.device), masks.cpu().tolist()
        return self.detect_counts(codes)[0].to(self.device)


# For example: fixed-stratifiedrand-h=0-d=8.0-g=0.50
def create_watermarker_from_string(vq, vocab_size: int, method: str, device: str) -> GentimeWatermark:
    parts = method.split("-")
    return GentimeWatermark(
        vq,
        vocab_size,
        SeedStrategy(parts[0]),
        SplitStrategy(parts[1]),
        int(parts[2].split("=")[1]),
        float(parts[3].split("=")[1]),
        float(parts[4].split("=")[1]),
        device=device,
    )
