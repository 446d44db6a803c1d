"""Gumbel-key ("Aaronson") watermark on MI355X -- SURVEY.md section 8a row G1.

Mirrors ``wmar_audio/watermark/engine.py`` (``get_wm_window_hash`` :13-26, ``gumbel_sample`` :29-75,
``gumbel_score_tok`` :123-134): same names, arguments and return dtypes, tensors on the GPU.  The
reference's image code never calls them; ``GumbelWatermark`` (RAR + Gumbel key, BASELINE config 3)
is therefore an extension: fixed key (``ngram = 0``), detector = sum of the per-token scores
``-log(1 - rs[token])`` against their Gamma(L, 1) null distribution.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from .. import _lib


def get_wm_window_hash(ngrams: torch.Tensor = None, seed: int = 0) -> torch.Tensor:
    """engine.py:13-26.  ``ngram == 0``: the hash is the seed.  ``ngram > 0`` raises TypeError in the reference
    (``GENERATOR=`` keyword, :23); the evident intent -- first randint of the seeded generator xor the tokens --
    is what runs here."""
    batch_size, wm_ngram = ngrams.shape
    if wm_ngram == 0:
        return torch.full((batch_size,), seed, dtype=torch.int64)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    h0 = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g).item())
    out = torch.full((batch_size,), h0, dtype=torch.int64)
    ng = ngrams.detach().to("cpu", torch.int64)
    for ii in range(wm_ngram):
        out ^= ng[:, ii]
    return out


_KEYS: Dict[Tuple[int, int, str], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}


def key_for(seed: int, vocab_size: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(rs, log rs, -log(1 - rs)) float32 [V] on ``device`` for one window hash."""
    device = torch.device(device)
    k = (int(seed), int(vocab_size), str(device))
    if k not in _KEYS:
        rs = np.zeros(vocab_size, np.float32)
        lr = np.zeros(vocab_size, np.float32)
        sc = np.zeros(vocab_size, np.float32)
        _lib.check(_lib.load().wmar_gumbel_key_build(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), vocab_size, rs.ctypes.data,
                                                      lr.ctypes.data, sc.ctypes.data))
        _KEYS[k] = tuple(torch.from_numpy(a).to(device) for a in (rs, lr, sc))
        if len(_KEYS) > 4096:
            _KEYS.pop(next(iter(_KEYS)))
    return _KEYS[k]


def _key_rows(window_hash: torch.Tensor, vocab_size: int, device, which: int):
    """Key rows for a batch of hashes: ([V] tensor, stride 0) when all rows share one hash, else ([B, V], V)."""
    hs = [int(h) for h in window_hash.detach().cpu().tolist()]
    if len(set(hs)) == 1:
        return key_for(hs[0], vocab_size, device)[which], 0
    return torch.stack([key_for(h, vocab_size, device)[which] for h in hs]).contiguous(), vocab_size


def gumbel_sample(logits: torch.Tensor, window_hash: torch.Tensor, use_sampling: bool = False, temp: float = 1.0,
                  top_p: float = 0.0, top_k: int = 0) -> torch.Tensor:
    """engine.py:29-75 for logits float32 [B, V] on the GPU -> next tokens int64 [B]."""
    if not logits.is_cuda:
        raise RuntimeError("gumbel_sample: logits must be on the GPU (wmar_amd has no CPU path)")
    lg = logits.detach().to(torch.float32).contiguous()
    B, V = lg.shape
    out = torch.empty(B, dtype=torch.int64, device=lg.device)
    if B == 0:
        return out
    key, stride = _key_rows(window_hash, V, lg.device, 1)
    with torch.cuda.device(lg.device):
        _lib.check(_lib.load().wmar_gumbel_sample(lg.data_ptr(), B, V, key.data_ptr(), stride, int(bool(use_sampling)), float(temp),
                                                  float(top_p), int(top_k), out.data_ptr(), _lib.stream_ptr(lg.device)))
    return out


def gumbel_score_tok(tokens: torch.Tensor, window_hash: torch.Tensor, vocab_size: int) -> torch.Tensor:
    """engine.py:123-134: tokens int64 [B] -> int64 scores (the reference accumulates into ``zeros_like(tokens)``, so
    ``-log(1 - rs)[token]`` arrives truncated)."""
    if not tokens.is_cuda:
        raise RuntimeError("gumbel_score_tok: tokens must be on the GPU (wmar_amd has no CPU path)")
    tk = tokens.detach().to(torch.int64).contiguous().view(-1)
    out = torch.empty_like(tk)
    if tk.numel() == 0:
        return out
    key, stride = _key_rows(window_hash, vocab_size, tk.device, 2)
    with torch.cuda.device(tk.device):
        _lib.check(_lib.load().wmar_gumbel_score(tk.data_ptr(), tk.numel(), 1, vocab_size, key.data_ptr(), stride, out.data_ptr(),
                                                 None, _lib.stream_ptr(tk.device)))
    return out


class GumbelWatermark:
    """Fixed-key Gumbel watermark for an image-token model (extension, see module docstring)."""

    def __init__(self, vocab_size: int, seed: int = 42, temperature: float = 1.0, top_p: float = 0.0, top_k: int = 0,
                 device="cuda"):
        self.vocab_size = int(vocab_size)
        self.seed = int(seed)
        self.temperature, self.top_p, self.top_k = float(temperature), float(top_p), int(top_k)
        self.device = torch.device(device)
        self.rs, self.log_rs, self.score_key = key_for(self.seed, self.vocab_size, self.device)

    def __str__(self):
        return f"gumbel_seed={self.seed}_T={self.temperature}_topp={self.top_p}_topk={self.top_k}"

    def sample(self, logits: torch.Tensor) -> torch.Tensor:
        h = torch.full((logits.shape[0],), self.seed, dtype=torch.int64)
        return gumbel_sample(logits, h, True, self.temperature, self.top_p, self.top_k)

    def scores(self, codes: torch.Tensor) -> torch.Tensor:
        """float32 [B, L] per-token scores -log(1 - rs[code])."""
        codes = codes.to(self.device, torch.int64).contiguous()
        B, L = codes.shape
        out = torch.empty(B, L, dtype=torch.float32, device=self.device)
        if B:
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().wmar_gumbel_score(codes.data_ptr(), B, L, self.vocab_size, self.score_key.data_ptr(), 0,
                                                         None, out.data_ptr(), _lib.stream_ptr(self.device)))
        return out

    def detect(self, codes: torch.Tensor) -> torch.Tensor:
        """p-values float64 [B]: under H0 the scores are i.i.d. Exp(1), so their sum is Gamma(L, 1)."""
        s = self.scores(codes).to(torch.float64)
        L = torch.full((s.shape[0],), float(s.shape[1]), dtype=torch.float64, device=s.device)
        return torch.special.gammaincc(L, s.sum(dim=1))
