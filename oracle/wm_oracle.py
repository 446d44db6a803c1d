"""TEST INFRASTRUCTURE ONLY -- numpy/ctypes front end of oracle/wm_oracle.c.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.  The product package (``wmar_amd``) never does.

It restates, on the CPU, the integer part of the reference's watermark path:
``GentimeWatermark._split_with_seed`` / ``_process_logits`` / ``detect``
(wmar/watermarking/gentime_watermark.py:161-174, 229-271, 285-344) and the
sampling stage of ``sample_with_past`` (deps/taming/modules/transformer/mingpt.py:348-363).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SPLIT = {"rand": 0, "stratifiedrand": 1}
SEED = {"fixed": 0, "linear": 1, "spatial": 2}


def build() -> str:
    """Compile oracle/libwm_oracle.so (gcc) if missing or stale."""
    so = os.path.join(_HERE, "libwm_oracle.so")
    srcs = [os.path.join(_HERE, "wm_oracle.c"), os.path.join(_HERE, "..", "include", "wmar_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        i64p = np.ctypeslib.ndpointer(np.int64, flags="C")
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
        L.wmo_randperm.argtypes = [C.c_uint64, C.c_int64, i64p, C.c_int64]
        L.wmo_context_seed.restype = C.c_uint64
        L.wmo_context_seed.argtypes = [C.c_uint64, C.c_int64]
        L.wmo_greenlist.restype = C.c_int64
        L.wmo_greenlist.argtypes = [C.c_uint64, i64p, C.c_int64, i64p, C.c_int64, C.c_int64, C.c_double, C.c_int, i64p]
        L.wmo_process_logits.argtypes = [C.c_uint64, i64p, C.c_int64, i64p, C.c_int64, C.c_int64, C.c_double, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_double, i64p, C.c_int64, C.c_int64, f32p]
        L.wmo_sample_row.restype = C.c_int64
        L.wmo_sample_row.argtypes = [f32p, C.c_int64, C.c_double, C.c_int, C.c_double, f32p, C.c_void_p, C.c_void_p]
        L.wmo_gumbel_key.restype = None
        L.wmo_gumbel_key.argtypes = [C.c_uint64, C.c_int64, f32p]
        L.wmo_gumbel_sample_row.restype = C.c_int64
        L.wmo_gumbel_sample_row.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_float, C.c_float, C.c_int]
        L.wmo_gumbel_score.restype = C.c_float
        L.wmo_gumbel_score.argtypes = [f32p, C.c_int64]
        L.wmo_betainc_int.restype = C.c_double
        L.wmo_betainc_int.argtypes = [C.c_int64, C.c_int64, C.c_double]
        L.wmo_detect_one.argtypes = [C.c_uint64, i64p, C.c_int64, i64p, C.c_int64, C.c_int64, C.c_double, C.c_int,
                                     C.c_int, C.c_int, i64p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int64)]
        L.wmo_key_table.argtypes = [C.c_uint64, i64p, C.c_int64, i64p, C.c_int64, C.c_int64, C.c_double, C.c_int,
                                    C.c_int, C.c_int64, C.c_int64, np.ctypeslib.ndpointer(np.uint32, flags="C")]
        _LIB = L
    return _LIB


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


class KeyParams:
    """The watermark key: what GentimeWatermark.__init__ holds (gentime_watermark.py:110-154)."""

    def __init__(self, alive_ids, dead_ids, vocab_size, gamma, split="stratifiedrand", seed="linear",
                 context_size=1, salt_key=15485863, spatial_dim=16):
        self.alive = _i64(alive_ids)
        self.dead = _i64(dead_ids)
        self.vocab = int(vocab_size)
        self.gamma = float(gamma)
        self.split = SPLIT[split]
        self.seed = SEED[seed]
        self.h = int(context_size)
        self.salt = int(salt_key)
        self.spatial_dim = int(spatial_dim)

    def _common(self):
        return (self.salt, self.alive, len(self.alive), self.dead, len(self.dead), self.vocab, self.gamma, self.split)


def randperm(seed: int, n: int, n_out: int | None = None) -> np.ndarray:
    n_out = n if n_out is None else n_out
    out = np.zeros(max(n_out, 1), dtype=np.int64)
    lib().wmo_randperm(seed & (2**64 - 1), n, out, n_out)
    return out[:n_out]


def context_seed(salt: int, ctx_sum: int) -> int:
    return int(lib().wmo_context_seed(salt, ctx_sum))


def greenlist(key: KeyParams, ctx_sum: int) -> np.ndarray:
    """_get_greenlist_ids_for_context: ids in the reference's (unsorted) order."""
    seed = 0 if key.seed == 0 else context_seed(key.salt, ctx_sum)
    out = np.zeros(key.vocab + 8, dtype=np.int64)
    n = lib().wmo_greenlist(seed, key.alive, len(key.alive), key.dead, len(key.dead), key.vocab, key.gamma,
                            key.split, out)
    return out[:n].copy()


def greenlist_for_seed(key: KeyParams, seed: int) -> np.ndarray:
    out = np.zeros(key.vocab + 8, dtype=np.int64)
    n = lib().wmo_greenlist(seed & (2**64 - 1), key.alive, len(key.alive), key.dead, len(key.dead), key.vocab,
                            key.gamma, key.split, out)
    return out[:n].copy()


def process_logits(key: KeyParams, past_ids, logits, delta: float) -> np.ndarray:
    """_process_logits (in place on a float32 copy, returned)."""
    past = _i64(past_ids)
    lg = np.ascontiguousarray(np.asarray(logits, dtype=np.float32)).copy()
    assert past.ndim == 2 and lg.ndim == 2 and lg.shape[1] == key.vocab
    rc = lib().wmo_process_logits(*key._common(), key.seed, key.h, key.spatial_dim, float(delta), past,
                                  past.shape[0], past.shape[1], lg)
    if rc != 0:
        raise AssertionError("Spatial seeding only implemented for context size in [1,3]")
    return lg


def sample_rows(logits, q, temperature=1.0, top_k=None, top_p=None, return_all=False):
    """Rows of mingpt.py:351-363 after the logit processor. Returns int64 tokens [B]."""
    lg = np.ascontiguousarray(np.asarray(logits, dtype=np.float32))
    qq = np.ascontiguousarray(np.asarray(q, dtype=np.float32))
    B, V = lg.shape
    toks = np.zeros(B, dtype=np.int64)
    xs = np.zeros((B, V), dtype=np.float32) if return_all else None
    ps = np.zeros((B, V), dtype=np.float32) if return_all else None
    for b in range(B):
        toks[b] = lib().wmo_sample_row(lg[b], V, float(temperature), int(top_k) if top_k else 0,
                                       float(top_p) if top_p is not None else -1.0, qq[b],
                                       xs[b].ctypes.data if return_all else None,
                                       ps[b].ctypes.data if return_all else None)
    return (toks, xs, ps) if return_all else toks


def betainc_int(a: int, b: int, x: float) -> float:
    return float(lib().wmo_betainc_int(a, b, x))


def detect(key: KeyParams, codes, return_masks=False):
    """GentimeWatermark.detect: returns (pvals f64[B], n_scored i32[B], n_green i32[B][, masks])."""
    cd = _i64(codes)
    assert cd.ndim == 2
    B, L = cd.shape
    pv = np.zeros(B, dtype=np.float64)
    ns = np.zeros(B, dtype=np.int32)
    ng = np.zeros(B, dtype=np.int32)
    masks = []
    for b in range(B):
        a, g, p, ml = C.c_int32(), C.c_int32(), C.c_double(), C.c_int64()
        m = np.zeros(L + key.h + 1, dtype=np.int8)
        rc = lib().wmo_detect_one(*key._common(), key.seed, key.h, cd[b], L, C.byref(a), C.byref(g), C.byref(p),
                                  m.ctypes.data if return_masks else None, C.byref(ml))
        if rc == -1:
            raise ValueError("Must have at least 1 token to score after the first min_context_len tokens")
        if rc != 0:
            raise ValueError(f"spatial n-grams unsupported for this input (rc={rc})")
        pv[b], ns[b], ng[b] = p.value, a.value, g.value
        masks.append(m[: ml.value].tolist())
    return (pv, ns, ng, masks) if return_masks else (pv, ns, ng)


def key_table(key: KeyParams, row0: int, n_rows: int) -> np.ndarray:
    words = (key.vocab + 31) // 32
    out = np.zeros((n_rows, words), dtype=np.uint32)
    lib().wmo_key_table(*key._common(), key.seed, row0, n_rows, out)
    return out


# ------------------------------------------------------------------ Gumbel key (row G1)
def gumbel_key(seed: int, vocab: int) -> np.ndarray:
    """rs = torch.rand(vocab, generator=MT19937(seed)) (wmar_audio/watermark/engine.py:64-66)."""
    out = np.zeros(vocab, dtype=np.float32)
    lib().wmo_gumbel_key(C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), vocab, out)
    return out


def gumbel_sample(logits, window_hash, use_sampling=False, temp=1.0, top_p=0.0, top_k=0) -> np.ndarray:
    """gumbel_sample of engine.py:29-75: logits float32 [B, V], window_hash int [B] -> tokens int64 [B]."""
    logits = np.ascontiguousarray(np.asarray(logits, dtype=np.float32))
    B, V = logits.shape
    out = np.zeros(B, dtype=np.int64)
    keys = {}
    for b in range(B):
        h = int(window_hash[b])
        if h not in keys:
            keys[h] = gumbel_key(h, V)
        out[b] = lib().wmo_gumbel_sample_row(np.ascontiguousarray(logits[b]), V, keys[h], int(bool(use_sampling)),
                                             float(temp), float(top_p), int(top_k))
    return out


def gumbel_score_tok(tokens, window_hash, vocab: int, truncate=True) -> np.ndarray:
    """gumbel_score_tok of engine.py:123-134: tokens int64 [B] -> scores (int64-truncated like the reference, or fp32)."""
    tokens = np.asarray(tokens, dtype=np.int64)
    out = np.zeros(tokens.shape[0], dtype=np.float32)
    for b in range(tokens.shape[0]):
        rs = gumbel_key(int(window_hash[b]), vocab)
        out[b] = lib().wmo_gumbel_score(rs, int(tokens[b]))
    return out.astype(np.int64) if truncate else out
