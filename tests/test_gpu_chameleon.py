"""HIP Chameleon decode engine (bf16, RMSNorm fold, qk-norm, RoPE, SwiGLU, per-row positions) against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cham_oracle as CO  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402


def _cfg(hd=64, kv=None, qk=True, dim=256, layers=2, vocab=1024):
    H = dim // hd
    return synth.ChameleonConfig(dim=dim, n_layers=layers, n_heads=H, n_kv_heads=kv or H, vocab_size=vocab, multiple_of=64,
                                 qk_normalization=qk)


def _close(got, ref, frac):
    """bf16 pipelines agree to a fraction of the logit scale (+ one bf16 ulp of the largest logit: outputs ARE bf16 values)."""
    scale = float(ref.std())
    err = float((got - ref).abs().max())
    assert err <= frac * scale + 2.0 ** -7 * float(ref.abs().max()), (err, scale)
    assert float((got - ref).abs().mean()) <= 0.25 * frac * scale


@pytest.mark.parametrize("hd,kv,qk,M", [(64, None, True, 5), (128, None, True, 48), (64, 2, False, 33), (128, 1, True, 96)])
def test_forward_tokens_vs_oracle(hd, kv, qk, M):
    from wmar_amd.models.engine import ChameleonEngine
    cfg = _cfg(hd=hd, kv=kv, qk=qk, dim=256 if hd == 64 else 512)
    sd = synth.synth_chameleon_state(cfg, seed=3, logit_scale=4.0)
    e = ChameleonEngine(cfg, sd, max_batch=(M + 2) // 3, max_seq_len=32)
    rs = np.random.RandomState(M)
    cache_f, cache_r = CO.Cache(cfg.n_layers, M), CO.Cache(cfg.n_layers, M)
    start = torch.from_numpy(rs.randint(0, 3, size=M).astype(np.int32)) * 0     # every row starts at position 0
    for step in range(6):
        tok = torch.from_numpy(rs.randint(0, cfg.vocab_size, size=M).astype(np.int64))
        pos = start + step
        got = e.forward_tokens(tok.cuda(), pos.cuda()).cpu()
        ref_f = CO.forward_tokens(sd, cfg, tok, pos, cache_f, fold=True)
        ref_r = CO.forward_tokens(sd, cfg, tok, pos, cache_r, fold=False)
        _close(got, ref_f, 0.03)     # same algebra as the engine: differences are summation order + rare bf16 flips
        _close(got, ref_r, 0.08)     # the reference's rounding points


def test_ragged_positions_and_cache_rewrite():
    """rows at different positions in one step; a row restarted at position 0 ignores its stale cache."""
    from wmar_amd.models.engine import ChameleonEngine
    cfg = _cfg(hd=64, dim=256)
    sd = synth.synth_chameleon_state(cfg, seed=5, logit_scale=4.0)
    e = ChameleonEngine(cfg, sd, max_batch=2, max_seq_len=16)
    prompts = [[5, 6, 7, 8, 9], [11, 12], [3], [1, 2, 3, 4], [9, 9, 9], [7, 700]]
    M = len(prompts)
    maxlen = max(len(p) for p in prompts)
    cache = CO.Cache(cfg.n_layers, M)
    ref, nxt = CO.prefill_right_aligned(sd, cfg, prompts, cache, fold=True)
    got = None
    for j in range(maxlen):
        tok, pos = [], []
        for p in prompts:
            i = j - (maxlen - len(p))
            tok.append(p[i] if i >= 0 else 0)
            pos.append(max(i, 0))
        got = e.forward_tokens(torch.tensor(tok).cuda(), torch.tensor(pos, dtype=torch.int32).cuda())
    _close(got.cpu(), ref, 0.03)
    tok = torch.tensor([1, 2, 3, 4, 5, 6])
    got2 = e.forward_tokens(tok.cuda(), nxt.to(torch.int32).cuda()).cpu()
    ref2 = CO.forward_tokens(sd, cfg, tok, nxt, cache, fold=True)
    _close(got2, ref2, 0.03)
