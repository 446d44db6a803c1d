"""Randomised parity sweep (dev tool, run on the GPU box): fused sampler, Gumbel sampler and detector against the oracle on
many seeded rows with adversarial structure (ties, -inf, dominant tokens, bf16-valued logits, compaction)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from oracle import wm_oracle as W
from wmar_amd import _lib
from wmar_amd.watermarking.gumbel_watermark import gumbel_sample

L = _lib.load()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0

def rows(rs, B, V, kind):
    x = (rs.randn(B, V) * rs.choice([0.5, 2.0, 6.0])).astype(np.float32)
    if kind == 1: x = np.round(x * 2) / 2
    if kind == 2: x = torch.from_numpy(x).bfloat16().float().numpy()
    if kind == 3: x[:, rs.randint(0, V, size=V // 3)] = -np.inf
    if kind == 4: x[np.arange(B), rs.randint(0, V, size=B)] += 40.0
    if kind == 5: x[:] = x[:, :1]
    return x

for seed in range(n_seeds):
    rs = np.random.RandomState(1000 + seed)
    V = int(rs.choice([1024, 4096, 16384, 65536]))
    B = int(rs.choice([1, 7, 16]))
    kind = seed % 6
    lg = rows(rs, B, V, kind)
    q = rs.exponential(size=(B, V)).astype(np.float32)
    T = float(rs.choice([1.0, 0.7, 1.3]))
    tk = rs.choice([None, 1, 50, 250, V])
    tp = rs.choice([None, 0.0, 0.5, 0.92, 1.0])
    tk = None if tk is None else int(tk)
    tp = None if tp is None else float(tp)
    exp = W.sample_rows(lg, q, T, tk, tp)
    d_lg, d_q = torch.from_numpy(lg).cuda(), torch.from_numpy(q).cuda()
    out = torch.empty(B, dtype=torch.int64, device="cuda"); scratch = torch.empty(B, V, device="cuda")
    _lib.check(L.wmar_sample_fused(None, d_lg.data_ptr(), B, V, None, 0, 0, T, tk or 0, tp if tp is not None else -1.0, d_q.data_ptr(),
                                   scratch.data_ptr(), out.data_ptr(), _lib.stream_ptr()))
    ok = out.cpu().numpy().tolist() == exp.tolist()
    # compaction on a random allow list (exact when the rest is masked)
    ids = np.sort(rs.choice(V, size=max(32, V // 8), replace=False)).astype(np.int32)
    masked = lg.copy(); keep = np.zeros(V, bool); keep[ids] = True; masked[:, ~keep] = -np.inf
    exp2 = W.sample_rows(masked, q, T, None, tp)
    bits = np.zeros(V // 32, np.uint32); np.bitwise_or.at(bits, ids >> 5, np.uint32(1) << (ids & 31).astype(np.uint32))
    d_bits = torch.from_numpy(bits.view(np.int32)).cuda(); d_ids = torch.from_numpy(ids).cuda()
    lg3 = torch.cat([d_lg, d_lg, d_lg])           # identical streams: the guidance mix is the identity
    _lib.check(L.wmar_cham_sample(None, lg3.data_ptr(), B, V, None, 0, 0, T, tp if tp is not None else -1.0, 3.0, 1.2, d_bits.data_ptr(),
                                  d_ids.data_ptr(), len(ids), d_q.data_ptr(), scratch.data_ptr(), out.data_ptr(), _lib.stream_ptr()))
    ok2 = out.cpu().numpy().tolist() == exp2.tolist()
    # Gumbel
    okg = True
    if V <= 16384:
        h = torch.from_numpy(rs.randint(0, 2 ** 31, size=B).astype(np.int64))
        gp, gk = float(rs.choice([0.0, 0.3, 0.9])), int(rs.choice([0, 5, 100]))
        got = gumbel_sample(d_lg, h, True, T, gp, gk).cpu().numpy()
        okg = got.tolist() == W.gumbel_sample(lg, h.numpy(), True, T, gp, gk).tolist()
    print(f"seed {seed:3d} V={V:6d} B={B:2d} kind={kind} T={T} top_k={tk} top_p={tp}: sampler {'ok' if ok else 'MISMATCH'}  "
          f"compact {'ok' if ok2 else 'MISMATCH'}  gumbel {'ok' if okg else 'MISMATCH'}", flush=True)
    bad += (not ok) + (not ok2) + (not okg)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
