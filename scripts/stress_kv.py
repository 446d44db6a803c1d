"""Dev tool (WMAR_DEV_KNOBS build): one decode step repeated at a fixed position on a fixed cache; every repetition must rewrite the
same K / V rows in every layer and return the same logits.  On a mismatch: the first layer whose K / V row differs, and where."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
from wmar_amd import _lib
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
POS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state(cfg, seed=0, logit_scale=10.0)
seq = torch.randint(0, cfg.vocab_size, (B, 256), generator=torch.Generator().manual_seed(5)).cuda()
L = _lib.load()
eng = GPTEngine(cfg, sd, max_batch=64)
H, hd = cfg.n_head, cfg.n_embd // cfg.n_head
W = int(sys.argv[4]) if len(sys.argv) > 4 else 8       # cycle over positions POS-W+1 .. POS: every step then runs on buffers left by ANOTHER position
def kv(POS):
    out = []
    for which in (0, 1):
        a = np.empty((cfg.n_layer, 64, H, hd), dtype=np.float32)
        _lib.check(L.wmar_gpt_debug_kv_row(eng._h, which, int(POS), a.ctypes.data_as(C.c_void_p)))
        out.append(a[:, :B])
    return out
HAVE_ST = hasattr(L, "wmar_gpt_debug_stats")
def stats():
    if not HAVE_ST:
        return None
    a = np.empty((cfg.n_layer, 64, H, 4), dtype=np.float64)
    _lib.check(L.wmar_gpt_debug_stats(eng._h, a.ctypes.data_as(C.c_void_p)))
    return a[:, :B]
ref_l, ref_kv, ref_st = {}, {}, {}
for t in range(POS + 1):
    lg = eng.decode_step(seq[:, t], t)
    if t > POS - W:
        ref_l[t] = lg.clone(); ref_kv[t] = kv(t); ref_st[t] = stats()
bad = 0
for r in range(REPS):
    t = POS - W + 1 + r % W
    lg = eng.decode_step(seq[:, t], t)
    if torch.equal(lg, ref_l[t]):
        continue
    bad += 1
    k, v = kv(t)
    ref_k, ref_v = ref_kv[t]
    rows = (lg != ref_l[t]).any(1).nonzero().view(-1).tolist()
    msg = f"rep {r} pos {t}: logits rows {rows[0]}..{rows[-1]} ({len(rows)}) differ, max |d| {float((lg - ref_l[t]).abs().max()):.2e};"
    for l in range(cfg.n_layer):
        dk, dv = k[l] != ref_k[l], v[l] != ref_v[l]
        if dk.any() or dv.any():
            d = dk | dv
            bs, hs, cs = np.nonzero(d.any((1, 2)))[0], np.nonzero(d.any((0, 2)))[0], np.nonzero(d.any((0, 1)))[0]
            msg += (f" first layer {l} ({'k' if dk.any() else ''}{'v' if dv.any() else ''}): rows {bs.min()}..{bs.max()} ({len(bs)}), heads {hs.tolist()[:24]},"
                    f" cols {cs.min()}..{cs.max()} ({len(cs)}), n {int(dk.sum())}/{int(dv.sum())}, max |dk| {float(np.abs(k[l] - ref_k[l]).max()):.2e} |dv| {float(np.abs(v[l] - ref_v[l]).max()):.2e}")
            if HAVE_ST:
                st = stats(); ds = st != ref_st[t]
                ls = np.nonzero(ds.any((1, 2, 3)))[0]
                if len(ls):
                    l0 = ls[0]; bb = np.nonzero(ds[l0].any((1, 2)))[0]; cc = np.nonzero(ds[l0].any((0, 1)))[0]
                    b0 = bb[0]
                    msg += f"; stats first differ at layer {l0}: rows {bb.min()}..{bb.max()} ({len(bb)}), fields {cc.tolist()}, row {b0} head 0: got {st[l0, b0, 0].tolist()} want {ref_st[t][l0, b0, 0].tolist()}"
                else:
                    msg += "; stats identical in every layer"
            break
    else:
        msg += " no K/V row differs (after the last layer's QKV)"
    print(msg, flush=True)
print("reps", REPS, "mismatching", bad)
