"""Dev tool (WMAR_DEV_KNOBS build, WMAR_DBG_SUMS=1): teacher-forced decode repeated; after every step the per-launch checksums
(gpt.hip k_dbg_sum) are compared with the first pass's -- the first slot that differs names the launch that was not reproducible."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WMAR_DBG_SUMS"] = "1"
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
from wmar_amd import _lib
ENGINES = int(sys.argv[1]) if len(sys.argv) > 1 else 2
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state(cfg, seed=0, logit_scale=10.0)
seq = torch.randint(0, cfg.vocab_size, (B, 256), generator=torch.Generator().manual_seed(5)).cuda()
L = _lib.load()
L.wmar_gpt_debug_sums.restype = C.c_int
NAMES = ["x' (fold)", "stats_q", "QKV pieces", "attention out", "k cache", "v cache", "proj slabs", "x (resid)", "stats", "hidden", "FC2 slabs"]
def slot_name(i):
    if i < 2: return ["embed x", "embed stats"][i]
    l, r = divmod(i - 2, 11)
    return f"layer {l}: {NAMES[r]}" if l < cfg.n_layer else f"final resid {r}"
ref = None
bad = 0
for e in range(ENGINES):
    eng = GPTEngine(cfg, sd, max_batch=64)
    cref = []
    for p in range(PASSES):
        cur = []
        ok = True
        for t in range(256):
            eng.decode_step(seq[:, t], t)
            buf = (C.c_ulonglong * 512)(); used = C.c_int(0)
            _lib.check(L.wmar_gpt_debug_sums(eng._h, buf, 512, C.byref(used)))
            s = np.frombuffer(buf, dtype=np.uint64)[:used.value].copy()
            idx = np.arange(len(s))
            is_cache = (idx >= 2) & (idx < 2 + 11 * cfg.n_layer) & np.isin((idx - 2) % 11, (4, 5))
            if p == 1:
                cref.append(s.copy())        # caches are full from this engine's pass 0 on: pass 1 is their reference
            if ref is None:
                cur.append(s)
                continue
            want = ref[t].copy()
            if p >= 2:
                want[is_cache] = cref[t][is_cache]
            else:
                s = s.copy(); s[is_cache] = want[is_cache]
            if ok and not np.array_equal(s, want):
                d = np.nonzero(s != want)[0]
                print(f"engine {e} pass {p} position {t}: first differing slot {d[0]} = {slot_name(int(d[0]))}; {len(d)} slots differ: {[slot_name(int(i)) for i in d[:6]]}", flush=True)
                bad += 1; ok = False
        if ref is None:
            ref = cur
    del eng
print("engines", ENGINES, "passes", PASSES, "mismatching passes", bad)
