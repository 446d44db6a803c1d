"""Dev tool: Taming decode attention vs cached length for 1 / 2 / 4 waves per (sequence, head)."""
import sys
sys.path.insert(0, ".")
import torch
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = GPTEngine(cfg, sd, max_batch=B); del sd
kvs = (1, 16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 256)
print("kv  " + " ".join(f"{k:5d}" for k in kvs))
rows = {}
for nw, (t1, t2) in ((1, (256, 256)), (2, (0, 256)), (4, (0, 0))):
    eng.set_attention_phases(t1, t2)
    rows[nw] = [eng.profile_role("attn", B, kv_len=kv, iters=96) for kv in kvs]
    print(f"NW{nw} " + " ".join(f"{t:5.1f}" for t in rows[nw]))
best = [min((rows[nw][i], nw) for nw in rows) for i in range(len(kvs))]
print("best " + " ".join(f"{nw:5d}" for _, nw in best), " mean of best %.2f us" % (sum(t for t, _ in best) / len(best)))
