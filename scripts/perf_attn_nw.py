"""Dev tool: Taming decode attention vs cached length for the WMAR_ATT_NW variants (run once per setting)."""
import sys, os
sys.path.insert(0, ".")
import torch
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
eng = GPTEngine(cfg, sd, max_batch=64); del sd
ts = []
for kv in (1, 32, 64, 96, 128, 160, 192, 224, 256):
    ts.append(eng.profile_role("attn", 64, kv_len=kv, iters=96))
print("NW", os.environ.get("WMAR_ATT_NW", "2"), " ".join(f"{t:.1f}" for t in ts), " mean %.2f us" % (sum(ts) / len(ts)))
