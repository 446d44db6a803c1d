import sys, os
sys.path.insert(0, ".")
import torch
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
eng = GPTEngine(cfg, sd, max_batch=64); del sd
for n in (48, 8, 2, 1):
    os.environ["WMAR_PROFILE_LAYERS"] = str(n)
    print(n, "layers:", " ".join(f"{r}={eng.profile_role(r, 64, kv_len=128, iters=96):.2f}" for r in ("qkv", "proj", "fc1", "fc2", "attn")), flush=True)
