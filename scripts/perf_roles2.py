"""Dev tool: per-role replay timings (HIP events), GEMM roles + resid only."""
import sys
import torch
sys.path.insert(0, ".")
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
B = 64
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
eng = GPTEngine(cfg, sd, max_batch=B); del sd
print(" ".join(f"{r}={eng.profile_role(r, B, kv_len=128, iters=96):.2f}" for r in ("qkv", "proj", "resid", "fc1", "fc2", "head")))
