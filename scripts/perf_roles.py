"""Dev tool: per-role replay timings of the decode step (HIP events)."""
import sys
import torch
sys.path.insert(0, ".")
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
B = 64
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
eng = GPTEngine(cfg, sd, max_batch=B); del sd
for kv in (1, 16, 64, 128, 192, 256):
    us = eng.profile_role("attn", B, kv_len=kv, iters=96)
    gb = 2.0 * B * cfg.n_embd * 4 * kv / 1e9
    print(f"attn kv={kv:3d}: {us:7.2f} us  {gb/us*1e3:6.2f} TB/s")
tot = 0
for r, n in (("qkv", 48), ("proj", 48), ("resid", 97), ("fc1", 48), ("fc2", 48), ("head", 1), ("embed", 1)):
    us = eng.profile_role(r, B, kv_len=128, iters=96)
    tot += us * n
    print(f"{r:6s} {us:7.2f} us x{n}")
print("sum w/o attn, sampler: %.3f ms" % (tot / 1e3))
