import sys, torch
sys.path.insert(0, ".")
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
B = 64
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
eng = GPTEngine(cfg, sd, max_batch=B); del sd
for kv in (16, 128, 256):
    us = eng.profile_role("attn", B, kv_len=kv, iters=96)
    print(f"attn kv={kv:3d}: {us:7.2f} us  {2.0*B*cfg.n_embd*4*kv/1e9/us*1e3:6.2f} TB/s")
