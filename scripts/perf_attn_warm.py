"""Dev tool: decode attention with the KV cache streamed from HBM (cycling through the layers) vs from the 256 MB Infinity Cache
(WMAR_PROFILE_LAYERS=1, dev build: the same layer every launch)."""
import sys
sys.path.insert(0, ".")
import torch
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
eng = GPTEngine(cfg, sd, max_batch=64); del sd
eng.set_attention_phases(256, 256)
print(" ".join(f"kv{kv}={eng.profile_role('attn', 64, kv_len=kv, iters=96):.1f}" for kv in (1, 32, 64, 128, 192, 256)))
