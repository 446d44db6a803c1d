"""Dev tool for PMC passes: replays one decode role (default fc1) a few times."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
role = sys.argv[1] if len(sys.argv) > 1 else "fc1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
eng = GPTEngine(cfg, sd, max_batch=B); del sd
torch.cuda.synchronize()
print(role, eng.profile_role(role, B, kv_len=128, iters=48))
