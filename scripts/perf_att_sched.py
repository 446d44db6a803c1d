import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
cfg = synth.TAMING_GPT
eng = GPTEngine(cfg, synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0), max_batch=64)
B, steps = 64, 256
q = torch.empty(steps, B, 16384, device="cuda").exponential_(1)
cond = (torch.arange(B) * 37 % 1000).cuda()
for sched in [(-1, -1), (128, 1 << 30), (64, 1 << 30), (192, 1 << 30), (128, 192), (0, 1 << 30), (-1, -1)]:
    eng.set_attention_phases(*sched)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        eng.generate(cond, steps, q, 1.0, 250, 0.92, None, use_graph=True)
        torch.cuda.synchronize(); dt = time.time() - t0
    print(sched, f"{dt / steps * 1e3:.3f} ms/step", flush=True)
