"""Dev tool: time the full-size Taming GPT decode loop (random weights)."""
import os, sys, time
import torch
ROOT = os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
graph = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = synth.TAMING_GPT
t0 = time.time()
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
torch.cuda.synchronize(); print("weights", time.time() - t0)
eng = GPTEngine(cfg, sd, max_batch=max(B, 1)); del sd; torch.cuda.empty_cache()
print("engine bytes", eng.device_bytes / 1e9)
ids = []
for line in open(os.path.join(ROOT, "wmar_amd/assets/vqgan_alive_ids.txt")): ids.extend(int(t) for t in line.split(","))
dead = sorted(set(range(16384)) - set(ids))
wm = GentimeWatermark({"alive_ids": torch.tensor(ids), "dead_ids": torch.tensor(dead), "embedding": None}, 16384,
                      SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
ctx = wm.wm_ctx()
q = torch.empty(steps, B, 16384, device="cuda").exponential_(1)
cond = (torch.arange(B) * 37 % 1000).cuda()
eng.set_timing(True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    toks = eng.generate(cond, steps, q, 1.0, 250, 0.92, ctx, use_graph=bool(graph))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"B={B} steps={steps} graph={graph}: {dt*1e3:.1f} ms total, {dt/steps*1e3:.3f} ms/step (device {eng.get_timing()[1]:.3f})")
if not graph:
    cls, _ = eng.get_timing()
    for k, (us, n) in cls.items():
        if n: print(f"  {k:7s} {us/n:8.2f} us avg x {n}")
pv = wm.detect(toks)
print("pvals", pv[:4].tolist())
