"""Dev experiment: two independent batch-64 decode pipelines on two HIP streams (two engines)."""
import sys, time, threading
import torch
sys.path.insert(0, ".")
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy

B, steps = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 128
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0)
engs = [GPTEngine(cfg, sd, max_batch=B) for _ in range(NS)]
del sd; torch.cuda.empty_cache()
ids = []
for line in open("wmar_amd/assets/vqgan_alive_ids.txt"): ids.extend(int(t) for t in line.split(","))
dead = sorted(set(range(16384)) - set(ids))
wm = GentimeWatermark({"alive_ids": torch.tensor(ids), "dead_ids": torch.tensor(dead), "embedding": None}, 16384,
                      SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
ctx = wm.wm_ctx()
qs = [torch.empty(steps, B, 16384, device="cuda").exponential_(1) for _ in range(NS)]
cond = (torch.arange(B) * 37 % 1000).cuda()
streams = [torch.cuda.Stream() for _ in range(NS)]
torch.cuda.synchronize()

def run(i, out):
    with torch.cuda.stream(streams[i]):
        out[i] = engs[i].generate(cond, steps, qs[i], 1.0, 250, 0.92, ctx, use_graph=True)

for it in range(3):
    out = [None] * NS
    torch.cuda.synchronize(); t0 = time.time()
    th = [threading.Thread(target=run, args=(i, out)) for i in range(NS)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{NS} streams x B={B} steps={steps}: {dt*1e3:.1f} ms -> {dt/steps*1e3/NS:.3f} ms per batch-step")
# single for reference
torch.cuda.synchronize(); t0 = time.time()
engs[0].generate(cond, steps, qs[0], 1.0, 250, 0.92, ctx, use_graph=True)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"1 stream: {dt/steps*1e3:.3f} ms per batch-step")
