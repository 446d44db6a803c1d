"""Dev tool: teacher-forced decode of the full-size Taming GPT, repeated; every pass must reproduce the first pass's logits
bit for bit (fresh engines included: their first pass runs on cold caches and untouched buffers)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
ENGINES = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = synth.TAMING_GPT
# WMAR_FAST_WEIGHTS=1: weights drawn on the GPU (seconds instead of a minute; any weights do for this test)
sd = synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=10.0) if os.environ.get("WMAR_FAST_WEIGHTS") else synth.synth_gpt_state(cfg, seed=0, logit_scale=10.0)
g = torch.Generator().manual_seed(5)
seq = torch.randint(0, cfg.vocab_size, (B, 256), generator=g).cuda()
ref = None
bad = 0
for e in range(ENGINES):
    eng = GPTEngine(cfg, sd, max_batch=64)
    for p in range(PASSES):
        cur = torch.empty(256, B, cfg.vocab_size, device="cuda") if ref is None else None
        for t in range(256):
            lg = eng.decode_step(seq[:, t], t)
            if cur is not None:
                cur[t].copy_(lg)
            elif not torch.equal(lg, ref[t]):
                d = (lg != ref[t])
                rows = d.any(1).nonzero().view(-1).tolist()
                print(f"engine {e} pass {p} position {t}: rows {rows[:8]} differ, {int(d.sum())} logits, max |d| {float((lg - ref[t]).abs().max()):.3e}", flush=True)
                bad += 1
                break
        if cur is not None:
            ref = cur
    del eng
print("engines", ENGINES, "passes", PASSES, "mismatching passes", bad)
