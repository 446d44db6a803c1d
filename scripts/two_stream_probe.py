"""Probe (round 5): do two independent batch-64 generation loops (two engines, two HIP streams, two host threads) overlap on one MI355X?
Each decode kernel is a short full-chip launch bound by fixed costs (ramp, first-load latency, tail): a second stream could fill those
bubbles.  Prints wall time of two generations run one after the other and run concurrently, with the fused projection launch on and off
(its XCD barrier needs all 192 workgroups resident: a co-running kernel can starve it -> the engine falls back by itself)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wmar_amd.models.engine import GPTEngine  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402

cfg = synth.TAMING_GPT
sd = synth.synth_gpt_state_fast(cfg, seed=0, device="cuda", logit_scale=30.0)
engs = [GPTEngine(cfg, sd, max_batch=64) for _ in range(2)]
del sd
B, steps = 64, 256
streams = [torch.cuda.Stream() for _ in range(2)]
qs, conds = [], []
for i in range(2):
    g = torch.Generator(device="cuda").manual_seed(i)
    qs.append(torch.empty(steps, B, cfg.vocab_size, device="cuda").exponential_(1, generator=g))
    conds.append((torch.arange(B, device="cuda") * 37 + i) % 1000)
torch.cuda.synchronize()


def gen(i, out):
    with torch.cuda.stream(streams[i]):
        out[i] = engs[i].generate(conds[i], steps, qs[i], 1.0, 250, 0.92, None, use_graph=True)
        streams[i].synchronize()


ref = [None, None]
for i in range(2):
    gen(i, ref)            # warm-up: captures the graphs
print("plan:", engs[0].plan_info(64)["proj"])
for rep in range(3):
    out = [None, None]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gen(0, out); gen(1, out)
    torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    out = [None, None]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=gen, args=(i, out)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); t_con = time.perf_counter() - t0
    same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    print(f"rep {rep}: sequential {t_seq * 1e3:.1f} ms ({t_seq / 2 / steps * 1e3:.3f} ms/step), concurrent {t_con * 1e3:.1f} ms "
          f"(x{t_seq / t_con:.3f}), tokens equal: {same}, plan now: {engs[0].plan_info(64)['proj'][:12]} fallbacks {engs[0].plan_info(64)['barrier_fallbacks']}/{engs[1].plan_info(64)['barrier_fallbacks']}")
