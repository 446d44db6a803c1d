"""Chameleon decode step time as a function of the cached length (dev tool): fills the cache by running forward_tokens
at increasing positions and times steps around selected lengths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wmar_amd.models.engine import ChameleonEngine
from wmar_amd.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = synth.CHAMELEON_7B
sd = synth.synth_chameleon_state(cfg, 0, "cuda", 8.0, gen_device="cuda")
e = ChameleonEngine(cfg, sd, max_batch=B, max_seq_len=1152)
del sd
M = 3 * B
tok = torch.randint(0, 65536, (M,), device="cuda")
for T in (1, 64, 128, 256, 512, 768, 1024, 1100):
    pos = torch.full((M,), T - 1, dtype=torch.int32, device="cuda")
    for _ in range(3): e.forward_tokens(tok, pos)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): e.forward_tokens(tok, pos)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    kv = M * 32 * T * 128 * 2 * 2 * 32
    print(f"T={T:5d}: {dt*1e3:7.3f} ms/step   KV bytes/step {kv/1e9:6.2f} GB", flush=True)
