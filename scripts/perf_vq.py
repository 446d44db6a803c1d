"""Dev tool: time the full-size Taming VQGAN decode/encode (random weights)."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wmar_amd.utils import synth
from wmar_amd.models.engine import VQGANEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = synth.TAMING_VQ
sd = synth.synth_vq_state_fast(cfg, 0, "cuda")
eng = VQGANEngine(cfg, sd, max_batch=B)
print("engine bytes", eng.device_bytes / 1e9)
codes = torch.randint(0, 16384, (B, 256), device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    img = eng.decode(codes)
    torch.cuda.synchronize(); t1 = time.time()
    c2 = eng.encode(img)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"B={B} decode {1e3*(t1-t0):.1f} ms ({252.7*B/(t1-t0)/1e3:.1f} TF/s)  encode {1e3*(t2-t1):.1f} ms ({140.5*B/(t2-t1)/1e3:.1f} TF/s)  l0={(c2!=codes).float().mean().item():.3f}")
