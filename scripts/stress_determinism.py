"""Dev tool: repeat the full-size Taming generation / tokenizer calls and report any run that differs from the first
(graph replay, eager loop on a sub-batch, VQGAN decode / encode)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # WMAR_ROOT: another checkout (bisecting)
from wmar_amd.utils import synth
from wmar_amd.models.taming_wrapper import TamingARMMWrapper
from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
GEN_ONLY = len(sys.argv) > 2 and sys.argv[2] == "gen"
# dirty the allocator first: freshly mapped HBM is zero, recycled blocks are not
if os.environ.get("JUNK", "nan") != "none":
    junk = [torch.full((1 << 28,), float(os.environ.get("JUNK", "nan")), device="cuda") for _ in range(8)]
    del junk
m = TamingARMMWrapper.synthetic(synth.TAMING_GPT, synth.TAMING_VQ, seed=0, max_batch=64)
wm = GentimeWatermark(m.get_vq(), m.get_total_vocab_size(), SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
m.set_watermarker(wm)
gp = {"temperature": 1.0, "top_k": 250, "top_p": 0.92}
cond = [(i * 37) % 1000 for i in range(64)]
torch.manual_seed(1)
q = m.draw_noise(256, 64)
a = m.sample(cond, gp, apply_watermark=True, q=q)
q8 = q[:, :8].contiguous()
bad = {"graph": 0, "eager8": 0, "graph8": 0, "dec": 0, "enc": 0}
img = m.codes_to_images(a[:16]); c1 = m.images_to_codes(img)
for it in range(N):
    x = m.sample(cond, gp, apply_watermark=True, q=q)
    if not torch.equal(a, x):
        bad["graph"] += 1; d = (a != x).nonzero(); print("graph replay differs: first (row, pos)", d[0].tolist(), "count", len(d))
    if GEN_ONLY:
        continue
    m.use_graph = False
    x = m.sample(cond[:8], gp, apply_watermark=True, q=q8)
    m.use_graph = True
    if not torch.equal(a[:8], x):
        bad["eager8"] += 1; d = (a[:8] != x).nonzero(); print("eager B=8 differs: first (row, pos)", d[0].tolist(), "count", len(d))
    x = m.sample(cond[:8], gp, apply_watermark=True, q=q8)
    if not torch.equal(a[:8], x):
        bad["graph8"] += 1; d = (a[:8] != x).nonzero(); print("graph B=8 differs: first (row, pos)", d[0].tolist(), "count", len(d))
    x = m.codes_to_images(a[:16])
    if not torch.equal(img, x):
        bad["dec"] += 1; print("decode differs: max", float((img - x).abs().max()), "n", int((img != x).sum()))
    x = m.images_to_codes(img)
    if not torch.equal(c1, x):
        bad["enc"] += 1; print("encode differs: n", int((c1 != x).sum()))
print("runs", N, bad)
