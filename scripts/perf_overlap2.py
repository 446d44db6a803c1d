"""Dev experiment: does ANY work on a second stream overlap with the captured decode loop?  (a) torch matmuls (b) the VQGAN engine."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wmar_amd.utils import synth
from wmar_amd.models.taming_wrapper import TamingARMMWrapper
from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
B = 64
model = TamingARMMWrapper.synthetic(synth.TAMING_GPT, synth.TAMING_VQ, 0, "cuda", B, 30.0)
wm = GentimeWatermark(model.get_vq(), 16384, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
model.set_watermarker(wm); wm.key_table(); _ = model.model.vq_engine
cond = (torch.arange(B) * 37 % 1000).cuda()
GEN = dict(temperature=1.0, top_k=250, top_p=0.92)
q = model.draw_noise(256, B)
a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
codes = model.sample(cond, GEN, True, q=q); torch.cuda.synchronize()
s2 = torch.cuda.Stream(); s1 = torch.cuda.Stream()
def t(fn, n=2):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
def mm():
    for _ in range(40): torch.mm(a, b)
def vq():
    im = model.codes_to_images(codes); model.images_to_codes(im)
def samp():
    if os.environ.get('SAMP_STREAM'):
        with torch.cuda.stream(s1): model.sample(cond, GEN, True, q=q)
        torch.cuda.current_stream().wait_stream(s1)
    else:
        model.sample(cond, GEN, True, q=q)
def both(other):
    def f():
        # the second stream's work is enqueued FIRST: the 256 graph launches of the loop fill the hardware queue and block the host
        t1 = time.time()
        with torch.cuda.stream(s2): other()
        t2 = time.time(); samp(); t0 = time.time() - t2 + t1
        torch.cuda.current_stream().wait_stream(s2)
        f.host = (t0 - t1, t2 - t1)
    return f
print(f"sample alone {t(samp):.1f} ms   40 matmuls alone {t(mm):.1f} ms   vqgan alone {t(vq):.1f} ms")
f = both(mm); print(f"sample || matmuls {t(f):.1f} ms  (host: enqueue sample {f.host[0]*1e3:.1f} ms, enqueue other {f.host[1]*1e3:.1f} ms)")
f = both(vq); print(f"sample || vqgan   {t(f):.1f} ms  (host: enqueue sample {f.host[0]*1e3:.1f} ms, enqueue other {f.host[1]*1e3:.1f} ms)")
