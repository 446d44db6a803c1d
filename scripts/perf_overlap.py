"""Dev experiment: VQGAN of batch i on a second stream while batch i+1 is being sampled."""
import sys, time
import torch
sys.path.insert(0, ".")
from wmar_amd.utils import synth
from wmar_amd.models.taming_wrapper import TamingARMMWrapper
from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
B = 64
model = TamingARMMWrapper.synthetic(synth.TAMING_GPT, synth.TAMING_VQ, 0, "cuda", B, 30.0)
wm = GentimeWatermark(model.get_vq(), 16384, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
model.set_watermarker(wm); wm.key_table(); _ = model.model.vq_engine
cond = (torch.arange(B) * 37 % 1000).cuda()
GEN = dict(temperature=1.0, top_k=250, top_p=0.92)
def serial(n):
    for _ in range(n):
        c = model.sample(cond, GEN, True); im = model.codes_to_images(c); c2 = model.images_to_codes(im); pv = wm.detect(c2)
    return pv
s2 = torch.cuda.Stream()
def piped(n):
    prev = None
    ev = None
    for i in range(n + 1):
        if i < n:
            c = model.sample(cond, GEN, True)
            e = torch.cuda.Event(); e.record()
        if prev is not None:
            with torch.cuda.stream(s2):
                s2.wait_event(pe)
                im = model.codes_to_images(prev); c2 = model.images_to_codes(im); pv = wm.detect(c2)
        if i < n: prev, pe = c, e
    torch.cuda.current_stream().wait_stream(s2)
    return pv
serial(1); torch.cuda.synchronize()
for fn in (serial, piped, serial, piped):
    torch.cuda.synchronize(); t0 = time.time(); fn(3); torch.cuda.synchronize(); dt = time.time() - t0
    print(fn.__name__, f"{dt/3*1e3:.1f} ms/batch  {B*3/dt:.1f} img/s")
