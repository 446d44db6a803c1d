import sys, time, ctypes as C
import torch
sys.path.insert(0, ".")
import wmar_amd._lib as l
from tests.test_gpu_watermark import _wm
import json
kat = json.load(open("tests/golden/key_kat.json"))
wm = _wm(kat["keys"]["taming"])
L = l.load()
B, V = 64, 16384
lg = (torch.randn(B, V, device="cuda") * 3)
q = torch.empty(B, V, device="cuda").exponential_(1)
past = torch.randint(0, V, (B, 3), device="cuda")
scratch = torch.empty_like(lg); tok = torch.empty(B, dtype=torch.int64, device="cuda")
ctx = wm.wm_ctx()
def run():
    l.check(L.wmar_sample_fused(C.byref(ctx), lg.data_ptr(), B, V, past.data_ptr(), 3, 3, 1.0, 250, 0.92, q.data_ptr(), scratch.data_ptr(), tok.data_ptr(), l.stream_ptr()))
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(200): run()
torch.cuda.synchronize(); print("sampler %.1f us" % ((time.time() - t0) / 200 * 1e6))
