import sys, time, ctypes as C
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.chdir(sys.path[0])
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
def run(top_k, top_p):
    l.check(L.wmar_sample_fused(C.byref(ctx), lg.data_ptr(), B, V, past.data_ptr(), 3, 3, 1.0, top_k, top_p, q.data_ptr(), scratch.data_ptr(), tok.data_ptr(), l.stream_ptr()))
for scale in (3.0, 30.0):
    lg = (torch.randn(B, V, device="cuda") * scale)
    for top_k, top_p in ((250, 0.92), (250, -1.0), (0, 0.92), (0, -1.0)):
        for _ in range(5): run(top_k, top_p)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(200): run(top_k, top_p)
        torch.cuda.synchronize(); print("logit scale %4.1f  top_k %3d top_p %5.2f: %.1f us" % (scale, top_k, top_p, (time.time() - t0) / 200 * 1e6))
