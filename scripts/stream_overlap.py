import torch, time, os
print({k: v for k, v in os.environ.items() if "HIP" in k or "HSA" in k or "GPU_" in k or "ROC" in k})
x = torch.randn(64, 256, device="cuda"); w = torch.randn(256, 256, device="cuda")
def work(n):
    y = x
    for _ in range(n):
        y = torch.tanh(y @ w)   # tiny kernels: a few workgroups each
    return y
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
work(200); torch.cuda.synchronize()
t0 = time.time(); work(2000); torch.cuda.synchronize(); t1 = time.time() - t0
t0 = time.time()
with torch.cuda.stream(s1): work(2000)
with torch.cuda.stream(s2): work(2000)
torch.cuda.synchronize(); t2 = time.time() - t0
print(f"one stream 2000 iters: {t1*1e3:.1f} ms; two streams 2x2000: {t2*1e3:.1f} ms (concurrent if ~= one stream)")
# big elementwise on few CUs
a = torch.randn(1 << 22, device="cuda")
def work2(n):
    b = a
    for _ in range(n): b = torch.sin(b)
    return b
work2(10); torch.cuda.synchronize()
t0 = time.time(); work2(300); torch.cuda.synchronize(); t1 = time.time() - t0
t0 = time.time()
with torch.cuda.stream(s1): work2(300)
with torch.cuda.stream(s2): work2(300)
torch.cuda.synchronize(); t2 = time.time() - t0
print(f"elementwise one: {t1*1e3:.1f} ms; two streams: {t2*1e3:.1f} ms")
