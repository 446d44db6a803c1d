"""Dev tool: same-box A/B of the 256-step Taming decode loop over several builds (build_alt/<name>, "tree" = the working tree).
   python scripts/ab_loop.py tree late0 late2ch4 ...   -> per variant: min and median ms/step over rounds x reps, round-robin."""
import os, subprocess, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
ROOT = os.environ.get("WMAR_ROOT") or %r
sys.path.insert(0, ROOT)
from wmar_amd.utils import synth
from wmar_amd.models.engine import GPTEngine
cfg = synth.TAMING_GPT
eng = GPTEngine(cfg, synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0), max_batch=64)
q = torch.empty(256, 64, 16384, device="cuda").exponential_(1)
cond = (torch.arange(64) * 37 %% 1000).cuda()
out = []
for it in range(int(sys.argv[1]) + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.generate(cond, 256, q, 1.0, 250, 0.92, None, use_graph=True)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 256 * 1e3)
print("MS", " ".join("%%.4f" %% v for v in out[1:]))
''' % R
variants = sys.argv[1:] or ["tree"]
rounds, reps = int(os.environ.get("AB_ROUNDS", 3)), int(os.environ.get("AB_REPS", 4))
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        name = v.split(":")[0]                      # "tree:WMAR_XR_NW8=1" = the working tree with that environment variable set
        for kv in v.split(":")[1:]:
            env[kv.split("=")[0]] = kv.split("=", 1)[1]
        if name != "tree":
            env["WMAR_ROOT"] = os.path.join(R, "build_alt", name)
        o = subprocess.run([sys.executable, "-c", CHILD, str(reps)], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("MS")]
        if not line:
            print(v, "FAILED", o.stderr[-400:]); continue
        res[v] += [float(x) for x in line[0].split()[1:]]
for v in variants:
    if res[v]:
        print(f"{v:24s} min {min(res[v]):.4f}  median {statistics.median(res[v]):.4f}  max {max(res[v]):.4f} ms/step  (n={len(res[v])})")
