"""Dev tool (round 5): the per-launch timeline of the five per-layer launches INSIDE the captured 256-step loop, from in-kernel stamps.

Build a stamped library and run on the GPU box:
    WMAR_EXTRA_HIPCC_FLAGS=-DWMAR_STAMPS scripts/mk_alt.sh stamps ; python -m wmar_amd.build   # (the second restores the product build)
    WMAR_ROOT=build_alt/stamps WMAR_STAMPS=1 python scripts/stamp_table.py [steps=129] > profiles/r05_stamp_table.md

Wave 0 of every workgroup records s_memrealtime (100 MHz, chip-wide) at entry / exit and s_memtime (shader cycles) at entry, first
operands landed, main loop done, exit (decoder_kernels.h, WMAR_ST_*).  The last replay of the graph (cache length = steps) is read
back; layers 1..47 are averaged.  Columns:
  gap        first workgroup of this launch in  -  last workgroup of the previous launch out          (kernel boundary in the graph)
  ramp       last workgroup in  -  first workgroup in                                                 (dispatch of the grid)
  first-load entry -> first operands landed (per workgroup, mean / max)
  steady     first operands landed -> main loop done (mean / max)
  tail       main loop done -> exit: in-workgroup reduction, epilogue stores acknowledged (mean / max)
  drain      last workgroup out  -  mean workgroup out                                                (waiting for stragglers)
  total      first workgroup in -> last workgroup out
k_bx_xr splits its tail into barrier (phase-1 stores acknowledged -> XCD barrier passed) and phase 2 (fold + statistics, the four
folding workgroups per XCD); the attention's "first-load" is the prologue's operands and "finish" the q/k/v algebra + cache append."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wmar_amd import _lib  # noqa: E402
from wmar_amd.models.engine import GPTEngine  # noqa: E402
from wmar_amd.utils import synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 129
cfg = synth.TAMING_GPT
eng = GPTEngine(cfg, synth.synth_gpt_state_fast(cfg, 0, "cuda", logit_scale=30.0), max_batch=64)
B = 64
q = torch.empty(steps, B, cfg.vocab_size, device="cuda").exponential_(1)
cond = (torch.arange(B) * 37 % 1000).cuda()
for _ in range(2):
    eng.generate(cond, steps, q, 1.0, 250, 0.92, None, use_graph=True)
torch.cuda.synchronize()
L, R, U = cfg.n_layer, 5, 1536
buf = np.zeros(L * R * U * 8, dtype=np.uint64)
fn = eng._L.wmar_gpt_debug_stamps
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
_lib.check(fn(eng._h, buf.ctypes.data, buf.size))
st = buf.reshape(L, R, U, 8).astype(np.int64)
names = ["qkv+fold (k_qkvx_bx<6>)", "attention (k_attn_decode<64,1>)", "proj+fold+LN2 (k_bx_xr<6,4>)", "FC1 (k_fc1x)", "FC2 (k_gemm)"]
plan = eng.plan_info(64)
print(f"# In-loop launch timeline, Taming 48L x 1536, 64 rows, graph replay, step {steps} (cache length {steps}); layers 1..{L - 1} averaged")
print(f"# plan: qkv={plan['qkv']}; proj={plan['proj']}; fc1={plan['fc1']}; fc2={plan['fc2']}")
rows = {r: [] for r in range(R)}
prev_out = None
for l in range(L):
    for r in range(R):
        s = st[l, r]
        on = s[:, 1] != 0
        if not on.any():
            prev_out = None
            continue
        s = s[on]
        rt_in, rt_out = s[:, 0] * 10.0, s[:, 5] * 10.0            # ns
        cyc = (s[:, 4] - s[:, 1]).astype(np.float64)
        ghz = np.median(cyc / np.maximum(rt_out - rt_in, 1.0))     # shader cycles per ns, this launch
        def us(a):
            return a / ghz / 1e3
        landed = np.where(s[:, 2] > 0, s[:, 2], s[:, 1])
        d = dict(n=int(on.sum()), gap=(rt_in.min() - prev_out) / 1e3 if prev_out is not None else np.nan,
                 ramp=(rt_in.max() - rt_in.min()) / 1e3, total=(rt_out.max() - rt_in.min()) / 1e3,
                 drain=(rt_out.max() - rt_out.mean()) / 1e3, ghz=ghz,
                 first=us(landed - s[:, 1]).mean(), first_max=us(landed - s[:, 1]).max())
        if r == 1:      # attention: [6] = q/k/v finished
            d.update(finish=us(s[:, 6] - landed).mean(), steady=us(s[:, 3] - s[:, 6]).mean(), steady_max=us(s[:, 3] - s[:, 6]).max(),
                     tail=us(s[:, 4] - s[:, 3]).mean(), tail_max=us(s[:, 4] - s[:, 3]).max())
        elif r == 2:    # k_bx_xr: [3] = phase-1 stores acknowledged, [6] = barrier passed
            fold = s[:, 4] - s[:, 6]
            d.update(steady=us(s[:, 3] - landed).mean(), steady_max=us(s[:, 3] - landed).max(), barrier=us(s[:, 6] - s[:, 3]).mean(),
                     barrier_max=us(s[:, 6] - s[:, 3]).max(), tail=us(np.sort(fold)[-32:]).mean(), tail_max=us(fold).max())
        else:
            d.update(steady=us(s[:, 3] - landed).mean(), steady_max=us(s[:, 3] - landed).max(), tail=us(s[:, 4] - s[:, 3]).mean(),
                     tail_max=us(s[:, 4] - s[:, 3]).max())
        if l >= 1:
            rows[r].append(d)
        prev_out = rt_out.max()
print()
print("| launch | workgroups | gap | ramp | first-load mean / max | steady mean / max | tail mean / max | drain | total | GHz |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot = 0.0
for r in range(R):
    ds = rows[r]
    if not ds:
        continue
    m = lambda k: float(np.nanmean([d[k] for d in ds if k in d]))
    extra = ""
    if r == 1:
        extra = f" (+ finish q/k/v {m('finish'):.2f})"
    if r == 2:
        extra = f" (+ XCD barrier {m('barrier'):.2f} / {m('barrier_max'):.2f})"
    print(f"| {names[r]} | {ds[0]['n']} | {m('gap'):.2f} | {m('ramp'):.2f} | {m('first'):.2f} / {m('first_max'):.2f}{extra if r == 1 else ''} | "
          f"{m('steady'):.2f} / {m('steady_max'):.2f} | {m('tail'):.2f} / {m('tail_max'):.2f}{extra if r == 2 else ''} | {m('drain'):.2f} | {m('total'):.2f} | {m('ghz'):.2f} |")
    tot += m('gap') + m('total')
print(f"\nsum over the five launches of (gap + total): {tot:.1f} us per layer; x {L} layers = {tot * L / 1e3:.3f} ms of the step (microseconds; "
      f"stamped build: the waits behind the 'landed' stamps make it a few percent slower than the product)")
