"""Dev tool: run-to-run reproducibility of the RAR-XL step (128 rows: 64 conditions under guidance) and of the Chameleon-7B step
(48 rows: 16 prompts x 3 guidance streams) -- every repetition of a teacher-forced pass must return the first pass's logits bit for bit."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wmar_amd.utils import synth
which = sys.argv[1] if len(sys.argv) > 1 else "rar"
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bad = 0
if which == "rar":
    from wmar_amd.models.engine import RAREngine
    cfg = synth.RAR_XL
    eng = RAREngine(cfg, synth.synth_rar_state(cfg, seed=12, logit_scale=8.0), max_batch=64)
    g = torch.Generator().manual_seed(3)
    cond = torch.randint(0, 1000, (64,), generator=g) + cfg.codebook_size + 1
    both = torch.cat([cond, torch.full_like(cond, cfg.none_condition_id)]).cuda()
    toks = torch.randint(0, cfg.codebook_size, (64, 256), generator=g)
    ref = None
    for p in range(PASSES):
        cur = [] if ref is None else None
        eng.forward_position(torch.full((128,), -1, dtype=torch.int64).cuda(), both, 0)
        tok = both
        for n in range(256):
            lg = eng.forward_position(tok, both, n + 1)
            if cur is not None:
                cur.append(lg.clone())
            elif not torch.equal(lg, ref[n]):
                rows = (lg != ref[n]).any(1).nonzero().view(-1).tolist()
                print(f"RAR pass {p} step {n}: rows {rows[:8]}.. ({len(rows)}) differ, max |d| {float((lg - ref[n]).abs().max()):.2e}", flush=True)
                bad += 1
                break
            t = toks[:, n]; tok = torch.cat([t, t]).cuda()
        if cur is not None:
            ref = cur
else:
    from wmar_amd.models.engine import ChameleonEngine
    cfg = synth.CHAMELEON_7B
    sd = synth.synth_chameleon_state(cfg, 0, "cuda", 8.0, gen_device="cuda")
    eng = ChameleonEngine(cfg, sd, max_batch=16, max_seq_len=320)
    del sd
    M = 48
    g = torch.Generator().manual_seed(4)
    seq = torch.randint(0, 65536, (M, 256), generator=g).cuda()
    ref = None
    for p in range(PASSES):
        cur = [] if ref is None else None
        for t in range(256):
            pos = torch.full((M,), t, dtype=torch.int32, device="cuda")
            lg = eng.forward_tokens(seq[:, t].contiguous(), pos)
            if cur is not None:
                cur.append(lg.clone())
            elif not torch.equal(lg, ref[t]):
                rows = (lg != ref[t]).any(1).nonzero().view(-1).tolist()
                print(f"Chameleon pass {p} position {t}: rows {rows[:8]}.. ({len(rows)}) differ, max |d| {float((lg.float() - ref[t].float()).abs().max()):.2e}", flush=True)
                bad += 1
                break
        if cur is not None:
            ref = cur
print(which, "passes", PASSES, "mismatching", bad)
