"""Inputs of the bulk sampler fixture (tests/golden/sampler_bulk.npz), regenerated from the chunk index alone so that only the
reference's TOKENS need to be stored.  Used by tests/golden/make_golden.py (which runs the reference's sampling chain on them)
and by the parity tests.  CPU torch generators: identical on every machine with this torch build."""
import torch

BULK_CHUNKS, BULK_ROWS, BULK_V = 48, 512, 16384
BULK_PARAMS = [(250, 0.92, 1.0), (100, 0.8, 1.3), (None, 0.95, 0.9), (50, None, 1.0), (None, None, 1.0), (250, 0.92, 0.7)]
BULK_DELTA, BULK_GAMMA = 2.0, 0.25


def bulk_rows(c):
    """Chunk c: context tokens [B,1], logits [B,V], (top_k, top_p, temperature), watermark on/off."""
    g = torch.Generator().manual_seed(77000 + c)
    ctx = torch.randint(0, BULK_V, (BULK_ROWS, 1), generator=g)
    lg = torch.randn(BULK_ROWS, BULK_V, generator=g) * [1.0, 10.0, 40.0][c % 3]
    kind = (c // 3) % 4
    if kind == 2:
        lg = torch.round(lg * 2.0) / 2.0             # tie-heavy rows: many exactly equal logits
    elif kind == 3:
        lg = lg.bfloat16().float()                    # bf16-valued logits: ties in the low mantissa bits
    top_k, top_p, T = BULK_PARAMS[c % len(BULK_PARAMS)]
    use_wm = (c % 8) != 7
    return ctx, lg, top_k, top_p, T, use_wm


def bulk_noise(c):
    """The Exp(1) noise torch.multinomial draws for chunk c (torch.manual_seed(88000 + c) precedes the call)."""
    torch.manual_seed(88000 + c)
    return torch.empty(BULK_ROWS, BULK_V).exponential_(1)
