"""The arithmetic behind the bf16-matrix-pipe kernels (wmar_amd/csrc/bx_split.h: k_conv_bx, k_qkvx_bx, k_bx), restated in numpy.

An fp32 value is split into three bf16 pieces by round-to-nearest-even of the running remainder:  h = bf16(x), m = bf16(x - h),
l = bf16(x - h - m).  The kernels rely on three properties, checked here on random and edge-case inputs:
  1. the split is EXACT: h + m + l == x (both remainders are exactly representable, so the fp32 subtractions do not round);
  2. piece magnitudes: |m| <= 2^-8 |x|, |l| <= 2^-16 |x| (measured worst case 2^-17), so the dropped products m*l, l*m, l*l stay
     below 2^-24 |w x|;
  3. the six kept products h*h + h*m + m*h + h*l + l*h + m*m, each EXACT in fp32 (8 x 8 significand bits) and summed in fp64 here,
     reproduce the fp32 product's exact value to 2^-24 relative -- the accuracy of one fp32 rounding.
(The HIP kernels are checked against the reference's outputs in tests/test_gpu_vqgan.py / test_gpu_prod_shapes.py; this file pins
the algorithm, on CPU.)"""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 (what v_cvt_pk_bf16_f32 computes for finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)          # fp32 subtraction, as v_sub_f32
    m = bf16_rne(r)
    s = (r - m).astype(np.float32)
    l = bf16_rne(s)
    return h, m, l, r, s


def _inputs():
    rng = np.random.default_rng(3)
    a = rng.standard_normal(1 << 20).astype(np.float32)
    b = (rng.standard_normal(1 << 18) * np.exp(rng.uniform(-60, 60, 1 << 18))).astype(np.float32)     # wide exponent range
    bits = rng.integers(0, 1 << 32, 1 << 18, dtype=np.uint64).astype(np.uint32).view(np.float32)        # arbitrary bit patterns
    bits = bits[np.isfinite(bits) & (np.abs(bits) > 2.0 ** -100) & (np.abs(bits) < 2.0 ** 120)]
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0 / 256, 3.0e38, 1.0e-30, 16384.0, 0.02, -0.02,
                     float(np.float32(1.00390625)), float(np.float32(1.001953125))], dtype=np.float32)
    return np.concatenate([a, b, bits, edge])


def test_split_is_exact_and_pieces_shrink():
    x = _inputs()
    h, m, l, r, s = split3(x)
    # remainders are exact: x - h and r - m are representable in fp32 (checked in fp64)
    assert np.array_equal(r.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(s.astype(np.float64), r.astype(np.float64) - m.astype(np.float64))
    assert np.array_equal(l, s), "the second remainder has at most 8 significant bits: its bf16 rounding is exact"
    tot = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.array_equal(tot, x.astype(np.float64))
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(m) <= ax * 2.0 ** -8) and np.all(np.abs(l) <= ax * 2.0 ** -16)


def test_six_products_reproduce_the_fp32_product():
    rng = np.random.default_rng(5)
    x = _inputs()
    w = rng.permutation(x)[: x.size]
    keep = (np.abs(x.astype(np.float64) * w.astype(np.float64)) < 1e37) & (np.abs(x.astype(np.float64) * w.astype(np.float64)) > 1e-30)
    x, w = x[keep], w[keep]
    xh, xm, xl, _, _ = split3(x)
    wh, wm, wl, _, _ = split3(w)
    f = np.float64
    # every piece product is exact in fp32 (two 8-bit significands): computing it in fp64 is computing it exactly
    for a, b in ((wh, xh), (wh, xm), (wm, xh), (wh, xl), (wl, xh), (wm, xm)):
        p = a.astype(f) * b.astype(f)
        assert np.array_equal(p, (a * b).astype(f)), "a piece product must be exactly representable in fp32"
    six = wl.astype(f) * xh + wh.astype(f) * xl + wm.astype(f) * xm + wm.astype(f) * xh + wh.astype(f) * xm + wh.astype(f) * xh
    exact = w.astype(f) * x.astype(f)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -24, rel.max()       # dropped: wm xl + wl xm + wl xl
    # for comparison, the fp32 multiply itself rounds by up to 2^-24
    assert (np.abs((w * x).astype(f) - exact) / np.abs(exact)).max() <= 2.0 ** -24


def test_dot_product_error_is_below_an_fp32_fma_chain():
    """K = 1536 dot products: six-product terms summed 16 at a time in fp32 (the MFMA's accumulation granularity, restated as
    fp64 partial sums rounded to fp32) against a sequential fp32 chain; both against fp64."""
    rng = np.random.default_rng(7)
    K, N = 1536, 256
    x = (rng.standard_normal((N, K)) * 2).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.04).astype(np.float32)
    f = np.float64
    exact = (x.astype(f) * w.astype(f)).sum(-1)
    xh, xm, xl, _, _ = split3(x)
    wh, wm, wl, _, _ = split3(w)
    terms = wl.astype(f) * xh + wh.astype(f) * xl + wm.astype(f) * xm + wm.astype(f) * xh + wh.astype(f) * xm + wh.astype(f) * xh
    acc = np.zeros(N, dtype=np.float32)
    for k0 in range(0, K, 16):
        acc = (acc.astype(f) + terms[:, k0:k0 + 16].sum(-1)).astype(np.float32)
    chain = np.zeros(N, dtype=np.float32)
    for k in range(K):
        chain = (chain.astype(f) + x[:, k].astype(f) * w[:, k].astype(f)).astype(np.float32)     # fused multiply-add: one rounding
    e_bx, e_chain = np.abs(acc - exact).max(), np.abs(chain - exact).max()
    assert e_bx < e_chain, (e_bx, e_chain)


def test_non_finite_and_overflowing_inputs_are_the_documented_exception():
    """bx_split.h's stated limits, pinned: an infinite operand splits into (inf, NaN, NaN) -- inf - inf in the first remainder --
    so a contraction that fp32 arithmetic would return as +-inf comes out NaN on the bf16 pipe; a finite value whose high piece
    rounds up to infinity (|x| > 0x7f7f8000 = 3.3961e38) does the same; NaN stays NaN; everything else above 2^-100 is exact.
    The opt-out for models with such activations is WMAR_NO_BX=1 / WMAR_CONV_NO_BX=1 (fp32-input MFMA kernels)."""
    with np.errstate(invalid="ignore", over="ignore"):
        h, m, l, _, _ = split3(np.array([np.inf, -np.inf, np.nan], dtype=np.float32))
        assert np.isinf(h[0]) and np.isinf(h[1]) and np.isnan(h[2])
        assert np.isnan(m).all() and np.isnan(l).all()
        big = np.array([3.4e38, np.finfo(np.float32).max], dtype=np.float32)       # both round to +inf in bf16
        hb, mb, lb, _, _ = split3(big)
        assert np.isinf(hb).all() and not np.isfinite(mb).any()
        ok = np.array([3.3895e38, -3.3895e38, 2.0 ** -99], dtype=np.float32)       # just inside the exact range
        ho, mo, lo, _, _ = split3(ok)
        assert np.array_equal(ho.astype(np.float64) + mo.astype(np.float64) + lo.astype(np.float64), ok.astype(np.float64))
