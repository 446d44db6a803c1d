"""Shared pieces of the full-size image-tokenizer parity tests (tests/golden/fullsize_vq_vectors.npz, written by
make_golden.py::fullsize_vq_vectors from the reference's own Decoder / Encoder / VectorQuantizer modules at the shapes the
benchmarks run)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# a re-encoded code may differ from the reference's only where the reference's own margin between the best and the second-best
# codebook entry (squared distances of magnitude ~100) is below what a 5e-4 perturbation of the 256-dim pre-quantisation vector can move
NEAR_TIE_MARGIN = 5e-2


def load():
    return np.load(os.path.join(HERE, "golden", "fullsize_vq_vectors.npz"))


def sub(img: np.ndarray) -> np.ndarray:
    """every 4th pixel, phase (b, 2b + 1) for image b -- the sampling make_golden.py stored"""
    return np.stack([img[b, :, (b % 4)::4, ((2 * b + 1) % 4)::4] for b in range(img.shape[0])])


def check_codes(got: np.ndarray, want: np.ndarray, margin: np.ndarray, what: str):
    got, want = got.reshape(-1), want.reshape(-1)
    bad = np.nonzero(got != want)[0]
    assert len(bad) <= 0.01 * len(want), f"{what}: {len(bad)} of {len(want)} re-encoded codes differ"
    assert (margin[bad] < NEAR_TIE_MARGIN).all(), f"{what}: codes differ at positions with a clear margin: {margin[bad]}"
    return len(bad)
