"""Deterministic synthetic audio clips.

Used by the tests, ``bench.py`` and ``__graft_entry__.smoke()``; pure numpy, no
oracle or device code.  Clip ``i`` is seeded with ``1234 + i``: three sinusoids
(log-uniform 80 Hz .. 0.45*sr, amplitudes U[0.05, 0.3], random phase) + white
noise (sigma 0.05) under a slow AM envelope 0.6 + 0.4*sin(2*pi*f_m*t), peak
normalised to 0.5.  Every frame stays within 30 dB of the loudest, so the
reference's trim (``core/feature_extractor.py:72``) is a no-op and the frame
count is exactly 1 + N // hop.  The "speechy" variant adds 20 % leading and
trailing digital silence to exercise trim and the top_db clamp.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 1234


def make_clip(index: int, sr: int, seconds: float, speechy: bool = False) -> np.ndarray:
    rng = np.random.default_rng(SEED_BASE + int(index))
    n = int(round(sr * seconds))
    t = np.arange(n, dtype=np.float64) / sr
    freqs = np.exp(rng.uniform(np.log(80.0), np.log(0.45 * sr), size=3))
    amps = rng.uniform(0.05, 0.3, size=3)
    phases = rng.uniform(0.0, 2 * np.pi, size=3)
    y = np.zeros(n, dtype=np.float64)
    for f, a, p in zip(freqs, amps, phases):
        y += a * np.sin(2 * np.pi * f * t + p)
    y += 0.05 * rng.standard_normal(n)
    fm = rng.uniform(0.5, 3.0)
    y *= 0.6 + 0.4 * np.sin(2 * np.pi * fm * t)
    y *= 0.5 / np.max(np.abs(y))
    if speechy:
        lead = int(0.2 * n)
        y[:lead] = 0.0
        y[n - lead:] = 0.0
    return y.astype(np.float32)


def make_batch(n_clips: int, sr: int, seconds: float, first_index: int = 0,
               speechy: bool = False, workers: int = 1):
    """Returns (samples float32 [n_clips * N], offsets int64, lengths int64)."""
    n = int(round(sr * seconds))
    out = np.empty(n_clips * n, dtype=np.float32)

    def fill(i):
        out[i * n:(i + 1) * n] = make_clip(first_index + i, sr, seconds, speechy)

    if workers > 1 and n_clips > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(fill, range(n_clips)))
    else:
        for i in range(n_clips):
            fill(i)
    offsets = np.arange(n_clips, dtype=np.int64) * n
    lengths = np.full(n_clips, n, dtype=np.int64)
    return out, offsets, lengths
