"""Randomised ragged batches through the C-ABI against the oracle: lengths from a few samples to a few
seconds, silences at either end (trim), gain changes, int16 and float32 packing, aligned and unaligned
offsets -- the clip-edge, trimmed-span and scalar-load routes of every kernel."""
import numpy as np
import pytest

from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_clip
from tests.parity import check_frames, check_stats, oracle_stats

pytestmark = pytest.mark.gpu
SR = 22050


def _random_clip(rng, i, sr=SR, scale=1):
    kind = rng.integers(0, 6)
    n = int(rng.integers(2, 9000 * scale)) if kind == 0 else int(rng.integers(2304 * scale, 45000 * scale))
    y = make_clip(300 + i, sr, max(n / sr, 0.01))[:n].copy()
    if y.size < n:
        y = np.resize(y, n)
    if kind == 1:                                   # leading / trailing digital silence
        a, b = int(rng.integers(0, n // 3)), int(rng.integers(0, n // 3))
        y[:a] = 0
        y[n - b:] = 0
    elif kind == 2:                                 # quiet tail well below the 30 dB trim threshold
        a = int(rng.integers(n // 2, n))
        y[a:] *= 1e-3
    elif kind == 3:                                 # loud burst in a quiet clip
        y *= 0.01
        a = int(rng.integers(0, max(1, n - 2000)))
        y[a:a + 1500] *= 60
    elif kind == 4:                                 # low level: exercises amin / top_db clamps
        y *= float(10 ** rng.uniform(-4, -2))
    return y.astype(np.float32)


# the three wave-level frame kernels (n_fft 1024: one frame pair per wave; 2048: one frame, real-FFT split; 512: two
# pairs per wave), each with float32 and int16 packing
@pytest.mark.parametrize("seed,fmt,cfg", [(1, "f32", (22050, 1024, 256, 13)), (2, "f32", (22050, 1024, 256, 13)),
                                          (3, "s16", (22050, 1024, 256, 13)),
                                          (4, "f32", (44100, 2048, 512, 20)), (5, "s16", (44100, 2048, 512, 20)),
                                          (6, "f32", (16000, 512, 128, 40)), (7, "s16", (16000, 512, 128, 40))])
def test_random_ragged_batches(seed, fmt, cfg):
    rng = np.random.default_rng(seed)
    ctx = N.Context(0)
    SR, n_fft, hop, K = cfg
    plan = N.Plan(ctx, N.make_params(SR, n_fft, hop, K))
    try:
        clips = [_random_clip(rng, 100 * seed + i, SR, max(1, n_fft // 1024)) for i in range(28)]
        if fmt == "s16":
            q = [np.clip(np.rint(c.astype(np.float64) * 32768), -32768, 32767).astype(np.int16) for c in clips]
            clips = [(c.astype(np.float32) / np.float32(32768.0)) for c in q]
        lengths = np.array([c.size for c in clips], np.int64)
        gap = rng.integers(0, 7, size=len(clips))                      # unaligned packing
        offsets = np.concatenate([[int(rng.integers(0, 4))], np.cumsum(lengths + gap)[:-1] + int(rng.integers(0, 4))]).astype(np.int64)
        buf = np.zeros(int(offsets[-1] + lengths[-1] + 8), np.int16 if fmt == "s16" else np.float32)
        for c, o, qq in zip(clips, offsets, q if fmt == "s16" else clips):
            buf[o:o + c.size] = qq
        out = plan.extract_batch(buf, offsets, lengths, fmt=N.FMT_S16 if fmt == "s16" else N.FMT_F32, want_frames=True)
        n_ok = 0
        for i, c in enumerate(clips):
            try:
                ref = oracle_stats(c, SR, n_fft, hop, K)
            except ValueError:
                assert out["status"][i] == N.CLIP_TOO_SHORT, (i, c.size, out["status"][i])
                continue
            assert out["status"][i] == N.CLIP_OK, (i, c.size, out["status"][i])
            assert tuple(out["trim"][i]) == tuple(ref["trim"]), (i, out["trim"][i], ref["trim"])
            check_frames(out["frames"][i], ref, f"rand{seed}-{i}")
            check_stats(out["stats"][i], ref, K, f"rand{seed}-{i}")
            n_ok += 1
        assert n_ok >= 15
    finally:
        plan.close()
        ctx.close()
