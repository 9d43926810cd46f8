"""A call that the library splits into several chunks must give every clip the rows it gets alone.

afx_extract_batch splits at 32768 clips and afx_f0_batch at 1.28 M frames; both limits can be lowered for a test
through AFX_TEST_CHUNK_CLIPS / AFX_TEST_F0_CHUNK_FRAMES, which the library reads once when it loads -- hence the child
process.  Round 1 copied the whole per-frame range back after every chunk and so overwrote earlier chunks' rows."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_clip
sr = 22050
clips = [make_clip(40 + i, sr, 0.35 + 0.11 * (i % 5), speechy=(i % 3 == 0)) for i in range(11)]
lens = np.array([c.size for c in clips], np.int64)
offs = np.zeros(len(clips), np.int64); offs[1:] = np.cumsum((lens + 3) // 4 * 4)[:-1]
buf = np.zeros(int(offs[-1] + lens[-1]), np.float32)
for c, o in zip(clips, offs): buf[o:o + c.size] = c
ctx = N.Context(0); plan = N.Plan(ctx, N.make_params(sr, 1024, 256, 13))
out = plan.extract_batch(buf, offs, lens, want_frames=True)
f0 = plan.f0_batch(buf, offs, lens, 65.40639132514966, 2093.004522404789, want_frames=True)
bad = 0
for i, c in enumerate(clips):
    one = plan.extract_batch(c, np.zeros(1, np.int64), np.array([c.size], np.int64), want_frames=True)
    assert out["status"][i] == one["status"][0]
    for k in ("mfcc", "mfcc_delta", "mfcc_delta2", "rms"):
        if not np.array_equal(out["frames"][i][k], one["frames"][0][k]): bad += 1
    if not np.array_equal(out["stats"][i], one["stats"][0]): bad += 1
    o1 = plan.f0_batch(c, np.zeros(1, np.int64), np.array([c.size], np.int64), 65.40639132514966, 2093.004522404789, want_frames=True)
    T = 1 + c.size // 256
    a = f0["f0_flat"][f0["f0_offsets"][i]: f0["f0_offsets"][i] + T]; b = o1["f0_flat"][:T]
    if not np.array_equal(np.isnan(a), np.isnan(b)) or not np.allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=0, atol=0): bad += 1
    if not np.array_equal(f0["stats"][i], o1["stats"][0]): bad += 1
print("BAD", bad)
'''


@pytest.mark.gpu
def test_multi_chunk_calls_keep_every_clips_rows():
    env = dict(os.environ, AFX_TEST_CHUNK_CLIPS="3", AFX_TEST_F0_CHUNK_FRAMES="300", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "BAD 0" in r.stdout, r.stdout[-500:] + r.stderr[-500:]


@pytest.mark.gpu
def test_more_than_32768_clips_with_per_frame_output():
    """The real limit: 33 000 tiny clips in one call cross afx_extract_batch's 32768-clip chunk boundary."""
    import numpy as np
    from audio_feature_extraction_amd import _native as N
    n, ln, sr = 33000, 2560, 22050
    rng = np.random.default_rng(7)
    base = (0.1 * rng.standard_normal(ln + n)).astype(np.float32)
    tone = (0.2 * np.sin(2 * np.pi * 440.0 * np.arange(ln) / sr)).astype(np.float32)
    buf = np.empty(n * ln, np.float32)
    for i in range(n):                                    # every clip different: a shifted noise window + a tone
        buf[i * ln:(i + 1) * ln] = base[i:i + ln] + tone * (0.5 + (i % 7) / 7.0)
    offs = np.arange(n, dtype=np.int64) * ln
    lens = np.full(n, ln, np.int64)
    ctx = N.Context(0)
    plan = N.Plan(ctx, N.make_params(sr, 1024, 256, 13))
    try:
        out = plan.extract_batch(buf, offs, lens, want_frames=True)
        assert (out["status"] == 0).all() and (out["nframes"] > 8).all()
        for i in (0, 1, 32766, 32767, 32768, 32769, n - 1):
            one = plan.extract_batch(buf[i * ln:(i + 1) * ln].copy(), np.zeros(1, np.int64), np.array([ln], np.int64), want_frames=True)
            np.testing.assert_array_equal(out["stats"][i], one["stats"][0], err_msg=str(i))
            for k in ("mfcc", "mfcc_delta", "mfcc_delta2", "rms"):
                np.testing.assert_array_equal(out["frames"][i][k], one["frames"][0][k], err_msg=f"{i} {k}")
    finally:
        plan.close()
        ctx.close()
