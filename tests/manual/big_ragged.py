"""One-off robustness run: thousands of ragged clips in one call (MFCC and f0), a sample checked against the oracle."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_clip
from tests.parity import check_stats, oracle_stats
from oracle import cpu_ref as R, pyin_ref as P
rng = np.random.default_rng(7)
n = 6000
base = [make_clip(i, 22050, 3.0, speechy=bool(i % 3 == 0)) for i in range(40)]
clips = []
for i in range(n):
    b = base[i % 40]
    L = int(rng.integers(1200, b.size))
    o = int(rng.integers(0, b.size - L + 1))
    clips.append(b[o:o + L])
lengths = np.array([c.size for c in clips], np.int64)
offsets = np.concatenate([[0], np.cumsum((lengths + 3) // 4 * 4)[:-1]]).astype(np.int64)
buf = np.zeros(int(offsets[-1] + lengths[-1] + 8), np.float32)
for c, o in zip(clips, offsets):
    buf[o:o + c.size] = c
ctx = N.Context(0); plan = N.Plan(ctx, N.make_params(22050, 1024, 256, 13))
t0 = time.perf_counter(); out = plan.extract_batch(buf, offsets, lengths); t1 = time.perf_counter()
f0 = plan.f0_batch(buf, offsets, lengths, P.C2_HZ, P.C7_HZ); t2 = time.perf_counter()
print("clips", n, "samples", buf.size, "mfcc call %.1f ms" % ((t1 - t0) * 1e3), "f0 call %.1f ms" % ((t2 - t1) * 1e3),
      "status counts", np.bincount(out["status"], minlength=3).tolist(), np.bincount(f0["status"], minlength=3).tolist())
bad = 0
for i in rng.choice(n, 40, replace=False):
    try:
        ref = oracle_stats(clips[i], 22050, 1024, 256, 13)
    except ValueError:
        assert out["status"][i] == N.CLIP_TOO_SHORT; continue
    assert out["status"][i] == 0
    check_stats(out["stats"][i], ref, 13, f"big{i}")
    yp, _ = R.preprocess_audio(clips[i])
    if yp.size < 60000:
        r = P.extract_f0(yp)
        if abs(f0["stats"][i][2] - r["f0_missing_rate"]) > 0.05 or abs(f0["stats"][i][0] - r["f0_mean"]) > 0.01 * max(r["f0_mean"], 1):
            bad += 1; print("f0 differs", i, f0["stats"][i], r)
print("checked 40 clips; f0 outliers", bad)
