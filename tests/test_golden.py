"""The oracle against the committed golden fixtures (tests/golden, made by oracle/make_golden.py),
and -- on the GPU box -- the HIP path against the same fixtures."""
import glob
import os

import numpy as np
import pytest

from audio_feature_extraction_amd.synth import make_clip
from oracle import cpu_ref as R

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cfg*.npz")))
GOLDEN_F0 = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "f0_*.npz")))
STAT_KEYS = ("mfcc_mean", "mfcc_std", "mfcc_delta_mean", "mfcc_delta2_mean",
             "energy_mean", "energy_std", "energy_range")


def _load(path):
    g = np.load(path, allow_pickle=False)
    sr, n_fft, hop, K, idx, speechy = (int(v) for v in g["params"])
    y = make_clip(idx, sr, float(g["seconds"]), speechy=bool(speechy))
    return g, y, sr, n_fft, hop, K


def test_fixtures_present():
    assert len(GOLDEN) == 6 and len(GOLDEN_F0) == 2


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g, y, sr, n_fft, hop, K = _load(path)
    np.testing.assert_array_equal(y[:16], g["y_head"])                  # generator is deterministic
    np.testing.assert_array_equal(R.preemphasis(y, 0.97)[:16], g["y_pre_head"])
    out = R.extract_stats(y, sr=sr, frame_length=n_fft, hop_length=hop, n_mfcc=K, return_frames=True)
    assert tuple(g["trim"]) == out["trim"]
    # scipy/numpy point releases may reorder float32 sums: allow a few ulps of the row scale
    scale = np.abs(g["mfcc"]).max(axis=1, keepdims=True)
    assert (np.abs(out["mfcc"] - g["mfcc"]) / scale).max() < 5e-6
    np.testing.assert_allclose(out["rms"], g["rms"], rtol=1e-6)
    cs = float(np.abs(g["mfcc_mean"]).max())
    for k in STAT_KEYS:
        np.testing.assert_allclose(np.asarray(out[k]), g[k], rtol=1e-5, atol=1e-5 * cs * 1e-2)
        # float32 flow tracks the float64 truth
        np.testing.assert_allclose(g[k], g[k + "_f64"], rtol=2e-4, atol=2e-5 * cs * 1e-2 + 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_gpu_matches_golden(path):
    from audio_feature_extraction_amd import _native as N
    from tests.parity import MFCC_RTOL, RMS_RTOL, check_stats
    g, y, sr, n_fft, hop, K = _load(path)
    ctx = N.Context(0)
    plan = N.Plan(ctx, N.make_params(sr, n_fft, hop, K))
    out = plan.extract_batch(y, np.zeros(1, np.int64), np.array([y.size], np.int64), want_frames=True)
    assert out["status"][0] == 0 and tuple(out["trim"][0]) == tuple(g["trim"])
    fr = out["frames"][0]
    scale = np.abs(g["mfcc"]).max(axis=1, keepdims=True)
    assert (np.abs(fr["mfcc"] - g["mfcc"]) / scale).max() <= MFCC_RTOL
    np.testing.assert_allclose(fr["rms"], g["rms"], rtol=RMS_RTOL)
    ref = {k: g[k] for k in STAT_KEYS}
    check_stats(out["stats"][0], ref, K, os.path.basename(path))
    plan.close()
    ctx.close()


def _load_f0(path):
    g = np.load(path, allow_pickle=False)
    sr, n_fft, hop, idx, speechy = (int(v) for v in g["params"])
    y = make_clip(idx, sr, float(g["seconds"]), speechy=bool(speechy))
    return g, y, sr, n_fft, hop


@pytest.mark.parametrize("path", GOLDEN_F0, ids=[os.path.basename(p)[:-4] for p in GOLDEN_F0])
def test_pyin_oracle_reproduces_golden(path):
    from oracle import pyin_ref as P
    g, y, sr, n_fft, hop = _load_f0(path)
    np.testing.assert_array_equal(y[:16], g["y_head"])
    yp, _ = R.preprocess_audio(y)
    f0, _, vp = P.pyin(yp, sr=sr, frame_length=n_fft, hop_length=hop)
    same = np.isnan(f0) == np.isnan(g["f0"])
    v = ~np.isnan(f0) & ~np.isnan(g["f0"])
    same[v] &= np.abs(f0[v] - g["f0"][v]) <= 1e-9 * g["f0"][v]
    assert same.mean() >= 0.99                       # BLAS / libm point releases may flip an ill-conditioned frame
    np.testing.assert_allclose(vp, g["voiced_prob"], atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN_F0, ids=[os.path.basename(p)[:-4] for p in GOLDEN_F0])
def test_gpu_f0_matches_golden(path):
    from audio_feature_extraction_amd import _native as N
    g, y, sr, n_fft, hop = _load_f0(path)
    ctx = N.Context(0)
    plan = N.Plan(ctx, N.make_params(sr, n_fft, hop, 13))
    try:
        out = plan.f0_batch(y, np.zeros(1, np.int64), np.array([y.size], np.int64), 65.40639132514966, 2093.004522404789,
                            want_frames=True)
        f0 = out["f0_flat"][:g["f0"].size]
        same = np.isnan(f0) == np.isnan(g["f0"])
        v = ~np.isnan(f0) & ~np.isnan(g["f0"])
        same[v] &= np.abs(f0[v] - g["f0"][v]) <= 1e-9 * g["f0"][v]
        assert same.mean() >= 0.99
        if same.all():
            np.testing.assert_allclose(out["stats"][0], g["stats"], rtol=1e-10, atol=1e-12)
    finally:
        plan.close()
        ctx.close()
