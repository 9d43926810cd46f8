"""Sibling frame features on the GPU (k_frames3s<DESC> through afx_spectral_batch) against the oracle's restatement of
librosa.feature.spectral_centroid / _bandwidth / _rolloff / _contrast at their defaults
(04_feature_extraction_experiment/feature_extractor.py:497-506).  "Parity unpinned" as for the MFCC path: the oracle is
pinned by the known answers in tests/test_oracle_kat.py, not by a librosa run."""
import numpy as np
import pytest

from audio_feature_extraction_amd import AudioFeatureExtractor
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_clip
from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu


def _clips(sr):
    rng = np.random.default_rng(3)
    t = np.arange(int(1.2 * sr)) / sr
    tone = (0.3 * np.sin(2 * np.pi * 100 * sr / 2048 * t)).astype(np.float32)          # bin-centred
    return [make_clip(31, sr, 1.5), make_clip(32, sr, 2.0, speechy=True), tone,
            (0.1 * rng.standard_normal(int(0.9 * sr))).astype(np.float32), make_clip(33, sr, 0.11)]


@pytest.mark.parametrize("sr", [22050, 44100, 16000])
def test_spectral_descriptors_match_the_oracle(sr):
    ctx = N.Context(0)
    plan = N.Plan(ctx, N.make_params(sr, 2048, 512, 13, 128, "hann"))
    try:
        clips = _clips(sr)
        lens = np.array([c.size for c in clips], np.int64)
        offs = np.zeros(len(clips), np.int64)
        offs[1:] = np.cumsum((lens + 3) // 4 * 4)[:-1]
        buf = np.zeros(int(offs[-1] + lens[-1]), np.float32)
        for c, o in zip(clips, offs):
            buf[o:o + c.size] = c
        out = plan.spectral_batch(buf, offs, lens)
        assert (out["status"] == 0).all()
        for i, c in enumerate(clips):
            g = out["clips"][i]
            cen, bw, ro = R.spectral_centroid(c, sr)[0], R.spectral_bandwidth(c, sr)[0], R.spectral_rolloff(c, sr)[0]
            pk, vl = R.spectral_contrast_parts(c, sr)
            assert g["centroid"].shape == cen.shape and g["peak"].shape == pk.shape
            nyq = sr / 2
            assert np.abs(g["centroid"] - cen).max() <= 2e-5 * nyq, (sr, i)
            # a pure tone's bandwidth (a few Hz) is the f^2-weighted float32 noise floor of the far bins: ill-conditioned,
            # held to 5e-4 of Nyquist; every other frame to 5e-5
            bw_tol = np.where(bw > 1e-2 * nyq, 5e-5 * nyq, 5e-4 * nyq)
            assert (np.abs(g["bandwidth"] - bw) <= bw_tol).all(), (sr, i, np.abs(g["bandwidth"] - bw).max())
            # roll-off: a running float32 sum decides a bin; numpy adds in bin order, the GPU per lane and then across
            # lanes -- a frame whose 85 % point falls within rounding of a bin boundary may land one bin off
            off_by = np.abs(g["rolloff"] - ro) / (sr / 2048)
            assert off_by.max() <= 1.001 and (off_by > 0.5).mean() <= 0.02, (sr, i, off_by.max())
            sc = max(float(pk.max()), 1e-30)
            assert np.abs(g["peak"] - pk).max() <= 2e-5 * sc and np.abs(g["valley"] - vl).max() <= 2e-5 * sc, (sr, i)
    finally:
        plan.close()
        ctx.close()


def test_extract_spectral_features_dict():
    sr = 22050
    y = make_clip(40, sr, 3.0, speechy=True)
    ex = AudioFeatureExtractor(sr=sr)
    d = ex.extract_spectral_features(y)
    ref = R.extract_spectral_features(y, sr)
    assert list(d) == list(ref)                      # key order of feature_extractor.py:509-518
    for k in d:
        assert type(d[k]) is float
        tol = 1e-4 * max(abs(float(ref[k])), 1.0) if "rolloff" not in k else 2e-3 * abs(float(ref[k]))
        assert abs(d[k] - float(ref[k])) <= tol, (k, d[k], float(ref[k]))
    with pytest.raises(NotImplementedError):
        AudioFeatureExtractor(sr=8000).extract_spectral_features(make_clip(41, 8000, 1.0))     # librosa: band exceeds Nyquist
