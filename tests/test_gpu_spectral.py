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


def _truth64(y, sr, n_fft=2048, hop=512):
    """Centroid and bandwidth of the float32 signal in float64 arithmetic throughout (centred zero-padded Hann STFT), and
    for each the change it would suffer if every bin's magnitude were off by ONE float32 ulp of the frame's largest
    magnitude -- the most any float32 FFT can promise (bins that are truly zero come out as |noise| > 0, so the errors all
    push the f^2-weighted bandwidth the same way).  That is the conditioning of the two descriptors, frame by frame."""
    from scipy.signal import get_window
    yp = np.pad(np.asarray(y, np.float64), n_fft // 2)
    T = 1 + (yp.size - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(T)[None, :]
    S = np.abs(np.fft.rfft(yp[idx] * get_window("hann", n_fft, fftbins=True)[:, None], axis=0))
    freq = np.fft.rfftfreq(n_fft, 1.0 / sr)[:, None]
    norm = S.sum(axis=0, keepdims=True)
    norm = np.where(norm < np.finfo(np.float32).tiny, 1.0, norm)
    P = S / norm
    cen = (freq * P).sum(axis=0)
    dev = freq - cen[None, :]
    bw = np.sqrt((P * dev ** 2).sum(axis=0))
    ulp = np.finfo(np.float32).eps * S.max(axis=0) / norm[0]           # one ulp of the largest magnitude, as a share of the L1 norm
    cen_cond = ulp * np.abs(dev).sum(axis=0)
    bw_cond = np.sqrt(bw ** 2 + ulp * (dev ** 2).sum(axis=0)) - bw
    return cen, bw, cen_cond, bw_cond


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
            cen64, bw64, cen_cond, bw_cond = _truth64(c, sr)
            # float64 truth adjudicates: 2e-5 / 5e-5 of Nyquist, or -- where a descriptor is ill-conditioned -- what one
            # float32 ulp of the frame's largest magnitude on every bin would do to it (a near-pure tone's bandwidth is
            # a few Hz made of the f^2-weighted noise floor of a thousand empty bins; round 2 hard-coded 5e-4 of Nyquist
            # for such frames).  The float32 oracle must meet the same bound, or the bound is wrong.
            cen_tol = np.maximum(2e-5 * nyq, cen_cond)
            bw_tol = np.maximum(5e-5 * nyq, bw_cond)
            assert (np.abs(cen - cen64) <= cen_tol).all() and (np.abs(bw - bw64) <= bw_tol).all(), (sr, i, "oracle")
            assert (np.abs(g["centroid"] - cen64) <= cen_tol).all(), (sr, i, (np.abs(g["centroid"] - cen64) / cen_tol).max())
            assert (np.abs(g["bandwidth"] - bw64) <= bw_tol).all(), (sr, i, (np.abs(g["bandwidth"] - bw64) / bw_tol).max())
            # well-conditioned frames stay on the flat bound against the oracle, as before
            assert (np.abs(g["bandwidth"] - bw) <= 5e-5 * nyq)[bw > 1e-2 * nyq].all(), (sr, i)
            # roll-off: a running float32 sum decides a bin; numpy adds in bin order, the GPU per lane and then across
            # lanes -- a frame whose 85 % point falls within rounding of a bin boundary may land one bin off
            off_by = np.abs(g["rolloff"] - ro) / (sr / 2048)
            assert off_by.max() <= 1.001 and (off_by > 0.5).mean() <= 0.02, (sr, i, off_by.max())
            sc = max(float(pk.max()), 1e-30)
            assert np.abs(g["peak"] - pk).max() <= 2e-5 * sc, (sr, i)
            # valleys (mean of the smallest 2 % of a band's magnitudes) sit orders of magnitude below the peaks and the
            # contrast is a RATIO: each valley on its own scale (1e-3 relative) above the float32 FFT's noise floor
            # (~eps * sqrt(n_fft) of the frame's largest magnitude, 5e-6 of the clip maximum here)
            v_tol = 1e-3 * np.abs(vl) + 5e-6 * sc
            assert (np.abs(g["valley"] - vl) <= v_tol).all(), (sr, i, (np.abs(g["valley"] - vl) / v_tol).max())
            # and the contrast itself, in dB, wherever it is well conditioned (valley above -60 dB of the clip maximum)
            well = vl > 1e-3 * sc
            if well.any():
                con_g = 10.0 * np.log10(np.maximum(1e-10, g["peak"].astype(np.float64))) - 10.0 * np.log10(np.maximum(1e-10, g["valley"].astype(np.float64)))
                con_r = 10.0 * np.log10(np.maximum(1e-10, pk)) - 10.0 * np.log10(np.maximum(1e-10, vl))
                assert np.abs(con_g - con_r)[well].max() <= 1e-2, (sr, i, np.abs(con_g - con_r)[well].max())
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
