"""CPU oracle for the MFCC / RMS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the timed CPU baseline.
Nothing under ``audio_feature_extraction_amd/`` imports it.

PARITY UNPINNED.  The reference's arithmetic for this path lives in third-party
``librosa`` (pinned ``librosa==0.11.0`` in
``04_feature_extraction_experiment/requirements_old.txt:35``; un-vendored, not
installed here, no network), and the reference's own test file for the hot-path
class (``tests/test_feature_extractor.py``) is empty: there are no golden
vectors to pin against.  This file restates, in numpy/scipy, the published
librosa 0.11.0 semantics at the reference's call sites in
``audio_feature_extraction_toolkit/core/feature_extractor.py`` (cited per
function below), calling the very scipy routines librosa calls underneath
(``scipy.signal.lfilter``, ``get_window``, ``savgol_filter``, ``scipy.fft.rfft``,
``scipy.fft.dct``).  It is pinned only by the librosa-independent analytic
known-answer tests in ``tests/test_oracle_kat.py`` and the committed fixtures in
``tests/golden/`` that this file generated (``oracle/make_golden.py``).

Two precisions:
  * ``np.float32`` -- mirrors the reference's dtype flow (librosa.load yields
    float32 and every stage keeps it; the STFT is evaluated in float64 because
    scipy's window is float64, then stored complex64 -- as librosa does).
  * ``np.float64`` -- same rounded constants, exact-ish arithmetic: the truth
    used to adjudicate float32-vs-float32 differences.
"""
from __future__ import annotations

import numpy as np
import scipy.fft
import scipy.signal

__all__ = [
    "preemphasis", "rms", "trim", "stft", "mel_filterbank", "melspectrogram",
    "power_to_db", "mfcc", "delta", "preprocess_audio", "extract_mfcc",
    "extract_energy", "extract_stats", "frame_count", "zero_crossing_rate",
]


# --------------------------------------------------------------------------
# librosa.util.frame (centered callers pad first)
# --------------------------------------------------------------------------
def _frame(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    """librosa.util.frame(axis=-1): view of shape (frame_length, n_frames)."""
    if y.shape[-1] < frame_length:
        raise ValueError(
            f"Input is too short (n={y.shape[-1]}) for frame_length={frame_length}")
    n_frames = 1 + (y.shape[-1] - frame_length) // hop_length
    s = y.strides[-1]
    return np.lib.stride_tricks.as_strided(
        y, shape=(frame_length, n_frames), strides=(s, hop_length * s), writeable=False)


def frame_count(n_samples: int, hop_length: int) -> int:
    """T = 1 + floor(N'/hop) for center=True framing."""
    return 1 + n_samples // hop_length


# --------------------------------------------------------------------------
# A2  preprocess_audio -> librosa.effects.preemphasis   (feature_extractor.py:69)
# --------------------------------------------------------------------------
def preemphasis(y: np.ndarray, coef: float = 0.97) -> np.ndarray:
    """out[n] = y[n] - coef*y[n-1] via scipy.signal.lfilter with librosa's
    default initial state zi = 2*y[0] - y[1]  (so out[0] = 3*y[0] - y[1]).
    b, a and zi take y.dtype.  In float64 mode the coefficient is still the
    float32-rounded one, so only arithmetic rounding differs from float32 mode."""
    if y.shape[-1] < 2:
        raise ValueError("preemphasis needs at least 2 samples")
    c32 = np.float32(-coef) if y.dtype == np.float64 else -coef
    b = np.asarray([1.0, c32], dtype=y.dtype)
    a = np.asarray([1.0], dtype=y.dtype)
    zi = 2 * y[..., 0:1] - y[..., 1:2]
    zi = np.atleast_1d(zi)
    y_out, _ = scipy.signal.lfilter(b, a, y, zi=np.asarray(zi, dtype=y.dtype))
    return y_out


# --------------------------------------------------------------------------
# A10 extract_energy -> librosa.feature.rms            (feature_extractor.py:164)
# --------------------------------------------------------------------------
def rms(y: np.ndarray, frame_length: int = 2048, hop_length: int = 512) -> np.ndarray:
    """center=True, pad_mode='constant' (zeros); sqrt(mean(x**2)) per frame.
    Returns shape (1, T) like librosa."""
    if not np.isfinite(y).all():
        raise ValueError("Audio buffer is not finite everywhere")
    pad = int(frame_length // 2)
    yp = np.pad(y, (pad, pad), mode="constant")
    x = _frame(yp, frame_length, hop_length)
    power = np.mean(np.square(x, dtype=y.dtype), axis=-2, keepdims=True)
    return np.sqrt(power)


# --------------------------------------------------------------------------
# A3  preprocess_audio -> librosa.effects.trim(top_db=30) (feature_extractor.py:72)
# --------------------------------------------------------------------------
def _power_to_db_noclamp(S, ref_value, amin):
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec = log_spec - 10.0 * np.log10(np.maximum(amin, ref_value))
    return log_spec


def trim(y: np.ndarray, top_db: float = 30.0, frame_length: int = 2048,
         hop_length: int = 512):
    """librosa.effects.trim with ref=np.max: frames whose RMS (2048/512,
    centered) is within top_db of the loudest frame are non-silent; keep
    [first*hop, min(N, (last+1)*hop)).  Returns (y[start:end], (start, end))."""
    mse = rms(y, frame_length=frame_length, hop_length=hop_length)[0]
    # amplitude_to_db(mse, ref=np.max, amin=1e-5, top_db=None)
    magnitude = np.abs(mse)
    ref_value = np.max(magnitude)
    power = np.square(magnitude)
    db = _power_to_db_noclamp(power, ref_value ** 2, 1e-5 ** 2)
    non_silent = db > -top_db
    nonzero = np.flatnonzero(non_silent)
    if nonzero.size > 0:
        start = int(nonzero[0]) * hop_length
        end = min(y.shape[-1], (int(nonzero[-1]) + 1) * hop_length)
    else:
        start, end = 0, 0
    return y[start:end], (start, end)


def preprocess_audio(y: np.ndarray, coef: float = 0.97, top_db: float = 30.0):
    """AudioFeatureExtractor.preprocess_audio (feature_extractor.py:58-74)."""
    y_pre = preemphasis(y, coef=coef)
    y_trim, idx = trim(y_pre, top_db=top_db)
    return y_trim, idx


# --------------------------------------------------------------------------
# A4  librosa.stft (center=True, pad zeros, periodic window, rfft)
# --------------------------------------------------------------------------
def get_window(window: str, n_fft: int) -> np.ndarray:
    return scipy.signal.get_window(window, n_fft, fftbins=True)  # float64


def stft(y: np.ndarray, n_fft: int, hop_length: int, window: str = "hamming") -> np.ndarray:
    """Returns complex64 (float32 in) / complex128 (float64 in), shape (B, T).
    The float64 window times float32 frames gives a float64 product, so the
    transform itself runs in double and is rounded on store -- librosa's flow."""
    if not np.isfinite(y).all():
        raise ValueError("Audio buffer is not finite everywhere")
    win = get_window(window, n_fft).reshape(-1, 1)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    frames = _frame(yp, n_fft, hop_length)
    out_dtype = np.complex64 if y.dtype == np.float32 else np.complex128
    D = np.empty((1 + n_fft // 2, frames.shape[-1]), dtype=out_dtype, order="F")
    n_cols = max(1, (2 ** 8 * 2 ** 10) // (D.shape[0] * D.itemsize))
    for s in range(0, frames.shape[-1], n_cols):
        t = min(s + n_cols, frames.shape[-1])
        D[:, s:t] = scipy.fft.rfft(win * frames[:, s:t], axis=-2)
    return D


# --------------------------------------------------------------------------
# A5  librosa.filters.mel (Slaney scale, Slaney area norm, float32 storage)
# --------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        m = f >= min_log_hz
        mels[m] = min_log_mel + np.log(f[m] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz(mels):
    mels = np.asanyarray(mels, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        m = mels >= min_log_mel
        freqs[m] = min_log_hz * np.exp(logstep * (mels[m] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_frequencies(n_mels: int, fmin: float, fmax: float, htk: bool = False) -> np.ndarray:
    """librosa.mel_frequencies: n_mels points evenly spaced in mel between fmin and fmax; htk=True uses the HTK
    formula mel = 2595 log10(1 + f / 700) instead of Slaney's (the option the reference's older extractor passes,
    04_feature_extraction_experiment/audio_feature_extraction 2/audio_feature_extraction/feature_extractor.py:148-155)."""
    if htk:
        lo = 2595.0 * np.log10(1.0 + np.float64(fmin) / 700.0)
        hi = 2595.0 * np.log10(1.0 + np.float64(fmax) / 700.0)
        return 700.0 * (10.0 ** (np.linspace(lo, hi, n_mels) / 2595.0) - 1.0)
    mels = np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels)
    return _mel_to_hz(mels)


def mel_filterbank(sr: float, n_fft: int, n_mels: int = 128, fmin: float = 0.0,
                   fmax: float | None = None, htk: bool = False) -> np.ndarray:
    if fmax is None or fmax <= 0:
        fmax = float(sr) / 2
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax, htk=htk)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def melspectrogram(y, sr, n_fft, hop_length, window="hamming", n_mels=128, fmin=0.0, fmax=None, htk=False):
    S = np.abs(stft(y, n_fft, hop_length, window)) ** 2.0
    basis = mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk)
    if y.dtype == np.float64:
        basis = basis.astype(np.float64)
    return np.einsum("ft,mf->mt", S, basis, optimize=True)


# --------------------------------------------------------------------------
# A6  librosa.power_to_db(ref=1.0, amin=1e-10, top_db=80) -- clip-global clamp
# --------------------------------------------------------------------------
def power_to_db(S, amin: float = 1e-10, top_db: float | None = 80.0):
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec -= 10.0 * np.log10(np.maximum(amin, 1.0))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


# --------------------------------------------------------------------------
# A7  librosa.feature.mfcc -> scipy.fft.dct(type=2, norm='ortho')[:n_mfcc]
# --------------------------------------------------------------------------
def mfcc(y, sr, n_mfcc, n_fft, hop_length, window="hamming", n_mels=128,
         top_db: float | None = 80.0, return_logmel: bool = False, fmin=0.0, fmax=None, htk=False, lifter=0.0):
    L = power_to_db(melspectrogram(y, sr, n_fft, hop_length, window, n_mels, fmin, fmax, htk), top_db=top_db)
    M = scipy.fft.dct(L, axis=-2, type=2, norm="ortho")[:n_mfcc, :]
    if lifter > 0:          # librosa.feature.mfcc(lifter=...): M *= 1 + (lifter / 2) * sin(pi * n / lifter), n = 1..n_mfcc
        LI = np.sin(np.pi * np.arange(1, 1 + n_mfcc, dtype=M.dtype) / lifter)
        M = M * (1 + (lifter / 2) * LI).astype(M.dtype)[:, np.newaxis]
    return (M, L) if return_logmel else M


# --------------------------------------------------------------------------
# A8  librosa.feature.delta -> savgol_filter(width 9, polyorder=order, mode='interp')
# --------------------------------------------------------------------------
def delta(data, width: int = 9, order: int = 1):
    data = np.atleast_1d(data)
    if width > data.shape[-1]:
        raise ValueError(
            f"when mode='interp', width={width} cannot exceed data.shape[axis]={data.shape[-1]}")
    return scipy.signal.savgol_filter(data, width, deriv=order, polyorder=order,
                                      axis=-1, mode="interp")


# --------------------------------------------------------------------------
# A9/A10 per-method statistics                          (feature_extractor.py:116-179)
# --------------------------------------------------------------------------
def extract_mfcc(y, sr, n_mfcc, n_fft, hop_length, window="hamming", n_mels=128,
                 return_frames: bool = False, fmin=0.0, fmax=None, htk=False, lifter=0.0):
    M = mfcc(y, sr, n_mfcc, n_fft, hop_length, window, n_mels, fmin=fmin, fmax=fmax, htk=htk, lifter=lifter)
    d1 = delta(M)
    d2 = delta(M, order=2)
    out = {
        "mfcc_mean": np.mean(M, axis=1),
        "mfcc_std": np.std(M, axis=1),
        "mfcc_delta_mean": np.mean(d1, axis=1),
        "mfcc_delta2_mean": np.mean(d2, axis=1),
    }
    if return_frames:
        out.update(mfcc=M, mfcc_delta=d1, mfcc_delta2=d2)
    return out


def extract_energy(y, frame_length, hop_length, return_frames: bool = False):
    r = rms(y, frame_length=frame_length, hop_length=hop_length)
    out = {
        "energy_mean": np.mean(r),
        "energy_std": np.std(r),
        "energy_range": np.ptp(r),
    }
    if return_frames:
        out["rms"] = r
    return out


def zero_crossing_rate(y: np.ndarray, frame_length: int = 2048, hop_length: int = 512) -> np.ndarray:
    """librosa.feature.zero_crossing_rate(y, frame_length, hop_length, center=True) -> [T] float64
    (the per-frame feature the reference's experiment scripts store next to mfcc / f0 / energy,
    04_feature_extraction_experiment/feature_extraction.py:340-352).  Centre padding is ``mode='edge'``;
    ``librosa.zero_crossings(threshold=1e-10, zero_pos=True, pad=False)``: samples with |y| <= 1e-10 count as +0,
    a crossing is a change of ``np.signbit`` between neighbours, the first sample of a frame never is one."""
    y = np.asarray(y)
    yp = np.pad(y, (frame_length // 2, frame_length // 2), mode="edge")
    fr = _frame(yp, frame_length, hop_length).copy()
    fr[np.abs(fr) <= 1e-10] = 0
    sign = np.signbit(fr)
    z = np.zeros(fr.shape, dtype=bool)
    z[1:, :] = sign[1:, :] != sign[:-1, :]
    return np.mean(z, axis=0)


# --------------------------------------------------------------------------
# Sibling frame-level features: librosa.feature.spectral_centroid / spectral_bandwidth /
# spectral_rolloff / spectral_contrast at librosa's defaults, as the reference's experiment extractor calls them
# (04_feature_extraction_experiment/feature_extractor.py:497-506): n_fft=2048, hop_length=512, window='hann',
# center=True (zero padding), magnitude spectrogram.
# --------------------------------------------------------------------------
def _magnitude_spectrogram(y, n_fft=2048, hop_length=512, window="hann"):
    return np.abs(stft(y, n_fft, hop_length, window))


def _normalize_l1(S):
    """librosa.util.normalize(S, norm=1, axis=-2): columns whose norm is below tiny keep their scale."""
    length = np.sum(np.abs(S), axis=-2, keepdims=True)
    tiny = np.finfo(S.dtype).tiny
    length = np.where(length < tiny, 1.0, length).astype(S.dtype)
    return S / length


def spectral_centroid(y, sr, n_fft=2048, hop_length=512, window="hann"):
    S = _magnitude_spectrogram(y, n_fft, hop_length, window)
    freq = np.fft.rfftfreq(n_fft, 1.0 / sr)[:, None]
    return np.sum(freq * _normalize_l1(S), axis=-2, keepdims=True)


def spectral_bandwidth(y, sr, n_fft=2048, hop_length=512, window="hann", p=2):
    S = _magnitude_spectrogram(y, n_fft, hop_length, window)
    freq = np.fft.rfftfreq(n_fft, 1.0 / sr)[:, None]
    cen = np.sum(freq * _normalize_l1(S), axis=-2, keepdims=True)
    dev = np.abs(freq - cen)
    return np.sum(_normalize_l1(S) * dev ** p, axis=-2, keepdims=True) ** (1.0 / p)


def spectral_rolloff(y, sr, n_fft=2048, hop_length=512, window="hann", roll_percent=0.85):
    S = _magnitude_spectrogram(y, n_fft, hop_length, window)
    freq = np.fft.rfftfreq(n_fft, 1.0 / sr)[:, None]
    total = np.cumsum(S, axis=-2)
    threshold = roll_percent * total[-1]
    ind = np.where(total < threshold, np.nan, 1)
    return np.nanmin(ind * freq, axis=-2, keepdims=True)


def spectral_contrast_parts(y, sr, n_fft=2048, hop_length=512, window="hann", fmin=200.0, n_bands=6, quantile=0.02):
    """(peak, valley), each (n_bands + 1, T): the band extremes librosa.feature.spectral_contrast averages."""
    S = _magnitude_spectrogram(y, n_fft, hop_length, window)
    freq = np.fft.rfftfreq(n_fft, 1.0 / sr)
    octa = np.zeros(n_bands + 2)
    octa[1:] = fmin * (2.0 ** np.arange(0, n_bands + 1))
    if np.any(octa[:-1] >= 0.5 * sr):
        raise ValueError("Frequency band exceeds Nyquist. Reduce either fmin or n_bands.")
    valley = np.zeros((n_bands + 1, S.shape[1]))
    peak = np.zeros_like(valley)
    for k, (f_low, f_high) in enumerate(zip(octa[:-1], octa[1:])):
        current_band = np.logical_and(freq >= f_low, freq <= f_high)
        idx = np.flatnonzero(current_band)
        if k > 0:
            current_band[idx[0] - 1] = True
        if k == n_bands:
            current_band[idx[-1] + 1:] = True
        sub_band = S[current_band]
        if k < n_bands:
            sub_band = sub_band[:-1]
        idx = int(np.maximum(np.rint(quantile * np.sum(current_band)), 1))
        sortedr = np.sort(sub_band, axis=-2)
        valley[k] = np.mean(sortedr[:idx], axis=-2)
        peak[k] = np.mean(sortedr[-idx:], axis=-2)
    return peak, valley


def spectral_contrast(y, sr, **kw):
    peak, valley = spectral_contrast_parts(y, sr, **kw)
    return power_to_db(peak) - power_to_db(valley)


def extract_spectral_features(y, sr):
    """The dict of 04_feature_extraction_experiment/feature_extractor.py:509-518."""
    c, b, r = spectral_centroid(y, sr)[0], spectral_bandwidth(y, sr)[0], spectral_rolloff(y, sr)[0]
    con = spectral_contrast(y, sr)
    return {"spectral_centroid_mean": np.mean(c), "spectral_centroid_std": np.std(c),
            "spectral_bandwidth_mean": np.mean(b), "spectral_bandwidth_std": np.std(b),
            "spectral_rolloff_mean": np.mean(r), "spectral_rolloff_std": np.std(r),
            "spectral_contrast_mean": np.mean(con), "spectral_contrast_std": np.std(con)}


def extract_stats(y_raw, sr=22050, frame_length=1024, hop_length=256, n_mfcc=13,
                  pre_emphasis=0.97, window="hamming", n_mels=128,
                  dtype=np.float32, return_frames: bool = False, fmin=0.0, fmax=None, htk=False, lifter=0.0):
    """preprocess_audio -> extract_mfcc + extract_energy, as extract_features
    does (feature_extractor.py:193-199) minus file load and pYIN.  Returns a
    dict of numpy values (statistics; per-frame arrays when return_frames)."""
    y = np.asarray(y_raw, dtype=np.float32).astype(dtype)
    y_proc, (start, end) = preprocess_audio(y, coef=pre_emphasis)
    out = {"trim": (start, end)}
    out.update(extract_mfcc(y_proc, sr, n_mfcc, frame_length, hop_length, window,
                            n_mels, return_frames=return_frames, fmin=fmin, fmax=fmax, htk=htk, lifter=lifter))
    out.update(extract_energy(y_proc, frame_length, hop_length, return_frames=return_frames))
    if return_frames:
        out["y_processed"] = y_proc
    return out
