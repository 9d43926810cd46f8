"""RIFF/WAVE decode for ``AudioFeatureExtractor.load_audio`` -- the decode+mono
half of ``librosa.load`` (reference: core/feature_extractor.py:52), restated
without soundfile: PCM 8/16/24/32-bit and IEEE float 32/64, any channel count,
scaled exactly as libsndfile scales to float32 (int16 / 32768, int24 / 2**23,
int32 / 2**31, (uint8 - 128) / 128), channels averaged in float32.

Resampling (librosa's ``soxr_hq`` when the file rate differs from ``sr``) is an
ingest step outside the GPU hot path; ``resample`` below uses scipy's polyphase
filter and is NOT bit-compatible with soxr."""
from __future__ import annotations

import struct
from math import gcd

import numpy as np

_FMT_PCM, _FMT_FLOAT, _FMT_EXT = 1, 3, 0xFFFE


class WavError(ValueError):
    pass


def _parse(buf: bytes):
    if len(buf) < 12 or buf[0:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise WavError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid = buf[pos:pos + 4]
        size = struct.unpack_from("<I", buf, pos + 4)[0]
        body = pos + 8
        if cid == b"fmt ":
            if size < 16:
                raise WavError("short fmt chunk")
            tag, ch, rate, _, align, bits = struct.unpack_from("<HHIIHH", buf, body)
            if tag == _FMT_EXT and size >= 40:
                tag = struct.unpack_from("<H", buf, body + 24)[0]
            fmt = (tag, ch, rate, align, bits)
        elif cid == b"data":
            data = (body, min(size, len(buf) - body))
            break
        pos = body + size + (size & 1)
    if fmt is None or data is None:
        raise WavError("missing fmt or data chunk")
    return fmt, data


def read_wav_raw(path: str):
    """Returns (array [frames, channels] in the file's sample type, rate, kind) with
    kind in {'s16', 'u8', 's24', 's32', 'f32', 'f64'}; 's24' comes back as int32."""
    with open(path, "rb") as f:
        buf = f.read()
    (tag, ch, rate, align, bits), (off, size) = _parse(buf)
    if ch < 1:
        raise WavError("no channels")
    bps = bits // 8
    if bps * ch == 0:
        raise WavError("bad sample size")
    n = size // (bps * ch)
    raw = memoryview(buf)[off: off + n * bps * ch]
    if tag == _FMT_PCM:
        if bits == 16:
            a, kind = np.frombuffer(raw, "<i2"), "s16"
        elif bits == 8:
            a, kind = np.frombuffer(raw, np.uint8), "u8"
        elif bits == 32:
            a, kind = np.frombuffer(raw, "<i4"), "s32"
        elif bits == 24:
            b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
            a = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
            a = np.where(a & 0x800000, a - 0x1000000, a).astype(np.int32)
            kind = "s24"
        else:
            raise WavError(f"unsupported PCM width {bits}")
    elif tag == _FMT_FLOAT:
        if bits == 32:
            a, kind = np.frombuffer(raw, "<f4"), "f32"
        elif bits == 64:
            a, kind = np.frombuffer(raw, "<f8"), "f64"
        else:
            raise WavError(f"unsupported float width {bits}")
    else:
        raise WavError(f"unsupported WAVE format tag {tag}")
    return a.reshape(n, ch), int(rate), kind


def to_float32(a: np.ndarray, kind: str) -> np.ndarray:
    if kind == "s16":
        return a.astype(np.float32) * np.float32(1.0 / 32768.0)
    if kind == "u8":
        return (a.astype(np.int32) - 128).astype(np.float32) * np.float32(1.0 / 128.0)
    if kind == "s24":
        return a.astype(np.float32) * np.float32(1.0 / 8388608.0)
    if kind == "s32":
        return a.astype(np.float32) * np.float32(1.0 / 2147483648.0)
    return a.astype(np.float32)


def to_mono(y: np.ndarray) -> np.ndarray:
    """[frames, channels] float32 -> [frames] float32 (librosa.to_mono = np.mean over channels)."""
    if y.shape[1] == 1:
        return np.ascontiguousarray(y[:, 0])
    return np.mean(y.T, axis=0, dtype=np.float32)


# Resampling policy (librosa.load(path, sr=...) at feature_extractor.py:52 resamples with
# soxr_hq).  soxr is not available here and bit parity with it is not a goal -- "parity unpinned" -- so this states
# what the engine's own resampler guarantees instead, and tests/test_wavio.py holds it to that:
#   * output length ceil(n * sr_out / sr_in), librosa's rule (it fixes the length after any resampler);
#   * a linear-phase Kaiser-windowed sinc low-pass (beta 12.98, ~125 dB) evaluated as a polyphase filter
#     (scipy.signal.resample_poly with explicit taps), transition band 0.913 .. 1.0 of the lower rate's Nyquist -- the
#     band edges soxr documents for its HQ recipe;
#   * pass band (up to 0.9 x that Nyquist): gain within 1e-4 of unity; stop band (at and above it): below -100 dB;
#   * float64 arithmetic, float32 result (librosa returns float32).
_RESAMPLE_TAPS = {}


def _resample_filter(up: int, down: int) -> np.ndarray:
    key = (up, down)
    h = _RESAMPLE_TAPS.get(key)
    if h is None:
        import scipy.signal
        fs = float(up)                                      # internal rate in units of sr_in
        nyq_low = 0.5 * min(1.0, up / down)                 # the lower of the two Nyquist frequencies, same units
        width = (1.0 - 0.913) * nyq_low
        numtaps, beta = scipy.signal.kaiserord(125.0, width / (0.5 * fs))
        numtaps |= 1
        h = scipy.signal.firwin(numtaps, 0.5 * (0.913 + 1.0) * nyq_low, window=("kaiser", beta), fs=fs)   # unity DC gain: resample_poly scales by up
        _RESAMPLE_TAPS[key] = h
    return h


def resample(y: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    """y at sr_in -> float32 at sr_out, ceil(n * sr_out / sr_in) samples (policy above)."""
    import scipy.signal
    if int(sr_in) == int(sr_out):
        return np.asarray(y, np.float32)
    g = gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    out = scipy.signal.resample_poly(np.asarray(y, np.float64), up, down, window=_resample_filter(up, down))
    n = int(np.ceil(y.shape[-1] * sr_out / sr_in))     # librosa fixes the length to ceil(n * ratio)
    out = out[:n] if out.shape[-1] >= n else np.pad(out, (0, n - out.shape[-1]))
    return out.astype(np.float32)


def load(path: str, sr: int | None):
    """librosa.load(path, sr=sr): float32 mono, resampled to sr when it differs."""
    a, rate, kind = read_wav_raw(path)
    y = to_mono(to_float32(a, kind))
    if sr is not None and rate != sr:
        y = resample(y, rate, sr)
        rate = sr
    return y, rate


def write_wav_pcm16(path: str, y: np.ndarray, sr: int):
    """Test/bench helper: mono PCM16 writer (round-to-nearest of y * 32768, clipped)."""
    q = np.clip(np.rint(np.asarray(y, np.float64) * 32768.0), -32768, 32767).astype("<i2")
    data = q.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, int(sr), int(sr) * 2, 2, 16) + b"data" + struct.pack("<I", len(data))
    with open(path, "wb") as f:
        f.write(hdr + data)
