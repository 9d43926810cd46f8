"""MI355X-native ``AudioFeatureExtractor`` -- drop-in for the reference class
(audio_feature_extraction_toolkit/core/feature_extractor.py:7-237): same
constructor arguments and attributes (:10-39), same methods, same result-dict
keys / key order / Python types (:109-114, :146-151, :175-178, :202-207) and the
same error behaviour (log + re-raise in ``load_audio`` / ``extract_features``;
``batch_process`` logs and skips a failing file, :233-235).

The MFCC / RMS arithmetic that the reference delegates to librosa runs in the
hand-written HIP kernels of ``libafx.so`` through the ctypes C-ABI in
``include/afx.h``.  There is no CPU fallback: without the library or a GPU the
calls raise.
"""
from __future__ import annotations

import logging
import threading
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _native, wavio

# librosa.note_to_hz('C2') / ('C7') -- the reference evaluates these at class-definition
# time (feature_extractor.py:15-16); written out so that importing needs no librosa.
_C2_HZ = 65.40639132514966
_C7_HZ = 2093.004522404789



def _status_error(status: int, what: str, n_frames: int = 0) -> Exception:
    if status == _native.CLIP_NONFINITE:
        return ValueError("Audio buffer is not finite everywhere")
    if status == _native.CLIP_TOO_SHORT:
        return ValueError(
            f"when mode='interp', width=9 cannot exceed data.shape[axis]={n_frames} ({what})")
    return RuntimeError(f"{what}: clip status {status}")


class AudioFeatureExtractor:
    """音頻特徵提取器類 (GPU)"""

    def __init__(self,
                 sr: int = 22050,
                 frame_length: int = 1024,
                 hop_length: int = 256,
                 n_mfcc: int = 13,
                 f0_min: float = _C2_HZ,
                 f0_max: float = _C7_HZ,
                 pre_emphasis: float = 0.97,
                 *,
                 window: str = "hamming",
                 n_mels: int = 128,
                 fmin: float = 0.0,
                 fmax: Optional[float] = None,
                 htk: bool = False,
                 lifter: float = 0.0,
                 device: Optional[Sequence[int] | int] = None):
        """Positional arguments are the reference's (feature_extractor.py:10-17).
        Keyword-only extensions default to the reference's hard-coded values:
        ``window`` ('hamming', :133; 'hann' also supported), ``n_mels`` (librosa's 128), ``fmin`` / ``fmax`` /
        ``htk`` (librosa.filters.mel's band edges and mel scale) and ``lifter`` (librosa.feature.mfcc's) -- the
        variants the reference's experiment extractors pass (04_feature_extraction_experiment/
        audio_feature_extraction 2/audio_feature_extraction/feature_extractor.py:148-181),
        ``device`` (GPU index or list of indices; None = all visible GPUs for
        ``batch_process``, GPU 0 for single-clip calls)."""
        self.sr = sr
        self.frame_length = frame_length
        self.hop_length = hop_length
        self.n_mfcc = n_mfcc
        self.f0_min = f0_min
        self.f0_max = f0_max
        self.pre_emphasis = pre_emphasis
        self.window = window
        self.n_mels = n_mels
        self.fmin, self.fmax, self.htk, self.lifter = fmin, fmax, htk, lifter
        self.device = device

        logging.basicConfig(level=logging.INFO)
        self.logger = logging.getLogger(__name__)
        self._plans: Dict[Tuple[int, int], _native.Plan] = {}
        self._plan_lock = threading.Lock()

    # ------------------------------------------------------------------ plumbing
    def _devices(self) -> List[int]:
        if self.device is None:
            n = _native.device_count()
            if n <= 0:
                raise _native.AfxError("no MI355X / HIP device visible (no CPU fallback exists)")
            return list(range(n))
        if isinstance(self.device, int):
            return [self.device]
        return [int(d) for d in self.device]

    def _plan(self, device: Optional[int] = None, lane: int = 0) -> _native.Plan:
        """One plan (own context + stream) per (device, lane); lanes > 0 are the extra in-flight
        workers ``batch_process`` runs per GPU."""
        if device is None:
            device = self._devices()[0]
        with self._plan_lock:
            pl = self._plans.get((device, lane))
            if pl is None:
                params = _native.make_params(self.sr, self.frame_length, self.hop_length, self.n_mfcc,
                                             self.n_mels, self.window, self.pre_emphasis,
                                             self.fmin, self.fmax, self.htk, self.lifter)
                pl = _native.Plan(_native.Context(device), params)
                self._plans[(device, lane)] = pl
            return pl

    def _uses_reference_stages(self) -> bool:
        """True when no stage method has been replaced on the instance or in a subclass
        (README.md:135-136 shows users monkey-patching ``preprocess_audio``); only then may
        ``extract_features`` / ``batch_process`` take the fused single-pass GPU path."""
        for name in ("preprocess_audio", "extract_mfcc", "extract_energy", "extract_f0", "load_audio"):
            if name in self.__dict__ or getattr(type(self), name) is not getattr(AudioFeatureExtractor, name):
                return False
        return True

    def _stats_to_dicts(self, stats: np.ndarray) -> Tuple[Dict[str, Any], Dict[str, Any]]:
        K = self.n_mfcc
        mfcc = {
            "mfcc_mean": stats[0:K].tolist(),
            "mfcc_std": stats[K:2 * K].tolist(),
            "mfcc_delta_mean": stats[2 * K:3 * K].tolist(),
            "mfcc_delta2_mean": stats[3 * K:4 * K].tolist(),
        }
        energy = {
            "energy_mean": float(stats[4 * K]),
            "energy_std": float(stats[4 * K + 1]),
            "energy_range": float(stats[4 * K + 2]),
        }
        return mfcc, energy

    def _run_one(self, y: np.ndarray, flags: int) -> np.ndarray:
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = self._plan().extract_batch(y, np.zeros(1, np.int64), np.array([y.size], np.int64), flags=flags)
        if out["status"][0] != _native.CLIP_OK:
            raise _status_error(int(out["status"][0]), "extract", int(out["nframes"][0]))
        return out["stats"][0]

    # ------------------------------------------------------------------ reference API
    def load_audio(self, audio_path: str) -> Tuple[np.ndarray, int]:
        """載入音頻文件 -> (float32 mono, sr).  WAV decode + channel mean as librosa.load
        (feature_extractor.py:52); files at another rate are resampled on the host."""
        try:
            y, sr = wavio.load(audio_path, self.sr)
            return y, sr
        except Exception as e:
            self.logger.error(f"載入音頻文件失敗: {str(e)}")
            raise

    def preprocess_audio(self, y: np.ndarray) -> np.ndarray:
        """音頻預處理: pre-emphasis (coef=self.pre_emphasis) then silence trim at 30 dB
        (feature_extractor.py:58-74), on the GPU."""
        y_pre, start, end, status = self._plan().preprocess(np.asarray(y, dtype=np.float32))
        if status != _native.CLIP_OK:
            raise _status_error(status, "preprocess_audio")
        return y_pre[start:end]

    @staticmethod
    def _f0_to_dict(s: np.ndarray) -> Dict[str, Any]:
        return {
            "f0_mean": float(s[0]),
            "f0_std": float(s[1]),
            "f0_missing_rate": float(s[2]),
            "f0_quality": float(s[3]),
        }

    def _f0_on_gpu(self) -> bool:
        """The batched path may compute F0 on the device only while ``extract_f0`` is the stock method."""
        return "extract_f0" not in self.__dict__ and type(self).extract_f0 is AudioFeatureExtractor.extract_f0

    def _run_f0(self, y: np.ndarray, flags: int) -> np.ndarray:
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = self._plan().f0_batch(y, np.zeros(1, np.int64), np.array([y.size], np.int64),
                                    float(self.f0_min), float(self.f0_max), flags=flags)
        if out["status"][0] != _native.CLIP_OK:
            raise _status_error(int(out["status"][0]), "extract_f0")
        return out["stats"][0]

    def extract_f0(self, y: np.ndarray) -> Dict[str, Any]:
        """提取基頻特徵 (feature_extractor.py:76-114): ``librosa.pyin(y, fmin=f0_min, fmax=f0_max,
        frame_length, hop_length, sr)`` on the GPU -- difference function, probabilistic thresholds,
        Viterbi -- reduced to mean / std over the voiced frames, missing rate and quality."""
        return self._f0_to_dict(self._run_f0(y, 0))

    def extract_mfcc(self, y: np.ndarray) -> Dict[str, Any]:
        """提取MFCC特徵 of an already preprocessed signal (feature_extractor.py:116-151)."""
        return self._stats_to_dicts(self._run_one(y, 0))[0]

    def extract_energy(self, y: np.ndarray) -> Dict[str, Any]:
        """提取能量特徵 of an already preprocessed signal (feature_extractor.py:153-179).  Only ``librosa.feature.rms``
        is involved there, so a clip with fewer than 9 frames -- which fails ``extract_mfcc`` on the width-9 delta --
        still has its energy statistics."""
        return self._energy_from(y, 0)

    def _energy_from(self, y: np.ndarray, flags: int) -> Dict[str, Any]:
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = self._plan().extract_batch(y, np.zeros(1, np.int64), np.array([y.size], np.int64), flags=flags)
        st = int(out["status"][0])
        # a TOO_SHORT clip with at least two samples has its RMS statistics on every shape (include/afx.h, out_stats)
        if st != _native.CLIP_OK and not (st == _native.CLIP_TOO_SHORT and y.size >= 2 and out["nframes"][0] >= 1):
            raise _status_error(st, "extract_energy", int(out["nframes"][0]))
        return self._stats_to_dicts(out["stats"][0])[1]

    def extract_features(self, audio_path: str, *, features_to_extract: Optional[Sequence[str]] = None) -> Dict[str, Any]:
        """提取所有特徵 (feature_extractor.py:181-213).

        ``features_to_extract`` (keyword-only; README.md:141-146 shows ``['f0', 'mfcc', 'energy']``): the feature groups
        wanted.  None = all three, the reference's behaviour; a subset leaves the other groups' keys out of the dict
        (key order unchanged) and skips their GPU passes -- without ``'f0'`` no pYIN pass runs."""
        try:
            from ..parallel import normalize_features
            want = normalize_features(features_to_extract)
            y, _ = self.load_audio(audio_path)
            f0_features: Dict[str, Any] = {}
            mfcc_features: Dict[str, Any] = {}
            energy_features: Dict[str, Any] = {}
            if self._uses_reference_stages():
                flags = _native.FLAG_PREEMPH | _native.FLAG_TRIM
                if "mfcc" in want:
                    # fused: pre-emphasis + trim + MFCC + RMS in one pass over the samples
                    mfcc_features, energy_features = self._stats_to_dicts(self._run_one(y, flags))
                elif "energy" in want:
                    energy_features = self._energy_from(y, flags)
                if "f0" in want:
                    f0_features = self._f0_to_dict(self._run_f0(y, flags))
            else:
                y_processed = self.preprocess_audio(y)
                if "f0" in want:
                    f0_features = self.extract_f0(y_processed)
                if "mfcc" in want:
                    mfcc_features = self.extract_mfcc(y_processed)
                if "energy" in want:
                    energy_features = self.extract_energy(y_processed)
            features = {"file_path": audio_path}
            if "f0" in want:
                features.update(f0_features)
            if "mfcc" in want:
                features.update(mfcc_features)
            if "energy" in want:
                features.update(energy_features)
            return features
        except Exception as e:
            self.logger.error(f"特徵提取失敗: {str(e)}")
            raise

    # ------------------------------------------------------------------ per-frame export
    def extract_frame_features(self, audio_path: str) -> Dict[str, np.ndarray]:
        """The per-frame matrices the reference computes and then reduces (feature_extractor.py:127-138,
        :164, :87), in the layout its experiment scripts save and its DTW aligner reads
        (04_feature_extraction_experiment/feature_extraction.py:191-215, 340-352): ``mfcc`` is
        ``vstack([mfcc, delta, delta2])`` of shape (3*n_mfcc, T) float32, ``f0`` the pYIN track (T,)
        float64 with NaN on unvoiced frames, ``energy`` the RMS row (T,) float32, ``zcr`` the
        zero-crossing rate (T,) float64 -- all of the preprocessed signal, all computed on the GPU."""
        y, _ = self.load_audio(audio_path)
        y = np.ascontiguousarray(y, dtype=np.float32)
        flags = _native.FLAG_PREEMPH | _native.FLAG_TRIM
        off, ln = np.zeros(1, np.int64), np.array([y.size], np.int64)
        plan = self._plan()
        out = plan.extract_batch(y, off, ln, flags=flags, want_frames=True)
        if out["status"][0] != _native.CLIP_OK:
            raise _status_error(int(out["status"][0]), "extract_frame_features", int(out["nframes"][0]))
        fr = out["frames"][0]
        T = int(out["nframes"][0])
        f0 = plan.f0_batch(y, off, ln, float(self.f0_min), float(self.f0_max), flags=flags, want_frames=True)
        zcr = plan.zcr_batch(y, off, ln, flags=flags)
        return {
            "mfcc": np.vstack([fr["mfcc"], fr["mfcc_delta"], fr["mfcc_delta2"]]),
            "f0": f0["f0_flat"][:T].copy(),
            "energy": fr["rms"][0].copy(),
            "zcr": zcr["zcr_flat"][:T].copy(),
        }

    # ------------------------------------------------------------------ sibling frame features
    def _spectral_plan(self) -> _native.Plan:
        """librosa's defaults for the spectral descriptors: n_fft 2048, hop 512, Hann -- a plan of its own."""
        device = self._devices()[0]
        with self._plan_lock:
            pl = self._plans.get((device, "spectral"))
            if pl is None:
                pl = _native.Plan(_native.Context(device), _native.make_params(self.sr, 2048, 512, 13, 128, "hann", self.pre_emphasis))
                self._plans[(device, "spectral")] = pl
            return pl

    def extract_spectral_frames(self, y: np.ndarray) -> Dict[str, np.ndarray]:
        """Frame-level ``librosa.feature.spectral_centroid / spectral_bandwidth / spectral_rolloff / spectral_contrast`` of
        a signal at librosa's defaults, as the reference's experiment extractor calls them
        (04_feature_extraction_experiment/feature_extractor.py:497-506): STFT, moments, roll-off search and the band
        extremes of the contrast on the GPU; the contrast's dB difference with its clip-global ``top_db`` on the host."""
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = self._spectral_plan().spectral_batch(y, np.zeros(1, np.int64), np.array([y.size], np.int64))
        if out["status"][0] != _native.CLIP_OK:
            raise _status_error(int(out["status"][0]), "extract_spectral_features")
        d = out["clips"][0]

        def power_to_db(S):            # librosa.power_to_db(ref=1.0, amin=1e-10, top_db=80.0) of the whole matrix
            L = 10.0 * np.log10(np.maximum(1e-10, S.astype(np.float64)))
            return np.maximum(L, L.max() - 80.0)
        return {"spectral_centroid": d["centroid"], "spectral_bandwidth": d["bandwidth"], "spectral_rolloff": d["rolloff"],
                "spectral_contrast": power_to_db(d["peak"]) - power_to_db(d["valley"])}

    def extract_spectral_features(self, y: np.ndarray) -> Dict[str, Any]:
        """提取頻譜特徵: the eight statistics of 04_feature_extraction_experiment/feature_extractor.py:509-518."""
        f = self.extract_spectral_frames(y)
        out: Dict[str, Any] = {}
        for k in ("spectral_centroid", "spectral_bandwidth", "spectral_rolloff", "spectral_contrast"):
            out[k + "_mean"] = float(np.mean(f[k]))
            out[k + "_std"] = float(np.std(f[k]))
        return out

    @staticmethod
    def save_frame_features(features: Dict[str, np.ndarray], npz_path: str) -> None:
        """``np.savez(npz_path, **features)`` -- the reference's on-disk schema for frame-level features."""
        np.savez(npz_path, **features)

    def batch_process(self, audio_dir: str, *, features_to_extract: Optional[Sequence[str]] = None) -> List[Dict[str, Any]]:
        """批量處理音頻文件 (feature_extractor.py:215-237): every ``*.wav`` directly inside
        ``audio_dir`` in glob order; a failing file is logged and left out.  Files are
        sharded over the visible GPUs (no inter-GPU traffic; see parallel.py).
        ``features_to_extract``: as for ``extract_features``."""
        from ..parallel import normalize_features, process_files
        normalize_features(features_to_extract)            # a bad request fails the call, not every file
        files = list(Path(audio_dir).glob("*.wav"))
        if not self._uses_reference_stages():
            results = []
            for audio_file in files:
                try:
                    results.append(self.extract_features(str(audio_file), features_to_extract=features_to_extract))
                    self.logger.info(f"成功處理文件: {audio_file.name}")
                except Exception as e:
                    self.logger.error(f"處理文件 {audio_file.name} 失敗: {str(e)}")
                    continue
            return results
        return process_files(self, files, features_to_extract=features_to_extract)
