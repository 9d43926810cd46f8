"""CPU oracle for ``extract_f0`` (pYIN) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` and ``__graft_entry__.smoke()`` may import this module.

PARITY UNPINNED.  The reference calls ``librosa.pyin(y, fmin=C2, fmax=C7,
frame_length=self.frame_length, hop_length=self.hop_length, sr=self.sr)``
(``audio_feature_extraction_toolkit/core/feature_extractor.py:87-94``) and reduces the
result to four scalars (``:97-114``).  librosa (pinned 0.11.0,
``04_feature_extraction_experiment/requirements_old.txt:35``) is not installed and the
reference holds no vector for this stage, so this file restates the published algorithm
of ``librosa.core.pitch.pyin`` / ``librosa.sequence.viterbi`` at its defaults
(``n_thresholds=100, beta_parameters=(2, 18), boltzmann_parameter=2, resolution=0.1,
max_transition_rate=35.92, switch_prob=0.01, no_trough_prob=0.01, center=True,
pad_mode='constant'``) with the dtype flow of the pinned stack (numpy 1.24: ``np.fft``
evaluates in float64; ``np.cumsum`` of a float32 array stays float32).  It is pinned
only by the analytic known-answer tests in ``tests/test_pyin_oracle.py``.

Steps (names follow librosa's):
  1. centre-pad ``frame_length // 2`` zeros, frame with ``hop_length``;
  2. difference function d(tau) = e[0] + e[tau] - 2 acf[tau] from an FFT autocorrelation
     (float64) and a *float32* running energy, both zeroed below 1e-6; cumulative-mean
     normalisation over tau = 1..max_period, kept for tau = min_period..max_period;
  3. parabolic interpolation of the normalised function;
  4. per frame: troughs, for each of 100 thresholds a Boltzmann prior over the troughs
     below it, weighted by the Beta(2, 18) mass of the threshold; the global minimum
     also collects ``no_trough_prob`` x the mass of the thresholds it does not undercut;
  5. candidates -> pitch bins (10 per semitone from fmin), observation matrix with an
     unvoiced copy of every bin, Viterbi over 2 x n_pitch_bins states with a triangular
     +-(max_semitones_per_frame x 10) transition band and a 1 % voicing switch;
  6. f0 = fmin * 2**(bin / 120) on voiced frames, NaN elsewhere.
"""
from __future__ import annotations

import numpy as np
import scipy.signal
import scipy.stats

__all__ = ["pyin", "extract_f0", "yin_frames", "pyin_tables"]

C2_HZ = 65.40639132514966      # librosa.note_to_hz('C2')
C7_HZ = 2093.004522404789      # librosa.note_to_hz('C7')


def _frame(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    n = 1 + (y.shape[-1] - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n)[None, :]
    return y[idx]                                   # [frame_length, n_frames], as librosa.util.frame


def periods(sr: float, fmin: float, fmax: float, frame_length: int, win_length: int):
    min_period = int(np.floor(sr / fmax))
    max_period = min(int(np.ceil(sr / fmin)), frame_length - win_length - 1)
    return min_period, max_period


def yin_frames(y: np.ndarray, sr: float, fmin: float, fmax: float, frame_length: int, hop_length: int):
    """Cumulative-mean-normalised difference function, [max_period - min_period + 1, T] float64
    (librosa.core.pitch._cumulative_mean_normalized_difference)."""
    y = np.asarray(y, dtype=np.float32)
    win_length = frame_length // 2
    yp = np.pad(y, (frame_length // 2, frame_length // 2), mode="constant")
    y_frames = _frame(yp, frame_length, hop_length)                      # float32
    min_period, max_period = periods(sr, fmin, fmax, frame_length, win_length)
    # autocorrelation through the FFT: numpy 1.24 evaluates in float64
    yf = y_frames.astype(np.float64)
    a = np.fft.rfft(yf, frame_length, axis=0)
    b = np.fft.rfft(yf[win_length:0:-1, :], frame_length, axis=0)
    acf = np.fft.irfft(a * b, frame_length, axis=0)[win_length:, :]
    acf[np.abs(acf) < 1e-6] = 0
    # energy terms: cumsum of a float32 array is a sequential float32 accumulation
    energy = np.cumsum(y_frames ** 2, axis=0)
    energy = energy[win_length:, :] - energy[:-win_length, :]
    energy[np.abs(energy) < 1e-6] = 0
    yin = energy[:1, :] + energy - 2 * acf                               # float32 + float32, then float64
    num = yin[min_period:max_period + 1, :]
    tau = np.arange(1, max_period + 1)[:, None]
    cmean = np.cumsum(yin[1:max_period + 1, :], axis=0) / tau
    den = cmean[min_period - 1:max_period, :]
    return num / (den + np.finfo(den.dtype).tiny), min_period, max_period


def _parabolic_interpolation(x: np.ndarray) -> np.ndarray:
    """librosa.core.pitch._parabolic_interpolation along axis 0."""
    shifts = np.zeros_like(x)
    a = x[2:] + x[:-2] - 2 * x[1:-1]
    b = (x[2:] - x[:-2]) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(np.abs(b) >= np.abs(a), 0.0, -b / a)
    shifts[1:-1] = s
    return shifts


def _localmin(x: np.ndarray) -> np.ndarray:
    """librosa.util.localmin on a 1-D array: x[i] < x[i-1] and x[i] <= x[i+1]; last: x[-1] < x[-2]."""
    m = np.zeros(x.shape, dtype=bool)
    m[1:-1] = (x[1:-1] < x[:-2]) & (x[1:-1] <= x[2:])
    m[-1] = x[-1] < x[-2]
    return m


def pyin_tables(sr: float, fmin: float, fmax: float, hop_length: int, n_thresholds: int = 100,
                beta_parameters=(2, 18), resolution: float = 0.1, max_transition_rate: float = 35.92,
                switch_prob: float = 0.01):
    thresholds = np.linspace(0, 1, n_thresholds + 1)
    beta_cdf = scipy.stats.beta.cdf(thresholds, beta_parameters[0], beta_parameters[1])
    beta_probs = np.diff(beta_cdf)
    n_bins_per_semitone = int(np.ceil(1.0 / resolution))
    n_pitch_bins = int(np.floor(12 * n_bins_per_semitone * np.log2(fmax / fmin))) + 1
    max_semitones_per_frame = round(max_transition_rate * 12 * hop_length / sr)
    transition_width = max_semitones_per_frame * n_bins_per_semitone + 1
    # librosa.sequence.transition_local(n, width, window='triangle', wrap=False)
    n = n_pitch_bins
    local = np.zeros((n, n), dtype=np.float64)
    win = scipy.signal.get_window("triangle", transition_width, fftbins=False)
    for i in range(n):
        # pad the window to n states, centre it on i, no wrap-around
        trans_row = np.zeros(n)
        lo = i - transition_width // 2
        for k in range(transition_width):
            j = lo + k
            if 0 <= j < n:
                trans_row[j] = win[k]
        local[i] = trans_row / trans_row.sum()
    t_switch = np.array([[1 - switch_prob, switch_prob], [switch_prob, 1 - switch_prob]])
    transition = np.kron(t_switch, local)
    p_init = np.zeros(2 * n)
    p_init[n:] = 1 / n
    return dict(thresholds=thresholds, beta_probs=beta_probs, n_bins_per_semitone=n_bins_per_semitone,
                n_pitch_bins=n_pitch_bins, transition=transition, p_init=p_init,
                transition_width=transition_width, local=local)


def observation_probs(yin: np.ndarray, shifts: np.ndarray, sr: float, fmin: float, min_period: int, tb: dict,
                      boltzmann_parameter: float = 2.0, no_trough_prob: float = 0.01):
    """librosa.core.pitch.__pyin_helper."""
    thresholds, beta_probs = tb["thresholds"], tb["beta_probs"]
    n_pitch_bins, nbs = tb["n_pitch_bins"], tb["n_bins_per_semitone"]
    yin_probs = np.zeros_like(yin)
    for i, yin_frame in enumerate(yin.T):
        is_trough = _localmin(yin_frame)
        is_trough[0] = yin_frame[0] < yin_frame[1]
        (trough_index,) = np.nonzero(is_trough)
        if len(trough_index) == 0:
            continue
        trough_heights = yin_frame[trough_index]
        trough_thresholds = np.less.outer(trough_heights, thresholds[1:])
        trough_positions = np.cumsum(trough_thresholds, axis=0) - 1
        n_troughs = np.count_nonzero(trough_thresholds, axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            trough_prior = scipy.stats.boltzmann.pmf(trough_positions, boltzmann_parameter, n_troughs)
        trough_prior[~trough_thresholds] = 0
        probs = trough_prior.dot(beta_probs)
        global_min = np.argmin(trough_heights)
        n_thresholds_below_min = np.count_nonzero(~trough_thresholds[global_min, :])
        probs[global_min] += no_trough_prob * np.sum(beta_probs[:n_thresholds_below_min])
        yin_probs[trough_index, i] = probs
    yin_period, frame_index = np.nonzero(yin_probs)
    period_candidates = min_period + yin_period
    period_candidates = period_candidates + shifts[yin_period, frame_index]
    f0_candidates = sr / period_candidates
    bin_index = 12 * nbs * np.log2(f0_candidates / fmin)
    bin_index = np.clip(np.round(bin_index), 0, n_pitch_bins).astype(int)
    obs = np.zeros((2 * n_pitch_bins, yin.shape[1]))
    obs[bin_index, frame_index] = yin_probs[yin_period, frame_index]
    voiced_prob = np.clip(np.sum(obs[:n_pitch_bins, :], axis=0, keepdims=True), 0, 1)
    obs[n_pitch_bins:, :] = (1 - voiced_prob) / n_pitch_bins
    return obs, voiced_prob[0]


def viterbi(prob: np.ndarray, transition: np.ndarray, p_init: np.ndarray) -> np.ndarray:
    """librosa.sequence.viterbi (log domain, epsilon = tiny, first-index argmax)."""
    n_states, n_steps = prob.shape
    eps = np.finfo(prob.dtype).tiny
    log_trans = np.log(transition + eps)
    log_prob = np.log(prob.T + eps)
    log_p_init = np.log(p_init + eps)
    value = np.empty((n_steps, n_states))
    ptr = np.empty((n_steps, n_states), dtype=np.int64)
    value[0] = log_prob[0] + log_p_init
    lt_T = np.ascontiguousarray(log_trans.T)
    for t in range(1, n_steps):
        trans_out = value[t - 1] + lt_T                 # [j, k] = V[t-1, k] + log A[k, j]
        ptr[t] = np.argmax(trans_out, axis=1)
        value[t] = log_prob[t] + trans_out[np.arange(n_states), ptr[t]]
    states = np.empty(n_steps, dtype=np.int64)
    states[-1] = np.argmax(value[-1])
    for t in range(n_steps - 2, -1, -1):
        states[t] = ptr[t + 1, states[t + 1]]
    return states


def pyin(y: np.ndarray, fmin: float = C2_HZ, fmax: float = C7_HZ, sr: float = 22050,
         frame_length: int = 1024, hop_length: int = 256, return_internal: bool = False):
    """-> (f0 [T] float64 with NaN on unvoiced frames, voiced_flag [T] bool, voiced_prob [T])."""
    yin, min_period, max_period = yin_frames(y, sr, fmin, fmax, frame_length, hop_length)
    shifts = _parabolic_interpolation(yin)
    tb = pyin_tables(sr, fmin, fmax, hop_length)
    obs, voiced_prob = observation_probs(yin, shifts, sr, fmin, min_period, tb)
    states = viterbi(obs, tb["transition"], tb["p_init"])
    n = tb["n_pitch_bins"]
    freqs = fmin * 2 ** (np.arange(n) / (12 * tb["n_bins_per_semitone"]))
    f0 = freqs[states % n]
    voiced_flag = states < n
    f0 = np.where(voiced_flag, f0, np.nan)
    if return_internal:
        return f0, voiced_flag, voiced_prob, dict(yin=yin, shifts=shifts, obs=obs, states=states, tables=tb)
    return f0, voiced_flag, voiced_prob


def extract_f0(y: np.ndarray, sr: float = 22050, frame_length: int = 1024, hop_length: int = 256,
               fmin: float = C2_HZ, fmax: float = C7_HZ, return_frames: bool = False) -> dict:
    """feature_extractor.py:76-114."""
    f0, voiced_flag, _ = pyin(y, fmin, fmax, sr, frame_length, hop_length)
    f0_valid = f0[~np.isnan(f0)]
    if len(f0_valid) > 0:
        f0_mean = np.mean(f0_valid)
        f0_std = np.std(f0_valid)
        f0_missing_rate = np.sum(np.isnan(f0)) / len(f0)
        f0_quality = 1 - f0_missing_rate
    else:
        f0_mean, f0_std, f0_missing_rate, f0_quality = 0, 0, 1, 0
    out = {"f0_mean": float(f0_mean), "f0_std": float(f0_std),
           "f0_missing_rate": float(f0_missing_rate), "f0_quality": float(f0_quality)}
    if return_frames:
        out["f0"] = f0
        out["voiced_flag"] = voiced_flag
    return out
