"""Known-answer tests of the pYIN oracle (oracle/pyin_ref.py) -- librosa-independent facts."""
import numpy as np
import pytest

from oracle import pyin_ref as P

SR = 22050


def tone(freq, seconds=0.6, amp=0.3, sr=SR):
    t = np.arange(int(sr * seconds)) / sr
    return (amp * np.sin(2 * np.pi * freq * t)).astype(np.float32)


def test_periods_and_bins_at_reference_defaults():
    mn, mx = P.periods(SR, P.C2_HZ, P.C7_HZ, 1024, 512)
    assert (mn, mx) == (10, 338)
    tb = P.pyin_tables(SR, P.C2_HZ, P.C7_HZ, 256)
    assert tb["n_pitch_bins"] == 601 and tb["transition_width"] == 51
    assert tb["transition"].shape == (1202, 1202)
    np.testing.assert_allclose(tb["transition"].sum(axis=1), 1.0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(tb["beta_probs"].sum(), 1.0, rtol=0, atol=1e-12)
    # Beta(2, 18) CDF in closed form
    x = tb["thresholds"]
    np.testing.assert_allclose(np.cumsum(tb["beta_probs"]), (1 - (1 - x) ** 18 * (1 + 18 * x))[1:], atol=1e-13)


def test_difference_function_is_sum_of_squared_differences():
    rng = np.random.default_rng(3)
    y = (0.2 * rng.standard_normal(4096)).astype(np.float32)
    yin, mn, mx = P.yin_frames(y, SR, P.C2_HZ, P.C7_HZ, 1024, 256)
    # frame 8 lies fully inside the clip: rebuild d(tau) directly in float64
    t = 8
    fr = np.pad(y, (512, 512))[t * 256: t * 256 + 1024].astype(np.float64)
    d = np.array([np.sum((fr[1:513] - fr[1 + tau:513 + tau]) ** 2) for tau in range(mx + 1)])
    cm = np.cumsum(d[1:]) / np.arange(1, mx + 1)
    ref = d[mn:] / cm[mn - 1:]
    np.testing.assert_allclose(yin[:, t], ref, rtol=2e-4)        # float32 energy terms limit the agreement


@pytest.mark.parametrize("freq", [110.0, 220.0, 440.0, 880.0])
def test_pure_tone_is_tracked_on_the_semitone_grid(freq):
    f0, voiced, vp = P.pyin(tone(freq))
    mid = slice(3, -3)
    assert voiced[mid].all()
    # A2/A3/A4/A5 sit exactly on pitch bins (multiples of 10 bins above C2 + 9 semitones)
    np.testing.assert_allclose(f0[mid], freq, rtol=6e-3)
    assert abs(np.median(f0[mid]) - freq) / freq < 1e-3


def test_noise_and_silence_are_unvoiced():
    rng = np.random.default_rng(0)
    n = (0.1 * rng.standard_normal(SR // 2)).astype(np.float32)
    f0, voiced, _ = P.pyin(n)
    assert voiced.mean() < 0.1
    f0, voiced, vp = P.pyin(np.zeros(4000, np.float32))
    assert not voiced.any() and np.isnan(f0).all() and (vp == 0).all()
    s = P.extract_f0(np.zeros(4000, np.float32))
    assert s == {"f0_mean": 0.0, "f0_std": 0.0, "f0_missing_rate": 1.0, "f0_quality": 0.0}


def test_frame_count_matches_mfcc_framing():
    y = tone(200.0, 0.5)
    f0, _, _ = P.pyin(y)
    assert f0.shape[0] == 1 + y.size // 256


def test_viterbi_prefers_first_index_on_ties_and_follows_transitions():
    # two states, deterministic emissions: path must follow the emissions
    prob = np.array([[0.9, 0.1, 0.9], [0.1, 0.9, 0.1]])
    tr = np.array([[0.5, 0.5], [0.5, 0.5]])
    np.testing.assert_array_equal(P.viterbi(prob, tr, np.array([0.5, 0.5])), [0, 1, 0])
    # exact tie everywhere: numpy argmax picks the first index
    prob = np.full((2, 4), 0.5)
    np.testing.assert_array_equal(P.viterbi(prob, tr, np.array([0.5, 0.5])), [0, 0, 0, 0])
