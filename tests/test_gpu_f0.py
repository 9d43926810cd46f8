"""extract_f0 on the GPU (afx_f0_batch) against the pYIN oracle, through the C-ABI."""
import numpy as np
import pytest

from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_clip
from oracle import cpu_ref as R
from oracle import pyin_ref as P

pytestmark = pytest.mark.gpu

SR = 22050


@pytest.fixture(scope="module")
def plan():
    ctx = N.Context(0)
    pl = N.Plan(ctx, N.make_params(SR, 1024, 256, 13))
    yield pl
    pl.close()
    ctx.close()


def run(plan, clips, flags):
    lengths = np.array([c.size for c in clips], np.int64)
    pad = (lengths + 3) // 4 * 4
    offsets = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.int64)
    buf = np.zeros(int(pad.sum()), np.float32)
    for c, o in zip(clips, offsets):
        buf[o:o + c.size] = c
    out = plan.f0_batch(buf, offsets, lengths, P.C2_HZ, P.C7_HZ, flags=flags, want_frames=True)
    f0 = [out["f0_flat"][o:o + 1 + n // 256] for o, n in zip(out["f0_offsets"], lengths)]
    return out, f0


def voiced_tone(freq, seconds, vib=0.0, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(int(SR * seconds)) / SR
    f = freq * (1 + vib * np.sin(2 * np.pi * 5 * t))
    ph = 2 * np.pi * np.cumsum(f) / SR
    y = 0.3 * np.sin(ph) + 0.1 * np.sin(2 * ph + 0.3) + 0.05 * np.sin(3 * ph + 1.0)
    y += 0.005 * rng.standard_normal(t.size)
    return y.astype(np.float32)


NON_IDENTICAL = {}      # tag -> (frames that differ from the oracle's track, frames): printed by the last test of the module


def check_clip(f0_gpu, stats_gpu, y_processed, tag, max_diff=0):
    """max_diff: frames allowed to differ from the oracle's decoded track.  0 for single-source clips (the decoded
    state sequence is the oracle's, frame for frame); the random-mixture test passes its own bound (DESIGN.md 7:
    the 0-versus-2e-19 unvoiced observation of a strongly voiced frame depends on BLAS's summation order)."""
    f0_ref, voiced_ref, _ = P.pyin(y_processed, sr=SR, frame_length=1024, hop_length=256)
    assert f0_gpu.shape == f0_ref.shape, tag
    same = (np.isnan(f0_gpu) == np.isnan(f0_ref))
    v = ~np.isnan(f0_ref) & ~np.isnan(f0_gpu)
    same[v] &= np.abs(f0_gpu[v] - f0_ref[v]) <= 1e-9 * f0_ref[v]
    n_diff = int((~same).sum())
    NON_IDENTICAL[tag] = (n_diff, int(same.size))
    print(f"[f0] {tag}: {n_diff} of {same.size} frames differ from the oracle's track")
    assert n_diff <= max_diff, (tag, n_diff, same.size, np.flatnonzero(~same)[:10])
    ref = P.extract_f0(y_processed, sr=SR, frame_length=1024, hop_length=256)
    if same.all():
        np.testing.assert_allclose(stats_gpu, [ref["f0_mean"], ref["f0_std"], ref["f0_missing_rate"], ref["f0_quality"]],
                                   rtol=1e-10, atol=1e-12, err_msg=tag)
    else:
        assert abs(stats_gpu[0] - ref["f0_mean"]) <= 5e-3 * max(ref["f0_mean"], 1.0), tag
        assert abs(stats_gpu[2] - ref["f0_missing_rate"]) <= 0.02, tag


def test_f0_of_preprocessed_signals_matches_oracle(plan):
    clips = [voiced_tone(220.0, 0.8), voiced_tone(147.0, 1.1, vib=0.02, seed=1), voiced_tone(523.25, 0.5, vib=0.01, seed=2),
             make_clip(3, SR, 0.7), np.zeros(3000, np.float32)]
    out, f0 = run(plan, clips, flags=0)
    assert out["status"].tolist() == [0] * len(clips)
    for i, c in enumerate(clips):
        check_clip(f0[i], out["stats"][i], c, f"staged{i}")
    assert out["stats"][4].tolist() == [0.0, 0.0, 1.0, 0.0]


def test_f0_fused_with_preemphasis_and_trim(plan):
    sil = np.zeros(int(0.25 * SR), np.float32)
    clips = [np.concatenate([sil, voiced_tone(196.0, 0.9, vib=0.015, seed=4), sil]),
             make_clip(12, SR, 1.0, speechy=True)]
    out, f0 = run(plan, clips, flags=N.FLAG_PREEMPH | N.FLAG_TRIM)
    for i, c in enumerate(clips):
        yp, _ = R.preprocess_audio(c)
        check_clip(f0[i][:1 + yp.size // 256], out["stats"][i], yp, f"fused{i}")


def test_f0_range_ends_noise_floor_and_pitch_jumps(plan):
    """The corners of the Viterbi pass: pitches next to fmin / fmax (the edge-class transition rows, shared out over the
    waves of k_f0_viterbi), near-silence (the path sits in the first bins) and jumps far outside the transition band
    (out-of-band moves: k_f0_backtrack leaves its prefetched window and reads the column directly).  Several window
    lengths of frames, so that the ring is primed, wrapped and drained."""
    rng = np.random.default_rng(11)
    t = np.arange(int(0.9 * SR)) / SR
    clips = [(0.4 * np.sin(2 * np.pi * f * t) + 0.002 * rng.standard_normal(t.size)).astype(np.float32)
             for f in (66.0, 70.0, 2000.0, 2085.0)]
    clips.append((1e-4 * rng.standard_normal(t.size)).astype(np.float32))
    for fa, fb, seg in ((100.0, 1500.0, 0.1), (70.0, 2000.0, 0.05), (90.0, 700.0, 0.03)):
        fi = np.where((np.floor(t / seg).astype(int) & 1) == 0, fa, fb)
        clips.append((0.4 * np.sin(2 * np.pi * np.cumsum(fi) / SR) + 0.002 * rng.standard_normal(t.size)).astype(np.float32))
    clips += [voiced_tone(330.0, d, seed=7) for d in (0.012, 0.03, 0.06, 0.075)]       # 2 .. 7 frames: shorter than the ring
    out, f0 = run(plan, clips, flags=0)
    assert (out["status"] == 0).all()
    for i, c in enumerate(clips):
        # a frame that straddles a jump holds two sources: the documented fragile case of the mixtures --
        # 2 of 78 frames of the 70 / 2000 Hz clip differ from the oracle's track, with this round's kernels and with
        # the previous round's alike (same frames, profiles/r03_ab_runs.txt); everything else is identical
        jump = 5 <= i <= 7
        check_clip(f0[i], out["stats"][i], c, f"mixture-jump{i}" if jump else f"corner{i}", max_diff=3 if jump else 0)


def test_f0_flags_nonfinite_and_short_clips(plan):
    bad = voiced_tone(200.0, 0.3).copy()
    bad[100] = np.inf
    clips = [bad, voiced_tone(300.0, 0.02), np.array([0.1], np.float32)]
    out, f0 = run(plan, clips, flags=0)
    assert out["status"].tolist() == [N.CLIP_NONFINITE, 0, 0]
    check_clip(f0[1], out["stats"][1], clips[1], "short")
    check_clip(f0[2], out["stats"][2], clips[2], "one-sample")


@pytest.mark.parametrize("sr,n_fft,hop", [(16000, 512, 128), (44100, 2048, 512)])
def test_f0_other_frame_sizes(sr, n_fft, hop):
    """BASELINE configs 3 and 5 shapes: different lag counts, pitch-range clipping (max_period = n_fft/2 - 1)."""
    ctx = N.Context(0)
    pl = N.Plan(ctx, N.make_params(sr, n_fft, hop, 13))
    try:
        t = np.arange(int(sr * 0.6)) / sr
        rng = np.random.default_rng(5)
        clips = []
        for f in (130.0, 310.0):
            ph = 2 * np.pi * f * t * (1 + 0.01 * np.sin(2 * np.pi * 4 * t))
            clips.append((0.3 * np.sin(ph) + 0.08 * np.sin(2 * ph) + 0.004 * rng.standard_normal(t.size)).astype(np.float32))
        lengths = np.array([c.size for c in clips], np.int64)
        pad = (lengths + 3) // 4 * 4
        offsets = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.int64)
        buf = np.zeros(int(pad.sum()), np.float32)
        for c, o in zip(clips, offsets):
            buf[o:o + c.size] = c
        out = pl.f0_batch(buf, offsets, lengths, P.C2_HZ, P.C7_HZ, flags=0, want_frames=True)
        for i, c in enumerate(clips):
            T = 1 + c.size // hop
            f0 = out["f0_flat"][out["f0_offsets"][i]: out["f0_offsets"][i] + T]
            ref, _, _ = P.pyin(c, sr=sr, frame_length=n_fft, hop_length=hop)
            same = np.isnan(f0) == np.isnan(ref)
            v = ~np.isnan(f0) & ~np.isnan(ref)
            same[v] &= np.abs(f0[v] - ref[v]) <= 1e-9 * ref[v]
            NON_IDENTICAL[f"{sr}/{n_fft} clip{i}"] = (int((~same).sum()), int(same.size))
            print(f"[f0] {sr}/{n_fft} clip{i}: {int((~same).sum())} of {same.size} frames differ from the oracle's track")
            assert same.all(), (sr, i, same.mean(), np.flatnonzero(~same)[:10])
            r = P.extract_f0(c, sr=sr, frame_length=n_fft, hop_length=hop)
            np.testing.assert_allclose(out["stats"][i], [r["f0_mean"], r["f0_std"], r["f0_missing_rate"], r["f0_quality"]],
                                       rtol=1e-10, atol=1e-12)
    finally:
        pl.close()
        ctx.close()


@pytest.mark.parametrize("fmin,fmax,tones", [(100.0, 400.0, (150.0, 260.0)),      # 241 pitch bins
                                             (200.0, 300.0, (220.0, 262.0)),      # 71 bins: every target is within 2 band of a range end
                                             (50.0, 5000.0, (130.0, 310.0))])     # 798 bins: more than one target per Viterbi thread
def test_f0_other_pitch_ranges(plan, fmin, fmax, tones):
    """f0_min / f0_max other than the reference's C2..C7: other state counts for the Viterbi kernel (its target-to-wave
    mapping, the edge-class sources, several targets per thread) and other lag ranges for the yin kernel."""
    t = np.arange(int(SR * 0.8)) / SR
    rng = np.random.default_rng(11)
    clips = []
    for f in tones:
        ph = 2 * np.pi * f * t * (1 + 0.01 * np.sin(2 * np.pi * 4 * t))
        c = 0.3 * np.sin(ph) + 0.08 * np.sin(2 * ph) + 0.004 * rng.standard_normal(t.size)
        c[: SR // 10] = 0.004 * rng.standard_normal(SR // 10)                     # an unvoiced lead-in
        clips.append(c.astype(np.float32))
    lengths = np.array([c.size for c in clips], np.int64)
    pad = (lengths + 3) // 4 * 4
    offsets = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.int64)
    buf = np.zeros(int(pad.sum()), np.float32)
    for c, o in zip(clips, offsets):
        buf[o:o + c.size] = c
    out = plan.f0_batch(buf, offsets, lengths, fmin, fmax, flags=0, want_frames=True)
    assert (out["status"] == 0).all()
    for i, c in enumerate(clips):
        T = 1 + c.size // 256
        f0 = out["f0_flat"][out["f0_offsets"][i]: out["f0_offsets"][i] + T]
        ref, _, _ = P.pyin(c, fmin, fmax, sr=SR, frame_length=1024, hop_length=256)
        same = np.isnan(f0) == np.isnan(ref)
        v = ~np.isnan(f0) & ~np.isnan(ref)
        same[v] &= np.abs(f0[v] - ref[v]) <= 1e-9 * ref[v]
        assert same.mean() >= 0.99, (fmin, fmax, i, same.mean(), np.flatnonzero(~same)[:12])
        assert v.sum() >= 0.5 * T, (fmin, fmax, i, int(v.sum()))                  # the tone is found
    # back on the default range the plan rebuilds its tables
    out2 = plan.f0_batch(buf, offsets, lengths, P.C2_HZ, P.C7_HZ, flags=0)
    assert (out2["status"] == 0).all()


def test_f0_random_mixtures(plan):
    """Voiced / noisy / silent stretches in random order: onsets, offsets and octave ambiguities."""
    rng = np.random.default_rng(11)
    clips = []
    for i in range(8):
        parts = []
        for _ in range(int(rng.integers(2, 5))):
            n = int(rng.integers(1500, 7000))
            kind = rng.integers(0, 4)
            if kind == 0:
                parts.append(np.zeros(n, np.float32))
            elif kind == 1:
                parts.append((0.05 * rng.standard_normal(n)).astype(np.float32))
            else:
                f = float(rng.uniform(80, 900))
                seg = voiced_tone(f, n / SR, vib=float(rng.uniform(0, 0.03)), seed=int(rng.integers(1 << 30)))[:n]
                parts.append(float(rng.uniform(0.1, 1.0)) * seg)
        clips.append(np.concatenate(parts).astype(np.float32))
    out, f0 = run(plan, clips, flags=0)
    # The observation columns agree to 1e-12 with the oracle, yet a decoded path may differ around a segment
    # boundary: for a strongly voiced frame the candidate probabilities sum to 1 up to rounding, librosa's unvoiced
    # observation is (1 - clip(sum, 0, 1)) / n_bins, i.e. exactly 0 or ~2e-19 depending on the last bit of that sum
    # (log: -708 or -43), and the last bit depends on BLAS's summation order inside trough_prior.dot(beta_probs).
    # No implementation can pin that bit; such frames are rare and confined to transitions.
    fracs = []
    for i, c in enumerate(clips):
        ref, _, _ = P.pyin(c, sr=SR, frame_length=1024, hop_length=256)
        same = np.isnan(f0[i]) == np.isnan(ref)
        v = ~np.isnan(f0[i]) & ~np.isnan(ref)
        same[v] &= np.abs(f0[i][v] - ref[v]) <= 1e-9 * ref[v]
        fracs.append(same.mean())
        NON_IDENTICAL[f"mixture{i}"] = (int((~same).sum()), int(same.size))
        print(f"[f0] mixture{i}: {int((~same).sum())} of {same.size} frames differ from the oracle's track")
        assert same.mean() >= 0.9, (i, same.mean(), np.flatnonzero(~same)[:12])
    assert np.mean(fracs) >= 0.98 and np.median(fracs) == 1.0, fracs
    # round 1 measured 6 frames of one clip; a drift from that shows here
    assert sum(NON_IDENTICAL[f"mixture{i}"][0] for i in range(8)) <= 12, {k: v for k, v in NON_IDENTICAL.items() if k.startswith("mixture")}


def test_f0_batch_is_clipwise_independent(plan):
    """Size-independent properties at batch scale: a clip's result does not depend on its neighbours, its position
    in the batch or the chunking of the workspace (every clip twice, shuffled; one clip also on its own)."""
    rng = np.random.default_rng(3)
    base = [voiced_tone(float(rng.uniform(90, 600)), float(rng.uniform(0.4, 2.5)), vib=0.01, seed=i) for i in range(24)]
    base += [make_clip(400 + i, SR, float(rng.uniform(0.5, 2.0)), speechy=bool(i & 1)) for i in range(8)]
    order = rng.permutation(2 * len(base))
    clips = [base[k % len(base)] for k in order]
    out, f0 = run(plan, clips, flags=N.FLAG_PREEMPH | N.FLAG_TRIM)
    assert (out["status"] == 0).all()
    first = {}
    for pos, k in enumerate(order):
        b = int(k % len(base))
        if b in first:
            np.testing.assert_array_equal(out["stats"][pos], out["stats"][first[b]])
            np.testing.assert_array_equal(f0[pos], f0[first[b]])
        else:
            first[b] = pos
    solo, f0s = run(plan, [base[5]], flags=N.FLAG_PREEMPH | N.FLAG_TRIM)
    np.testing.assert_array_equal(solo["stats"][0], out["stats"][first[5]])
    assert np.isfinite(out["stats"]).all() and (out["stats"][:, 2] >= 0).all() and (out["stats"][:, 2] <= 1).all()
    np.testing.assert_allclose(out["stats"][:, 2] + out["stats"][:, 3], 1.0, atol=1e-15)


def test_zz_report_non_identical_frames():
    """Not a check of its own: prints, per clip tested above, how many frames differed from the oracle's decoded
    track (run with -s, or read the assertion message of a failing bound), so that a drift from 0 is visible."""
    tot = sum(v[0] for v in NON_IDENTICAL.values())
    frames = sum(v[1] for v in NON_IDENTICAL.values())
    print(f"[f0] non-identical frames: {tot} of {frames} over {len(NON_IDENTICAL)} clips; per clip: "
          + ", ".join(f"{k}={v[0]}" for k, v in NON_IDENTICAL.items() if v[0]))
    single = {k: v for k, v in NON_IDENTICAL.items() if not k.startswith("mixture")}
    assert sum(v[0] for v in single.values()) == 0, single
