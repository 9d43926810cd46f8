"""GPU parity: the HIP path (through the C-ABI in include/afx.h) against the CPU oracle on
the same seeded inputs.  Run on the MI355X box with ``pytest -m gpu``."""
import numpy as np
import pytest

from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_clip
from oracle import cpu_ref as R
from tests.parity import check_frames, check_stats, oracle_stats

pytestmark = pytest.mark.gpu

CONFIGS = {
    "cfg2": dict(sr=22050, n_fft=1024, hop=256, n_mfcc=13),        # BASELINE configs[0..1,3]
    "cfg3": dict(sr=16000, n_fft=512, hop=128, n_mfcc=40),         # speech config
    "cfg5": dict(sr=44100, n_fft=2048, hop=512, n_mfcc=20),        # music config
    "small": dict(sr=8000, n_fft=256, hop=64, n_mfcc=13),
    "oddhop": dict(sr=16000, n_fft=512, hop=160, n_mfcc=13),
}


@pytest.fixture(scope="module")
def ctx():
    c = N.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def plans(ctx):
    cache = {}

    def get(name, window="hamming"):
        key = (name, window)
        if key not in cache:
            c = CONFIGS[name]
            cache[key] = N.Plan(ctx, N.make_params(c["sr"], c["n_fft"], c["hop"], c["n_mfcc"], 128, window, 0.97))
        return cache[key]
    return get


def run_one(plan, y, flags=N.FLAG_PREEMPH | N.FLAG_TRIM):
    out = plan.extract_batch(np.ascontiguousarray(y, np.float32), np.zeros(1, np.int64),
                             np.array([y.size], np.int64), flags=flags, want_frames=True)
    return out


def test_preprocess_bit_exact_and_trim_index(plans):
    plan = plans("cfg2")
    for idx, speechy in ((0, False), (1, True), (2, True)):
        y = make_clip(idx, 22050, 3.0, speechy=speechy)
        y_pre, s, e, st = plan.preprocess(y)
        ref_pre = R.preemphasis(y, 0.97)
        _, (rs, re) = R.trim(ref_pre)
        assert st == 0
        np.testing.assert_array_equal(y_pre, ref_pre)          # float32 bit-exact
        assert (s, e) == (rs, re)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_per_frame_and_stats_parity(plans, name):
    c = CONFIGS[name]
    plan = plans(name)
    for idx, speechy, secs in ((0, False, 2.0), (7, True, 2.5)):
        y = make_clip(idx, c["sr"], secs, speechy=speechy)
        out = run_one(plan, y)
        assert out["status"][0] == 0
        ref = oracle_stats(y, c["sr"], c["n_fft"], c["hop"], c["n_mfcc"])
        assert tuple(out["trim"][0]) == ref["trim"]
        assert out["nframes"][0] == ref["mfcc"].shape[1]
        what = f"{name} clip{idx}"
        check_frames(out["frames"][0], ref, what)
        check_stats(out["stats"][0], ref, c["n_mfcc"], what)
        # the float64 truth adjudicates: the GPU must be as close to it as the f32 oracle is (x4)
        truth = oracle_stats(y, c["sr"], c["n_fft"], c["hop"], c["n_mfcc"], dtype=np.float64)
        sc = np.abs(truth["mfcc"]).max(axis=1, keepdims=True)
        e_gpu = (np.abs(out["frames"][0]["mfcc"] - truth["mfcc"]) / sc).max()
        e_ref = (np.abs(ref["mfcc"] - truth["mfcc"]) / sc).max()
        assert e_gpu <= max(4 * e_ref, 2e-5), f"{what}: gpu {e_gpu:.2e} vs oracle-f32 {e_ref:.2e}"


@pytest.mark.parametrize("sr,n_fft,hop,K,n_mels,opt", [
    (22050, 1024, 256, 13, 40, dict(fmin=80.0, fmax=8000.0, htk=True, lifter=22.0)),     # the older extractor's options
    (16000, 512, 128, 13, 24, dict(fmin=0.0, fmax=8000.0, htk=True, lifter=0.0)),
    (44100, 2048, 512, 20, 64, dict(fmin=30.0, fmax=16000.0, htk=False, lifter=40.0)),
    (8000, 256, 64, 13, 40, dict(fmin=100.0, fmax=3800.0, htk=True, lifter=22.0)),        # generic kernel
])
def test_mel_and_mfcc_option_variants(ctx, sr, n_fft, hop, K, n_mels, opt):
    """fmin / fmax / htk / lifter (04_feature_extraction_experiment/audio_feature_extraction 2/audio_feature_extraction/
    feature_extractor.py:148-181) through every frame kernel."""
    plan = N.Plan(ctx, N.make_params(sr, n_fft, hop, K, n_mels, "hamming", 0.97, **opt))
    try:
        for idx, speechy in ((21, False), (22, True)):
            y = make_clip(idx, sr, 1.5, speechy=speechy)
            out = run_one(plan, y)
            assert out["status"][0] == 0
            ref = R.extract_stats(y, sr=sr, frame_length=n_fft, hop_length=hop, n_mfcc=K, n_mels=n_mels, return_frames=True, **opt)
            assert tuple(out["trim"][0]) == ref["trim"]
            check_frames(out["frames"][0], ref, f"opt {sr}/{n_fft} clip{idx}")
            check_stats(out["stats"][0], ref, K, f"opt {sr}/{n_fft} clip{idx}")
    finally:
        plan.close()


def test_config1_five_second_clip(plans):
    y = make_clip(0, 22050, 5.0)
    out = run_one(plans("cfg2"), y)
    assert out["nframes"][0] == 431                            # SURVEY.md section 8 size table
    ref = oracle_stats(y, 22050, 1024, 256, 13)
    check_frames(out["frames"][0], ref, "cfg1")
    check_stats(out["stats"][0], ref, 13, "cfg1")


def test_hann_window(plans):
    y = make_clip(4, 22050, 2.0)
    out = run_one(plans("cfg2", "hann"), y)
    ref = oracle_stats(y, 22050, 1024, 256, 13, window="hann")
    check_frames(out["frames"][0], ref, "hann")


def test_staged_api_no_preemph_no_trim(plans):
    # extract_mfcc(y) / extract_energy(y) of an already processed signal (flags = 0)
    y = make_clip(5, 22050, 2.0, speechy=True)
    yp, _ = R.preprocess_audio(y)
    out = run_one(plans("cfg2"), yp, flags=0)
    ref_m = R.extract_mfcc(yp, 22050, 13, 1024, 256, return_frames=True)
    ref_e = R.extract_energy(yp, 1024, 256, return_frames=True)
    ref = {**ref_m, **ref_e}
    check_frames(out["frames"][0], ref, "staged")
    check_stats(out["stats"][0], ref, 13, "staged")


def test_ragged_batch_with_edge_cases(plans):
    plan = plans("cfg2")
    sr = 22050
    clips = [
        make_clip(10, sr, 1.3),
        np.zeros(sr, np.float32),                                  # digital silence: not trimmed, c0 = -100*sqrt(128)
        make_clip(11, sr, 0.05),                                   # 1102 samples -> T = 5 < 9
        make_clip(12, sr, 2.0, speechy=True),
        make_clip(13, sr, 0.7),
        np.array([0.5], np.float32),                               # < 2 samples
        make_clip(14, sr, 1.0),
    ]
    bad = make_clip(15, sr, 1.0).copy()
    bad[1234] = np.nan
    clips.insert(3, bad)
    lengths = np.array([c.size for c in clips], np.int64)
    pad = (lengths + 3) // 4 * 4
    offsets = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.int64)
    buf = np.zeros(int(pad.sum()), np.float32)
    for c, o in zip(clips, offsets):
        buf[o:o + c.size] = c
    out = plan.extract_batch(buf, offsets, lengths, want_frames=True)
    expect = [0, 0, N.CLIP_TOO_SHORT, N.CLIP_NONFINITE, 0, 0, N.CLIP_TOO_SHORT, 0]
    assert out["status"].tolist() == expect
    for i, c in enumerate(clips):
        if expect[i]:
            with pytest.raises(ValueError):
                oracle_stats(c, sr, 1024, 256, 13)                 # the oracle raises where the GPU flags
            continue
        ref = oracle_stats(c, sr, 1024, 256, 13)
        check_frames(out["frames"][i], ref, f"ragged{i}")
        check_stats(out["stats"][i], ref, 13, f"ragged{i}")
    assert out["stats"][1][0] == pytest.approx(-100.0 * np.sqrt(128), rel=1e-5)
    # unaligned packing (offsets not multiples of 4) takes the scalar load path: same results
    offs2 = np.concatenate([[1], 1 + np.cumsum(lengths + 1)[:-1]]).astype(np.int64)
    buf2 = np.zeros(int(offs2[-1] + lengths[-1] + 8), np.float32)
    for c, o in zip(clips, offs2):
        buf2[o:o + c.size] = c
    out2 = plan.extract_batch(buf2, offs2, lengths)
    assert out2["status"].tolist() == expect
    good = [i for i, e in enumerate(expect) if e == 0]
    np.testing.assert_array_equal(out2["stats"][good], out["stats"][good])


def test_s16_upload_matches_f32_of_same_pcm(plans):
    plan = plans("cfg2")
    y = make_clip(20, 22050, 1.5)
    q = np.clip(np.rint(y.astype(np.float64) * 32768), -32768, 32767).astype(np.int16)
    yf = q.astype(np.float32) / np.float32(32768.0)
    a = plan.extract_batch(q, np.zeros(1, np.int64), np.array([q.size], np.int64), fmt=N.FMT_S16)
    b = plan.extract_batch(yf, np.zeros(1, np.int64), np.array([q.size], np.int64))
    np.testing.assert_array_equal(a["stats"], b["stats"])        # /32768 is exact: bit-identical


def test_device_resident_input_and_timing(ctx, plans):
    plan = plans("cfg2")
    clips = [make_clip(30 + i, 22050, 1.0) for i in range(8)]
    n = clips[0].size
    buf = np.concatenate(clips)
    offsets = np.arange(8, dtype=np.int64) * n
    lengths = np.full(8, n, np.int64)
    host = plan.extract_batch(buf, offsets, lengths)
    d = N.DeviceBuffer(ctx, buf.nbytes)
    d.upload(buf)
    plan.set_timing(True)
    plan.timings(reset=True)
    dev = plan.extract_batch(d, offsets, lengths)
    t = plan.timings()
    plan.set_timing(False)
    d.free()
    np.testing.assert_array_equal(host["stats"], dev["stats"])
    assert t["frames"][1] == 1 and t["frames"][0] > 0.0


def test_linearity_of_rms_and_shift_of_mfcc(plans):
    # size-independent properties: scaling the input by g scales RMS by g and shifts c0 by
    # 20*log10(g)*sqrt(128), leaving the other coefficients (no clamping active) unchanged
    plan = plans("cfg2")
    y = make_clip(40, 22050, 2.0)
    g = np.float32(0.5)
    a, b = run_one(plan, y), run_one(plan, y * g)
    np.testing.assert_allclose(b["frames"][0]["rms"], a["frames"][0]["rms"] * g, rtol=1e-6)
    d = b["frames"][0]["mfcc"] - a["frames"][0]["mfcc"]
    np.testing.assert_allclose(d[0], 20 * np.log10(0.5) * np.sqrt(128), rtol=1e-5)
    assert np.abs(d[1:]).max() < 2e-3


@pytest.mark.parametrize("n_mels", [40, 100, 64, 20, 256])
def test_other_mel_counts_on_the_tuned_kernel(ctx, n_mels):
    """n_mels that are not a multiple of 8 leave the last filter oct partly empty (1024/256 path); more than 128
    filters fall back to the generic kernel."""
    from oracle import cpu_ref as R
    plan = N.Plan(ctx, N.make_params(22050, 1024, 256, 13, n_mels, "hamming", 0.97))
    try:
        y = make_clip(31, 22050, 1.2, speechy=True)
        out = run_one(plan, y)
        assert out["status"][0] == 0
        ref = R.extract_stats(y, sr=22050, frame_length=1024, hop_length=256, n_mfcc=13, n_mels=n_mels, return_frames=True)
        check_frames(out["frames"][0], ref, f"mels{n_mels}")
        check_stats(out["stats"][0], ref, 13, f"mels{n_mels}")
    finally:
        plan.close()


def test_zero_crossing_rate_is_exact(plans):
    from oracle import cpu_ref as R
    plan = plans("cfg2")
    rng = np.random.default_rng(9)
    clips = [make_clip(40, 22050, 1.0), make_clip(41, 22050, 1.4, speechy=True),
             np.tile(np.array([1, -1], np.float32), 3000), (1e-11 * rng.standard_normal(4000)).astype(np.float32),
             make_clip(42, 22050, 0.03), np.array([0.2], np.float32)]
    lengths = np.array([c.size for c in clips], np.int64)
    pad = (lengths + 3) // 4 * 4
    offsets = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.int64)
    buf = np.zeros(int(pad.sum()), np.float32)
    for c, o in zip(clips, offsets):
        buf[o:o + c.size] = c
    for flags in (0, N.FLAG_PREEMPH | N.FLAG_TRIM):
        out = plan.zcr_batch(buf, offsets, lengths, flags=flags)
        assert (out["status"] == 0).all()
        for i, c in enumerate(clips):
            if flags and c.size < 2:
                continue                                      # the oracle's pre-emphasis needs two samples
            yp = R.preprocess_audio(c)[0] if flags else c
            ref = R.zero_crossing_rate(yp, 1024, 256)
            got = out["zcr_flat"][out["zcr_offsets"][i]: out["zcr_offsets"][i] + ref.size]
            np.testing.assert_array_equal(got, ref, err_msg=f"clip {i} flags {flags}")
