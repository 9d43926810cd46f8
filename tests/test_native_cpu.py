"""CPU-only checks of libafx.so: it loads, exports every symbol include/afx.h declares, and its
host-side table builders agree with the oracle.  No device compute is attempted here."""
import ctypes
import os
import re

import numpy as np
import pytest

from audio_feature_extraction_amd import _native as N
from oracle import cpu_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    lib = N.lib()
    header = open(os.path.join(ROOT, "include", "afx.h")).read()
    assert lib.afx_version() == int(re.search(r"#define\s+AFX_VERSION\s+(\d+)", header).group(1))
    declared = set(re.findall(r"\b(afx_[a-z0-9_]+)\s*\(", header))
    declared -= {"afx_status", "afx_clip_status"}
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert isinstance(lib.afx_last_error(), bytes)


def test_params_struct_matches_header_layout():
    p = N.Params()
    N.lib().afx_default_params(ctypes.byref(p))
    assert (p.sr, p.n_fft, p.hop, p.n_mfcc, p.n_mels) == (22050, 1024, 256, 13, 128)   # reference defaults
    assert p.window == N.WINDOW_HAMMING and p.preemph == pytest.approx(0.97)
    assert (p.trim_top_db, p.trim_frame, p.trim_hop) == (30.0, 2048, 512)
    assert (p.top_db, p.delta_width) == (80.0, 9) and p.amin == pytest.approx(1e-10)
    assert ctypes.sizeof(N.Params) == 72
    assert (p.fmin, p.fmax, p.htk, p.lifter) == (0.0, 0.0, 0, 0.0)          # the packaged class passes none of them


@pytest.mark.parametrize("sr,n_fft,n_mfcc,window", [(22050, 1024, 13, "hamming"), (16000, 512, 40, "hamming"),
                                                     (44100, 2048, 20, "hann"), (8000, 256, 13, "hamming")])
def test_host_tables_match_oracle(sr, n_fft, n_mfcc, window):
    import scipy.fft
    p = N.make_params(sr, n_fft, n_fft // 4, n_mfcc, 128, window)
    win, mel, dct = N.build_tables(p)
    np.testing.assert_array_equal(mel, R.mel_filterbank(sr, n_fft, 128))       # librosa's float32 values, bit for bit
    np.testing.assert_allclose(win, R.get_window(window, n_fft).astype(np.float32), rtol=0, atol=1e-7)
    D = scipy.fft.dct(np.eye(128), axis=0, type=2, norm="ortho")[:n_mfcc]
    np.testing.assert_allclose(dct, D, atol=1e-7)


def test_invalid_parameters_are_rejected_with_messages():
    with pytest.raises(NotImplementedError):
        N.build_tables(N.make_params(22050, 1000, 250, 13))        # frame_length not a power of two
    with pytest.raises(ValueError):
        N.build_tables(N.make_params(22050, 1024, 256, 200))       # n_mfcc > n_mels
    with pytest.raises(ValueError):
        N.make_params(window="blackman")


def test_no_device_means_loud_failure_not_fallback():
    if N.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(N.AfxError):
        N.Context(0)
    from audio_feature_extraction_amd import AudioFeatureExtractor
    ex = AudioFeatureExtractor()
    with pytest.raises(N.AfxError):
        ex.extract_mfcc(np.zeros(4096, np.float32))


def test_f0_host_tables_match_the_pyin_oracle():
    """The tables afx_f0_batch uploads (closed-form Beta masses, log-transition row classes, bin
    frequencies) against oracle/pyin_ref.py, which builds them with scipy as librosa does."""
    from oracle import pyin_ref as P
    for sr, n_fft, hop in [(22050, 1024, 256), (16000, 512, 128), (44100, 2048, 512)]:
        t = N.f0_build_tables(sr, n_fft, hop, P.C2_HZ, P.C7_HZ)
        mn, mx = P.periods(sr, P.C2_HZ, P.C7_HZ, n_fft, n_fft // 2)
        tb = P.pyin_tables(sr, P.C2_HZ, P.C7_HZ, hop)
        assert (t["min_period"], t["max_period"], t["n_bins"]) == (mn, mx, tb["n_pitch_bins"])
        assert 2 * t["band"] + 1 == tb["transition_width"]
        assert t["cap"] >= (t["n_lag"] + 1) // 2
        np.testing.assert_allclose(t["beta"], tb["beta_probs"], rtol=0, atol=5e-15)   # closed form vs scipy.special.betainc
        nb, band = t["n_bins"], t["band"]
        np.testing.assert_allclose(t["freqs"], P.C2_HZ * 2 ** (np.arange(nb) / 120), rtol=4e-16)     # pow: one ulp
        logA = np.log(tb["transition"] + np.finfo(np.float64).tiny)
        w = 2 * band + 1
        for b in list(range(0, band + 2)) + [nb // 2] + list(range(nb - band - 2, nb)):
            rc = 1 + b if b < band else (1 + band + (nb - 1 - b) if b > nb - 1 - band else 0)
            for d in range(w):
                j = b + d - band
                if 0 <= j < nb:
                    assert abs(t["lt"][0, rc, d] - logA[b, j]) < 1e-13            # voiced -> voiced (stay)
                    assert abs(t["lt"][1, rc, d] - logA[b, nb + j]) < 1e-13       # voiced -> unvoiced (switch)
                    assert abs(t["lt"][0, rc, d] - logA[nb + b, nb + j]) < 1e-13


@pytest.mark.parametrize("sr,n_fft,n_mels", [(22050, 1024, 128), (22050, 1024, 40), (22050, 1024, 64), (16000, 1024, 128),
                                               (44100, 1024, 256), (8000, 1024, 20),
                                               (44100, 2048, 128), (22050, 2048, 40), (16000, 512, 128), (8000, 512, 40)])
def test_wave_mel_schedule_reproduces_the_filterbank(sr, n_fft, n_mels):
    """The per-lane tap schedule of the wave-level frame kernels is a re-ordering of librosa.filters.mel: summing
    every lane's weights back onto (filter, bin) must give the dense float32 matrix exactly, every real filter
    must have exactly one owner lane per spectrum, first bins are aligned to the 16-byte reads (2 bins of a frame
    pair at n_fft 1024 / 512, 4 bins of one frame at 2048), no padded tap leaves the LDS image, and at the
    reference configurations a width-1 round puts no two different read addresses on one LDS slot within a
    ds_read_b128 lane group.  n_fft 512: two spectra per wave, the upper half-wave repeats the lower one's schedule."""
    p = N.make_params(sr, n_fft, n_fft // 4, 13, n_mels)
    _, mel, _ = N.build_tables(p)
    s = N.build_mel_schedule(p)
    nbins = n_fft // 2 + 1
    align = 4 if n_fft == 2048 else 2
    lanes = 32 if n_fft == 512 else 64
    max_slot = {1024: 1087, 2048: 2175, 512: 543}[n_fft]
    dense = np.zeros((n_mels, nbins + 64), np.float64)
    owners = np.zeros(n_mels, int)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15] + list(range(20, 28)), list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31],
              [32, 33, 34, 35, 44, 45, 46, 47] + list(range(52, 60)), list(range(36, 44)) + [48, 49, 50, 51, 60, 61, 62, 63]]
    for r in range(s["rounds"]):
        nb, wd = s["nb"][r], s["width"][r]
        w = s["weights"][s["woff"][r]: s["woff"][r] + nb * 256].reshape(nb, 64, 4)
        if lanes == 32:
            np.testing.assert_array_equal(s["meta"][r, :32], s["meta"][r, 32:])
            np.testing.assert_array_equal(w[:, :32], w[:, 32:])
        for lane in range(lanes):
            meta = int(s["meta"][r, lane])
            bin0, m, own = meta & 2047, (meta >> 11) & 511, (meta >> 20) & 1
            assert bin0 % align == 0
            taps = w[:, lane, :].reshape(-1)
            if own:
                owners[m] += 1
            if taps.any():
                assert m < n_mels
                dense[m, bin0: bin0 + 4 * nb] += taps
            assert bin0 + 4 * nb - 1 <= max_slot
        if wd == 1 and (sr, n_fft, n_mels) in ((22050, 1024, 128), (44100, 2048, 128)):     # a perfect matching exists there
            for g in groups[: lanes // 16]:
                slots = {}
                for lane in g:
                    b0 = int(s["meta"][r, lane]) & 2047
                    slots.setdefault((b0 // align) % 16, set()).add(b0)
                assert all(len(v) == 1 for v in slots.values()), (r, slots)
    assert (owners == 1).all()
    np.testing.assert_array_equal(dense[:, :nbins].astype(np.float32), mel)
    assert not dense[:, nbins:].any()


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax,htk", [(22050, 1024, 40, 80.0, 8000.0, True), (16000, 512, 24, 0.0, 8000.0, True),
                                                          (22050, 1024, 128, 50.0, 7600.0, False), (44100, 2048, 64, 20.0, None, False)])
def test_mel_option_variants_match_the_oracle_bit_for_bit(sr, n_fft, n_mels, fmin, fmax, htk):
    """fmin / fmax / htk as the reference's older extractor passes them (04_feature_extraction_experiment/
    audio_feature_extraction 2/audio_feature_extraction/feature_extractor.py:148-155)."""
    p = N.make_params(sr, n_fft, n_fft // 4, 13, n_mels, fmin=fmin, fmax=fmax, htk=htk)
    _, mel, _ = N.build_tables(p)
    np.testing.assert_array_equal(mel, R.mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk))
    # HTK known answers: mel(700 Hz) = 2595 log10(2), and the band edges are evenly spaced on that scale
    if htk:
        assert abs(2595.0 * np.log10(2.0) - 781.17284) < 1e-4           # mel(700 Hz) on the HTK scale
        f = R.mel_frequencies(n_mels + 2, fmin, fmax, htk=True)
        m = 2595.0 * np.log10(1.0 + f / 700.0)
        np.testing.assert_allclose(np.diff(m), np.diff(m)[0], rtol=1e-9)
    with pytest.raises(ValueError):
        N.build_tables(N.make_params(sr, n_fft, n_fft // 4, 13, n_mels, fmin=4000.0, fmax=3000.0))


def test_lifter_scales_the_dct_rows():
    p0 = N.make_params(22050, 1024, 256, 13, 40)
    p1 = N.make_params(22050, 1024, 256, 13, 40, lifter=22.0)
    d0, d1 = N.build_tables(p0)[2], N.build_tables(p1)[2]
    n = np.arange(1, 14, dtype=np.float32)
    f = (1 + (22.0 / 2) * np.sin(np.pi * n / 22.0)).astype(np.float32)      # librosa.feature.mfcc(lifter=22)
    np.testing.assert_allclose(d1, d0 * f[:, None], rtol=2e-7)
    assert f[0] > 1.5 and abs(f[10] - (1 + 11 * np.sin(np.pi * 11 / 22))) < 1e-5      # n = 11 = L / 2: factor 1 + L / 2


def test_bench_interval_union():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.union_ms([]) == 0.0
    assert abs(bench.union_ms([(0.0, 1.0), (0.5, 2.0), (3.0, 3.5)]) - 2.5) < 1e-12
    assert abs(bench.union_ms([(2.0, 3.0), (0.0, 1.0), (0.25, 0.5)]) - 2.0) < 1e-12
    assert 1 <= bench.usable_cpus() <= (os.cpu_count() or 1)
