"""Shared parity helpers for the GPU tests.

Tolerances (BASELINE.json north_star): MFCC-derived values within 1e-4 relative, RMS
energy within 1e-5 relative.  "Relative" for a coefficient row means relative to that
row's magnitude, |a - b| <= rtol * max|ref_row|: high-order MFCCs and delta means cross
zero, so a per-element ratio is meaningless (hard parts).  For the
delta / delta2 means -- which are ~1e-2 while the coefficients they difference are
~1e2 -- the row scale is floored at 1e-3 of the coefficient scale; per-frame delta rows are held to 1e-4 of their
own maximum (floor: 1e-2 of their coefficient row)."""
import numpy as np

from oracle import cpu_ref as R

MFCC_RTOL = 1e-4
RMS_RTOL = 1e-5


def oracle_stats(y, sr, n_fft, hop, n_mfcc, window="hamming", pre_emphasis=0.97, dtype=np.float32):
    return R.extract_stats(y, sr=sr, frame_length=n_fft, hop_length=hop, n_mfcc=n_mfcc,
                           pre_emphasis=pre_emphasis, window=window, dtype=dtype, return_frames=True)


def assert_rows_close(got, ref, rtol, what, floor=0.0):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    if ref.ndim == 1:
        got, ref = got[:, None], ref[:, None]
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), floor)
    err = np.abs(got - ref) / np.maximum(scale, 1e-30)
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    assert err.max() <= rtol, f"{what}: max scaled error {err.max():.3e} > {rtol:g} (row {np.argmax(err.max(axis=1))})"


def check_frames(fr, ref, what=""):
    """fr: dict of per-frame GPU arrays; ref: oracle dict (return_frames=True)."""
    assert fr["mfcc"].shape == ref["mfcc"].shape, f"{what} frames {fr['mfcc'].shape} vs {ref['mfcc'].shape}"
    gmax = float(np.abs(ref["mfcc"]).max())
    # rows that are (numerically) zero, e.g. c1.. of digital silence, are scaled by 1e-3 of the largest row
    assert_rows_close(fr["mfcc"], ref["mfcc"], MFCC_RTOL, what + " mfcc", floor=1e-3 * gmax)
    cscale = np.maximum(np.abs(ref["mfcc"]).max(axis=1, keepdims=True), 1e-3 * gmax)
    # deltas on their OWN row scale (round 1 scaled them by the ~100x larger coefficient row): 1e-4 of the delta row's
    # maximum, floored at 1e-2 of the coefficient row (a delta row of a nearly constant coefficient is pure rounding)
    for k in ("mfcc_delta", "mfcc_delta2"):
        dscale = np.maximum(np.abs(ref[k]).max(axis=1, keepdims=True), 1e-2 * cscale)
        err = np.abs(np.asarray(fr[k], np.float64) - ref[k]) / dscale
        assert err.max() <= MFCC_RTOL, f"{what} {k}: {err.max():.3e} (row {np.argmax(err.max(axis=1))})"
    np.testing.assert_allclose(fr["rms"], ref["rms"], rtol=RMS_RTOL, atol=1e-9, err_msg=what + " rms")


def check_stats(stats, ref, K, what=""):
    s = np.asarray(stats, np.float64)
    cscale = float(np.abs(ref["mfcc_mean"]).max())
    assert_rows_close(s[0:K], ref["mfcc_mean"], MFCC_RTOL, what + " mfcc_mean", floor=1e-3 * cscale)
    # float32 np.std of a (nearly) constant row is off by ~1 ulp of the row's mean (the float32 mean
    # itself is rounded), an error the float64-accumulating GPU reduction does not share
    std_err = np.abs(s[K:2 * K] - np.asarray(ref["mfcc_std"], np.float64))
    std_tol = MFCC_RTOL * max(float(np.abs(ref["mfcc_std"]).max()), 1e-3 * cscale) \
        + 4 * np.finfo(np.float32).eps * np.abs(np.asarray(ref["mfcc_mean"], np.float64))
    assert (std_err <= std_tol).all(), f"{what} mfcc_std: {std_err.max():.3e}"
    assert_rows_close(s[2 * K:3 * K], ref["mfcc_delta_mean"], MFCC_RTOL, what + " mfcc_delta_mean", floor=1e-3 * cscale)
    assert_rows_close(s[3 * K:4 * K], ref["mfcc_delta2_mean"], MFCC_RTOL, what + " mfcc_delta2_mean", floor=1e-3 * cscale)
    e = np.array([ref["energy_mean"], ref["energy_std"], ref["energy_range"]], np.float64)
    np.testing.assert_allclose(s[4 * K:4 * K + 3], e, rtol=RMS_RTOL, atol=1e-8 + RMS_RTOL * e[0], err_msg=what + " energy")
