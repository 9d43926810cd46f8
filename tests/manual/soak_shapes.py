"""One-off robustness run over the three wave-level shapes: ragged clips with leading / trailing silence (so that the trim
cuts and the redo launch are exercised), statistics and per-frame rows of a sample checked against the oracle."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_clip
from tests.parity import check_frames, check_stats, oracle_stats

for sr, n_fft, hop, K in ((22050, 1024, 256, 13), (16000, 512, 128, 40), (44100, 2048, 512, 20)):
    rng = np.random.default_rng(sr)
    base = [make_clip(900 + i, sr, 2.5, speechy=bool(i % 2)) for i in range(24)]
    clips = []
    for i in range(1500):
        b = base[i % 24]
        L = int(rng.integers(hop * 6, b.size))
        o = int(rng.integers(0, b.size - L + 1))
        c = b[o:o + L]
        if i % 3 == 0:
            c = np.concatenate([np.zeros(int(rng.integers(0, 6000)), np.float32), c, np.zeros(int(rng.integers(0, 6000)), np.float32)])
        clips.append(c)
    lengths = np.array([c.size for c in clips], np.int64)
    offsets = np.concatenate([[0], np.cumsum((lengths + 3) // 4 * 4)[:-1]]).astype(np.int64)
    buf = np.zeros(int(offsets[-1] + lengths[-1] + 8), np.float32)
    for c, o in zip(clips, offsets):
        buf[o:o + c.size] = c
    ctx = N.Context(0); plan = N.Plan(ctx, N.make_params(sr, n_fft, hop, K))
    t0 = time.perf_counter(); out = plan.extract_batch(buf, offsets, lengths); t1 = time.perf_counter()
    outf = plan.extract_batch(buf, offsets, lengths, want_frames=True)
    assert np.array_equal(out["status"], outf["status"]) and np.array_equal(out["trim"], outf["trim"])
    ok = out["status"] == 0
    # the statistics-only call runs k_tail, the per-frame call k_dct16* + k_stats: equal to a few float32 ulps of the values they
    # are formed from (row sums in another order, the folded DCT), each value on its own scale floored at 1e-2 of the clip's
    # largest coefficient mean -- tests/test_gpu_pipeline.py::test_fused_tail_equals_the_two_kernel_tail
    a, b = out["stats"][ok].astype(np.float64), outf["stats"][ok].astype(np.float64)
    scale = np.maximum(np.abs(b), 1e-2 * np.abs(b[:, :K]).max(axis=1, keepdims=True))
    assert (np.abs(a - b) <= 2e-5 * scale).all(), "statistics differ between the two output modes: %g" % (np.abs(a - b) / scale).max()
    checked = short = 0
    for i in rng.choice(len(clips), 70, replace=False):
        try:
            ref = oracle_stats(clips[i], sr, n_fft, hop, K)
        except ValueError:
            assert out["status"][i] == N.CLIP_TOO_SHORT, (i, out["status"][i]); short += 1; continue
        assert out["status"][i] == 0, (i, out["status"][i])
        assert tuple(out["trim"][i]) == tuple(ref["trim"]), (i, out["trim"][i], ref["trim"])
        check_stats(out["stats"][i], ref, K, f"{sr}/{n_fft} clip {i}")
        check_frames(outf["frames"][i], ref, f"{sr}/{n_fft} clip {i}")
        checked += 1
    print(f"{sr}/{n_fft}/{hop}/{K}: {len(clips)} clips in {1e3 * (t1 - t0):.1f} ms, status counts {np.bincount(out['status'], minlength=3).tolist()}, "
          f"{checked} checked against the oracle (trim indices exact), {short} too short")
    plan.close(); ctx.close()
