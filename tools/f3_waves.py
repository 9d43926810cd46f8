#!/usr/bin/env python3
"""Per-wave timeline of the speculative k_frames3 launch (diagnostic build: `make -C audio_feature_extraction_amd/csrc dbg`).

  AFX_LIB=$PWD/audio_feature_extraction_amd/libafx_dbg.so python tools/f3_waves.py [clips]

Prints when the waves of the launch start and end (100 MHz wall clock, relative to the first wave's entry), per
XCD and per CU: a long tail or a late start is time the frame kernel spends below full occupancy.
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audio_feature_extraction_amd import _native as N          # noqa: E402
from audio_feature_extraction_amd.synth import make_batch      # noqa: E402


def main() -> None:
    n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    samples, offsets, lengths = make_batch(n_clips, 22050, 10.0, first_index=0, workers=8)
    ctx = N.Context(0)
    plan = N.Plan(ctx, N.make_params(22050, 1024, 256, 13, 128))
    dbuf = N.DeviceBuffer(ctx, samples.nbytes)
    dbuf.upload(samples)
    out = None
    for _ in range(5):
        out = plan.extract_batch(dbuf, offsets, lengths, out=out)
    lib = ctypes.CDLL(N.LIB_PATH)
    waves = int(os.environ.get("AFX_F3_WAVES", "16")) * 256
    buf = np.zeros(4 * waves, dtype=np.uint64)
    rc = lib.afx_debug_f3_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(waves))
    if rc != 0:
        raise SystemExit(f"afx_debug_f3_stamps: {rc}")
    st = buf.reshape(waves, 4)
    live = st[:, 2] > 0
    st = st[live]
    t0 = st[:, 0].min()
    entry = (st[:, 0] - t0).astype(np.float64) / 100.0        # microseconds
    ready = (st[:, 1] - t0).astype(np.float64) / 100.0
    end = (st[:, 2] - t0).astype(np.float64) / 100.0
    hw = st[:, 3] & np.uint64(0xFFFFFFFF)
    xcc = (st[:, 3] >> np.uint64(32)) & np.uint64(0xF)
    cu = (hw >> np.uint64(8)) & np.uint64(0xF)
    sh = (hw >> np.uint64(12)) & np.uint64(0x1)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    simd = (hw >> np.uint64(4)) & np.uint64(0x3)
    q = lambda a: "min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]))
    print(f"waves {len(st)}  kernel span {end.max():.1f} us")
    print("entry  :", q(entry))
    print("ready  :", q(ready))
    print("end    :", q(end))
    print("busy   :", q(end - ready), " (end - ready per wave)")
    print("mean busy / span = %.3f" % ((end - ready).mean() / end.max()))
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"xcd {x}: waves {m.sum():5d} entry p50 {np.median(entry[m]):7.1f} end p50 {np.median(end[m]):7.1f} "
              f"end max {end[m].max():7.1f} busy mean {np.mean((end - ready)[m]):7.1f}")
    key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
    ends = np.array([end[key == k].max() for k in sorted(set(key.tolist()))])
    starts = np.array([entry[key == k].min() for k in sorted(set(key.tolist()))])
    print(f"distinct CUs {len(ends)}: CU end  {q(ends)}")
    print(f"                      CU start {q(starts)}")
    # per block: duration = end[b] - end[b - 1] inside one wave's run (the run's first block: from `ready`)
    nblk = 54 * n_clips
    if nblk <= 65536:
        bt = np.zeros(nblk, dtype=np.uint64)
        rc = lib.afx_debug_f3_blocks(bt.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nblk))
        if rc != 0:
            raise SystemExit(f"afx_debug_f3_blocks: {rc}")
        be = (bt - t0).astype(np.float64) / 100.0
        dur = np.zeros(nblk)
        kind = np.zeros(nblk, dtype=np.int64)      # 0 interior, 1 clip's first block, 2 clip's last, 3 run's first (interior)
        for w in range(waves):
            lo, hi = w * nblk // waves, (w + 1) * nblk // waves
            prev = ready[w] if w < len(ready) else 0.0
            for b in range(lo, hi):
                dur[b] = be[b] - prev
                prev = be[b]
                pos = b % 54
                kind[b] = 1 if pos == 0 else (2 if pos == 53 else (4 if pos == 52 else (3 if b == lo else 0)))
        for k, name in ((0, "interior, chained"), (3, "run's first (20-row load)"), (1, "clip's first block"),
                        (4, "clip's last but one"), (2, "clip's last block")):
            m = kind == k
            if m.any():
                print(f"block {name:28s}: n {m.sum():6d}  " + q(dur[m]))
        nb = np.array([(w + 1) * nblk // waves - w * nblk // waves for w in range(waves)])
        for v in sorted(set(nb.tolist())):
            m = nb == v
            print(f"waves with {v} blocks: {m.sum():5d}  busy " + q((end - ready)[m[:len(end)]]))
    per_simd = {}
    for k, s_, b in zip(key.tolist(), simd.tolist(), (end - ready).tolist()):
        per_simd.setdefault((k, s_), []).append(b)
    cnt = np.array([len(v) for v in per_simd.values()])
    print("waves per SIMD: min %d max %d mean %.2f" % (cnt.min(), cnt.max(), cnt.mean()))


if __name__ == "__main__":
    main()
