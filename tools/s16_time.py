"""int16 vs float32 device-resident batches through afx_extract_batch: python tools/s16_time.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_batch
n = 1000
samples, offsets, lengths = make_batch(n, 22050, 10.0, workers=16)
q = np.clip(np.rint(samples.astype(np.float64) * 32768), -32768, 32767).astype(np.int16)
ctx = N.Context(0); plan = N.Plan(ctx, N.make_params(22050, 1024, 256, 13))
for name, arr, fmt in (("f32", samples, N.FMT_F32), ("s16", q, N.FMT_S16)):
    d = N.DeviceBuffer(ctx, arr.nbytes); d.upload(arr)
    out = None
    for _ in range(3): out = plan.extract_batch(d, offsets, lengths, fmt=fmt, out=out)
    plan.set_timing(True); plan.timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(20): out = plan.extract_batch(d, offsets, lengths, fmt=fmt, out=out)
    dt = (time.perf_counter() - t0) / 20
    kt = plan.timings()
    print(name, f"{dt*1e3:.3f} ms/step", {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()})
    plan.set_timing(False)
    d.free()
