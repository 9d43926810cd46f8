"""Device-resident timing of the other BASELINE configs (parity cases, not bench lines): python tools/cfg_time.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_batch
for name, sr, nfft, hop, K, n in (("cfg3", 16000, 512, 128, 40, 1000), ("cfg5", 44100, 2048, 512, 20, 1000), ("cfg2", 22050, 1024, 256, 13, 1000)):
    samples, offsets, lengths = make_batch(n, sr, 10.0, workers=16)
    ctx = N.Context(0); plan = N.Plan(ctx, N.make_params(sr, nfft, hop, K))
    d = N.DeviceBuffer(ctx, samples.nbytes); d.upload(samples)
    out = None
    for _ in range(3): out = plan.extract_batch(d, offsets, lengths, out=out)
    plan.set_timing(True); plan.timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(10): out = plan.extract_batch(d, offsets, lengths, out=out)
    dt = (time.perf_counter() - t0) / 10
    fr = int(out["nframes"].sum())
    kt = plan.timings()
    print(name, f"{fr} frames {dt*1e3:.3f} ms/step {fr/dt/1e6:.1f} Mframes/s  algorithmic {fr*4*hop/dt/1e9:.0f} GB/s",
          {k: round(v[0] / max(v[1], 1), 3) for k, v in kt.items()})
    d.free(); plan.close(); ctx.close()
