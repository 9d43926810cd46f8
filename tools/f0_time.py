"""Times afx_f0_batch on n synthetic 10 s clips (device-resident): python tools/f0_time.py [n] [sr n_fft hop]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sr, nfft, hop = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (22050, 1024, 256)
ctx = N.Context(0); plan = N.Plan(ctx, N.make_params(sr, nfft, hop, 13))
samples, offsets, lengths = make_batch(n, sr, 10.0, workers=16)
d = N.DeviceBuffer(ctx, samples.nbytes); d.upload(samples)
for _ in range(3):      # the first calls of a process still pay the first touch of the ~14 KB per frame workspace
    plan.f0_batch(d, offsets, lengths, 65.40639132514966, 2093.004522404789)
t0 = time.perf_counter(); out = plan.f0_batch(d, offsets, lengths, 65.40639132514966, 2093.004522404789); dt = time.perf_counter() - t0
fr = int((1 + lengths // hop).sum())
print(f"AFX_F0_DEBUG={os.environ.get('AFX_F0_DEBUG','0')} n={n} {sr}/{nfft}/{hop} {dt*1e3:.1f} ms  {fr/dt/1e6:.2f} Mframes/s  quality {out['stats'][:,3].mean():.3f}")
