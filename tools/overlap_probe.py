"""Does splitting a batch over two HIP streams (two plans, two host threads) beat one stream?"""
import os, sys, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_batch
n = 1000
samples, offsets, lengths = make_batch(n, 22050, 10.0, workers=16)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctxs = [N.Context(0) for _ in range(P)]
plans = [N.Plan(c, N.make_params(22050, 1024, 256, 13)) for c in ctxs]
bufs = []
per = n // P
for i, c in enumerate(ctxs):
    lo, hi = offsets[i * per], offsets[(i + 1) * per - 1] + lengths[(i + 1) * per - 1]
    d = N.DeviceBuffer(c, int(hi - lo) * 4); d.upload(samples[lo:hi]); bufs.append((d, offsets[i * per:(i + 1) * per] - lo, lengths[i * per:(i + 1) * per]))
def run(i, steps, outs):
    d, o, l = bufs[i]
    out = None
    for _ in range(steps):
        out = plans[i].extract_batch(d, o, l, out=out)
    outs[i] = out
outs = [None] * P
for i in range(P): run(i, 3, outs)
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i, 20, outs)) for i in range(P)]
[t.start() for t in th]; [t.join() for t in th]
dt = (time.perf_counter() - t0) / 20
fr = sum(int(o["nframes"].sum()) for o in outs)
print(f"P={P}: {dt*1e3:.3f} ms per {fr} frames -> {fr/dt/1e6:.1f} Mframes/s")
