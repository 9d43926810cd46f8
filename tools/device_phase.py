"""Where a batch_process worker's device phase goes: one window of 16-bit clips (default 683 x 10 s @22050 Hz):
python tools/device_phase.py [clips]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audio_feature_extraction_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 683
L = 220500
lens = np.full(n, L, np.int64)
offs = np.arange(n, dtype=np.int64) * ((L + 3) // 4 * 4)
rng = np.random.default_rng(0)
buf = (rng.standard_normal(int(offs[-1] + L + 4)) * 3000).astype(np.int16)
ctx = N.Context(0)
plan = N.Plan(ctx, N.make_params(22050, 1024, 256, 13))
for rep in range(3):
    t = [time.perf_counter()]
    d = plan.device_buffer(buf.nbytes); t.append(time.perf_counter())
    d.upload(buf); t.append(time.perf_counter())
    out = plan.extract_batch(d, offs, lens, fmt=N.FMT_S16); t.append(time.perf_counter())
    f0 = plan.f0_batch(d, offs, lens, 65.40639132514966, 2093.004522404789, fmt=N.FMT_S16); t.append(time.perf_counter())
    d.free(); t.append(time.perf_counter())
    names = ("alloc", "upload", "mfcc", "f0", "free")
    print(f"{n} clips, {buf.nbytes / 1e6:.0f} MB: " + "  ".join(f"{k} {1e3 * (b - a):.1f} ms" for k, a, b in zip(names, t, t[1:])))
