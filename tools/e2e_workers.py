"""End-to-end time of process_files against workers per GPU and window size: python tools/e2e_workers.py n_files"""
import os, sys, tempfile, time, logging
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pathlib import Path
from audio_feature_extraction_amd import AudioFeatureExtractor, parallel
from audio_feature_extraction_amd.synth import make_clip
from audio_feature_extraction_amd.wavio import write_wav_pcm16
n = int(sys.argv[1]); dur = 10.0
logging.disable(logging.CRITICAL)
d = tempfile.mkdtemp(prefix="afx_e2e_")
base = [make_clip(i, 22050, dur) for i in range(16)]
for i in range(n):
    write_wav_pcm16(os.path.join(d, "clip%05d.wav" % i), np.roll(base[i % 16], 997 * i), 22050)
ex = AudioFeatureExtractor()
files = list(Path(d).glob("*.wav"))
for w in (1, 2, 3, 4):
    parallel.process_files(ex, files, workers_per_gpu=w)
for rep in range(3):
    for w, win in ((1, 80), (2, 80), (3, 80), (4, 80), (2, 160), (3, 48)):
        t0 = time.perf_counter(); out = parallel.process_files(ex, files, workers_per_gpu=w, max_batch_samples=win * 1024 * 1024); dt = time.perf_counter() - t0
        print(f"workers {w} window {win:3d} M: {len(out)} files {dt*1e3:6.1f} ms  pipeline {parallel.LAST_TIMING['pipeline']*1e3:6.1f}")
for f in os.listdir(d): os.remove(os.path.join(d, f))
os.rmdir(d)
