"""Same-box A/B of two versions of parallel.py (box-to-box variation of the end-to-end time is larger than most host-side
changes): save the other version as audio_feature_extraction_amd/_parallel_prev.py (not tracked), then
python tools/e2e_ab.py n_files  -> four interleaved rounds of process_files with and without pYIN for each version."""
import os, sys, tempfile, time, logging
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pathlib import Path
from audio_feature_extraction_amd import AudioFeatureExtractor
from audio_feature_extraction_amd.synth import make_clip
from audio_feature_extraction_amd.wavio import write_wav_pcm16
from audio_feature_extraction_amd import parallel as cur, _parallel_prev as prev
n = int(sys.argv[1]); dur = 10.0
logging.disable(logging.CRITICAL)
d = tempfile.mkdtemp(prefix="afx_e2e_")
base = [make_clip(i, 22050, dur) for i in range(16)]
for i in range(n):
    write_wav_pcm16(os.path.join(d, "clip%05d.wav" % i), np.roll(base[i % 16], 997 * i), 22050)
ex = AudioFeatureExtractor()
files = list(Path(d).glob("*.wav"))
for mod in (cur, prev): mod.process_files(ex, files)
for rep in range(4):
    for name, mod in (("prev", prev), ("cur", cur)):
        for feats in (None, ["mfcc", "energy"]):
            t0 = time.perf_counter(); out = mod.process_files(ex, files, features_to_extract=feats); dt = time.perf_counter() - t0
            print(f"{name:5s} {'all' if feats is None else 'no-f0':6s} {len(out)} files {dt*1e3:6.1f} ms  pipeline {mod.LAST_TIMING['pipeline']*1e3:6.1f} dicts {mod.LAST_TIMING['dicts']*1e3:5.1f}")
for f in os.listdir(d): os.remove(os.path.join(d, f))
os.rmdir(d)
