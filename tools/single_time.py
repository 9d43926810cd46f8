"""Latency of extract_features on one 5 s PCM16 WAV (BASELINE configs[0]): python tools/single_time.py"""
import os, sys, tempfile, time, logging
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audio_feature_extraction_amd import AudioFeatureExtractor
from audio_feature_extraction_amd.synth import make_clip
from audio_feature_extraction_amd.wavio import write_wav_pcm16
logging.disable(logging.CRITICAL)
d = tempfile.mkdtemp(prefix="afx_one_")
path = os.path.join(d, "clip.wav")
write_wav_pcm16(path, make_clip(3, 22050, 5.0, speechy=True), 22050)
ex = AudioFeatureExtractor()
ex.extract_features(path)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); out = ex.extract_features(path); ts.append(time.perf_counter() - t0)
y, _ = ex.load_audio(path); yp = ex.preprocess_audio(y)
t0 = time.perf_counter(); ex.extract_mfcc(yp); t1 = time.perf_counter(); ex.extract_energy(yp); t2 = time.perf_counter(); ex.extract_f0(yp); t3 = time.perf_counter()
print(f"extract_features(5 s clip): median {np.median(ts)*1e3:.2f} ms, min {min(ts)*1e3:.2f} ms; staged: mfcc {(t1-t0)*1e3:.2f} energy {(t2-t1)*1e3:.2f} f0 {(t3-t2)*1e3:.2f} ms; f0_mean {out['f0_mean']:.2f}")
os.remove(path); os.rmdir(d)
