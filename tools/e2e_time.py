"""End-to-end batch_process on a directory of PCM16 WAV files (decode + MFCC/RMS + pYIN + dict building), files/s.
python tools/e2e_time.py [n_files] [seconds]   -- writes synthetic clips to a temp dir first (not timed)."""
import os, sys, tempfile, time, logging
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audio_feature_extraction_amd import AudioFeatureExtractor
from audio_feature_extraction_amd.synth import make_clip
from audio_feature_extraction_amd.wavio import write_wav_pcm16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dur = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
logging.disable(logging.CRITICAL)
d = tempfile.mkdtemp(prefix="afx_e2e_")
base = [make_clip(i, 22050, dur) for i in range(16)]
for i in range(n):
    write_wav_pcm16(os.path.join(d, "clip%05d.wav" % i), np.roll(base[i % 16], 997 * i), 22050)
ex = AudioFeatureExtractor()
ex.batch_process(d)                                    # plans, tables, first-touch
from pathlib import Path
from audio_feature_extraction_amd import parallel
wins = [int(float(w) * 1e6) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [None]
for win in wins:
  for rep in range(2):
    t0 = time.perf_counter()
    out = ex.batch_process(d) if win is None else parallel.process_files(ex, list(Path(d).glob("*.wav")), max_batch_samples=win)
    dt = time.perf_counter() - t0
    if win is not None: print(f"window {win/1e6:.0f} M samples:", end=" ")
    frames = n * (1 + int(22050 * dur) // 256)
    print(f"batch_process: {len(out)} files of {dur:.0f} s in {dt*1e3:.0f} ms = {len(out)/dt:.0f} files/s, "
          f"{frames/dt/1e6:.2f} Mframes/s end to end (decode + MFCC/RMS + pYIN), host cpus {os.cpu_count()}")
    print("  phases (s):", {k: round(v, 4) for k, v in parallel.LAST_TIMING.items() if k != "timeline"})
    if os.environ.get("AFX_E2E_TIMELINE"):
        for r in parallel.LAST_TIMING.get("timeline", []):
            print("    sub-batch of %4d clips: begin %.1f ms, uploaded %.1f, f0 done %.1f, collected %.1f" % (r[1], r[2] * 1e3, r[3] * 1e3, r[4] * 1e3, r[5] * 1e3))
# features_to_extract (README.md:141-146): the same directory without the pYIN pass
for rep in range(2):
    t0 = time.perf_counter()
    out = ex.batch_process(d, features_to_extract=["mfcc", "energy"])
    dt = time.perf_counter() - t0
    print(f"batch_process(features_to_extract=['mfcc', 'energy']): {len(out)} files in {dt*1e3:.0f} ms = {len(out)/dt:.0f} files/s "
          f"(decode + MFCC/RMS, no pYIN)")
    print("  phases (s):", {k: round(v, 4) for k, v in parallel.LAST_TIMING.items() if k != "timeline"})
    if os.environ.get("AFX_E2E_TIMELINE"):
        for r in parallel.LAST_TIMING.get("timeline", []):
            print("    sub-batch of %4d clips: begin %.1f ms, uploaded %.1f, f0 done %.1f, collected %.1f" % (r[1], r[2] * 1e3, r[3] * 1e3, r[4] * 1e3, r[5] * 1e3))
for f in os.listdir(d): os.remove(os.path.join(d, f))
os.rmdir(d)
