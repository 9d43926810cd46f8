"""How many files per second the HOST side of batch_process sustains when the GPUs cost nothing: parallel.process_files over
N fake devices (plans whose passes return at once) with the REAL ingest -- glob, size-balanced sharding, afx_wav_probe /
afx_wav_read_s16 by native threads into packed int16 windows, the pipeline of windows, dict building.  The only 8-GPU
evidence obtainable without the node: the host must sustain >= N x the 1-GPU end-to-end rate or it is the bottleneck
(reference: 04_feature_extraction_experiment/feature_extraction_for_student.py:168-174, a multiprocessing.Pool over files).

    python tools/host_ceiling.py [n_files=2048] [seconds=10] [devices=8] [--touch]

--touch makes the fake upload read every sample once (np.sum over the window), a stand-in for the PCIe copy's host-side
read.  Needs no GPU; uses libafx.so's host-only entry points.  Files go to a temp dir (tmpfs when /dev/shm has room)."""
import logging
import os
import sys
import tempfile
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from audio_feature_extraction_amd import parallel  # noqa: E402
from audio_feature_extraction_amd.core.feature_extractor import AudioFeatureExtractor  # noqa: E402
from audio_feature_extraction_amd.hostinfo import usable_cpus  # noqa: E402
from audio_feature_extraction_amd.synth import make_clip  # noqa: E402
from audio_feature_extraction_amd.wavio import write_wav_pcm16  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
touch = "--touch" in sys.argv
n = int(args[0]) if len(args) > 0 else 2048
dur = float(args[1]) if len(args) > 1 else 10.0
ndev = int(args[2]) if len(args) > 2 else 8
K = 13


class FakeBuf:
    def upload(self, arr):
        if touch:
            self.s = int(np.asarray(arr).view(np.uint8)[::4096].sum())     # one read per page

    def free(self):
        pass


class FakePinned:
    """Stands in for _native.PinnedBuffer: a block that is allocated and touched once and then reused by the worker's pool, as
    page-locked memory is -- a fresh np.empty per window would charge the reader threads a page fault per 4 KB."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._a = np.zeros(self.nbytes, np.uint8)

    def array(self, dtype, count):
        return self._a[: count * np.dtype(dtype).itemsize].view(dtype)

    def free(self):
        self._a = None


class FakePlan:
    def device_buffer(self, nbytes):
        return FakeBuf()

    def pinned_buffer(self, nbytes):
        return FakePinned(nbytes)

    def extract_submit(self, dbuf, offs, lens, flags=0, fmt=0):
        self.n = len(offs)
        self.lens = np.asarray(lens)

    def extract_collect(self):
        return {"stats": np.zeros((self.n, 4 * K + 3), np.float32), "status": np.zeros(self.n, np.int32),
                "nframes": (1 + self.lens // 256).astype(np.int32)}

    def f0_batch(self, dbuf, offs, lens, fmin, fmax, flags=0, fmt=0):
        return {"stats": np.zeros((len(offs), 4)), "status": np.zeros(len(offs), np.int32)}


logging.disable(logging.CRITICAL)
ex = AudioFeatureExtractor.__new__(AudioFeatureExtractor)
ex.sr, ex.n_mfcc, ex.f0_min, ex.f0_max = 22050, K, 65.4, 2093.0
ex.logger = logging.getLogger("host_ceiling")
plans = {}
ex._devices = lambda: list(range(ndev))
ex._plan = lambda device=None, lane=0: plans.setdefault((device, lane), FakePlan())

shm = "/dev/shm"
need = n * int(22050 * dur) * 2
base_dir = shm if os.path.isdir(shm) and os.statvfs(shm).f_bavail * os.statvfs(shm).f_frsize > 2 * need else None
d = tempfile.mkdtemp(prefix="afx_ceiling_", dir=base_dir)
try:
    base = [make_clip(i, 22050, dur) for i in range(8)]
    for i in range(n):
        write_wav_pcm16(os.path.join(d, "clip%05d.wav" % i), np.roll(base[i % 8], 997 * i), 22050)
    from pathlib import Path
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        files = list(Path(d).glob("*.wav"))
        out = parallel.process_files(ex, files)
        dt = time.perf_counter() - t0
        assert len(out) == n
        best = dt if best is None else min(best, dt)
        print(f"rep {rep}: {n} files x {dur:.0f} s over {ndev} fake devices in {dt * 1e3:.0f} ms = {n / dt:.0f} files/s "
              f"({n * int(22050 * dur) * 2 / dt / 1e9:.2f} GB/s of PCM16); phases (s): "
              f"{ {k: round(v, 3) for k, v in parallel.LAST_TIMING.items() if k != 'timeline'} }")
    print(f"host ceiling: {n / best:.0f} files/s with {usable_cpus()} usable CPUs (os.cpu_count() = {os.cpu_count()}), "
          f"files on {'tmpfs' if base_dir else 'disk'}, touch = {touch}")
finally:
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)
