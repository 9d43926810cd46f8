"""Embarrassingly parallel file sharding for ``batch_process``.

Clips are independent (every cross-frame coupling -- trim max, top_db max, delta
stencil, mean/std -- is intra-clip), so N GPUs = N independent shards and the only
"collective" is a host-side gather of 4*n_mfcc+3 floats per file.  No RCCL, no xGMI
traffic.  Two deployment shapes share the partition/gather helpers here:

* in-process: one worker thread per visible GPU, each with its own afx context,
  plan and stream (``process_files``; what ``AudioFeatureExtractor.batch_process`` uses);
* one process per GPU under ``torch.distributed`` (``shard_range`` + ``gather_shards``;
  what ``bench.py --gpus N`` uses) -- the reference's only parallel precedent is a
  ``multiprocessing.Pool`` over files
  (04_feature_extraction_experiment/feature_extraction_for_student.py:168-174).
"""
from __future__ import annotations

import queue
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Dict, List, Sequence, Tuple

import numpy as np

from . import _native, wavio
from .hostinfo import usable_cpus


def lpt_partition(lengths: Sequence[int], n_parts: int) -> List[List[int]]:
    """Longest-processing-time-first: sort by length (desc), give each clip to the least
    loaded part.  Equal lengths degenerate to a round-robin deal.  Every part keeps its
    indices in ascending order."""
    n_parts = max(1, int(n_parts))
    parts: List[List[int]] = [[] for _ in range(n_parts)]
    load = [0] * n_parts
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    for i in order:
        p = min(range(n_parts), key=lambda q: (load[q], q))
        parts[p].append(i)
        load[p] += int(lengths[i]) + 1
    for p in parts:
        p.sort()
    return parts


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced shard [lo, hi) of rank ``rank`` out of ``world``."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_shards(local: np.ndarray, n_total: int, rank: int, world: int, group=None) -> np.ndarray:
    """All ranks contribute the rows of their ``shard_range`` shard; every rank gets the
    [n_total, ...] array in global order.  Host-side gather (gloo or nccl-backed object
    gather); the data path itself has no collective."""
    if world == 1:
        return np.asarray(local)
    import torch.distributed as dist
    parts: List[Any] = [None] * world
    dist.all_gather_object(parts, np.asarray(local), group=group)
    out = np.concatenate([np.asarray(p) for p in parts], axis=0)
    if out.shape[0] != n_total:
        raise RuntimeError(f"gathered {out.shape[0]} rows, expected {n_total}")
    return out


def _decode(path: str, sr: int):
    """-> ('s16'|'f32', mono array) ; PCM16 mono files at the target rate stay int16 so that
    the upload is 2 bytes/sample and the /32768 happens on the GPU (bit-identical)."""
    a, rate, kind = wavio.read_wav_raw(path)
    if kind == "s16" and a.shape[1] == 1 and rate == sr:
        return "s16", np.ascontiguousarray(a[:, 0])
    y = wavio.to_mono(wavio.to_float32(a, kind))
    if rate != sr:
        y = wavio.resample(y, rate, sr)
    return "f32", y


def _pack(clips: List[np.ndarray], dtype) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Packs clips with 4-element alignment (enables the kernels' 16-byte loads)."""
    lengths = np.array([c.size for c in clips], np.int64)
    padded = (lengths + 3) // 4 * 4
    offsets = np.zeros(len(clips), np.int64)
    if len(clips):
        offsets[1:] = np.cumsum(padded)[:-1]
    buf = np.zeros(int(padded.sum()), dtype)
    for c, o in zip(clips, offsets):
        buf[o:o + c.size] = c
    return buf, offsets, lengths


class _PinPool:
    """Page-locked window buffers of one (device, lane) worker, kept across windows and calls: the native reader fills
    them, the upload is one DMA.  At most two are out at a time (window k on the device, k + 1 being read)."""

    def __init__(self, plan):
        self.plan, self.free, self.lock = plan, [], threading.Lock()

    def get(self, nbytes: int):
        with self.lock:
            for i, b in enumerate(self.free):
                if b.nbytes >= nbytes:
                    return self.free.pop(i)
            for b in self.free:                      # too small: give the memory back before taking more
                b.free()
            self.free.clear()
        return self.plan.pinned_buffer(max(int(nbytes * 1.25), 1 << 20))

    def put(self, b):
        with self.lock:
            self.free.append(b)


class _DevPool:
    """The HBM window buffer of one (device, lane) worker, kept across windows and calls (a worker has one window on the
    device at a time).  hipMalloc / hipFree per window cost more than their own time: a free synchronises the whole device,
    i.e. also the passes the other workers have in flight."""

    def __init__(self, plan):
        self.plan, self.buf, self.lock = plan, None, threading.Lock()

    def get(self, nbytes: int):
        with self.lock:
            b, self.buf = self.buf, None
        if b is not None and getattr(b, "nbytes", 0) >= nbytes:
            return b
        if b is not None:
            b.free()
        return self.plan.device_buffer(max(int(nbytes * 1.25), 16))

    def put(self, b):
        with self.lock:
            old, self.buf = self.buf, b
        if old is not None:
            old.free()


_switch_lock = threading.Lock()
_switch_state = [0, 0.0]               # process_files calls in flight, the interpreter's own switch interval


def _fast_switch(on: bool):
    """The interpreter's thread switch interval at 0.2 ms while any process_files call runs (first in saves it, last out
    restores it: calls may overlap)."""
    with _switch_lock:
        if on:
            if _switch_state[0] == 0:
                _switch_state[1] = sys.getswitchinterval()
                sys.setswitchinterval(2e-4)
            _switch_state[0] += 1
        else:
            _switch_state[0] -= 1
            if _switch_state[0] == 0:
                sys.setswitchinterval(_switch_state[1])


LAST_TIMING: Dict[str, Any] = {}     # seconds per phase of the last process_files call (developer aid)
WORKERS_PER_GPU = 3     # sub-batches in flight per GPU (own context / stream / thread): copies, the bandwidth-bound
                        # kernels and the host round trip of one hide under the frame kernel of another
DECODE_THREADS_PER_GPU = 16


def _windows(sizes: Sequence[int], idxs: Sequence[int], budget: int) -> List[List[int]]:
    """Cuts a lane's files (in order) into windows whose estimated sample count stays under ``budget``;
    a single file larger than the budget is a window of its own.  The windows are of even size -- as many as the budget
    demands, each an even share of the lane's samples: a short last window costs a whole device pass (the pYIN pass of 30
    clips takes as long as that of 300: its Viterbi kernel is 862 dependent steps whatever the number of clips)."""
    total = sum(int(sizes[i]) for i in idxs)
    nwin = max(1, -(-total // max(int(budget), 1)))
    share = total / nwin
    out: List[List[int]] = []
    cur: List[int] = []
    tot = done = 0
    for i in idxs:
        sz = int(sizes[i])
        if cur and (tot + sz > budget or (done + tot >= share * (len(out) + 1) and len(out) + 1 < nwin)):
            out.append(cur)
            done += tot
            cur, tot = [], 0
        cur.append(i)
        tot += sz
    if cur:
        out.append(cur)
    return out


FEATURE_GROUPS = ("f0", "mfcc", "energy")        # README.md:141-146 (features_to_extract); dict order is this order


def normalize_features(features_to_extract) -> Tuple[str, ...]:
    """None -> all three groups (the reference's behaviour); otherwise the requested subset, validated, in dict order."""
    if features_to_extract is None:
        return FEATURE_GROUPS
    if isinstance(features_to_extract, str):
        features_to_extract = [features_to_extract]
    req = [str(f) for f in features_to_extract]
    bad = [f for f in req if f not in FEATURE_GROUPS]
    if bad:
        raise ValueError(f"features_to_extract: unknown feature group(s) {bad}; choose from {list(FEATURE_GROUPS)}")
    if not req:
        raise ValueError(f"features_to_extract is empty; choose from {list(FEATURE_GROUPS)}")
    return tuple(g for g in FEATURE_GROUPS if g in req)


def process_files(extractor, files: Sequence, max_batch_samples: int = 80 * 1024 * 1024,
                  workers_per_gpu: int = WORKERS_PER_GPU, features_to_extract=None) -> List[Dict[str, Any]]:
    """Shard over GPUs (and over a few workers per GPU) -> per worker a pipeline of bounded windows:
    decode window k + 1 on the shared host pool while window k is packed, uploaded and extracted ->
    dicts in input (glob) order.  Host memory holds at most two windows per worker, not the directory.

    On the device a worker owns two plans (own stream each): the MFCC / RMS pass of a sub-batch is queued with
    afx_extract_submit on the first and collected only after the pYIN pass of the same sub-batch (afx_f0_batch, second
    plan) has run beside it -- one upload serves both.  ``features_to_extract`` (README.md:141-146) leaves out the passes
    nobody asked for: without 'f0' no pYIN pass runs (it is ~30x the MFCC pass).

    Error behaviour is the reference's (feature_extractor.py:229-235): a file that cannot be loaded, a clip the
    kernels reject, or a device-level failure while its window is processed is logged and left out; the batch goes on."""
    import os
    log = extractor.logger
    n = len(files)
    if n == 0:
        return []
    devices = extractor._devices()
    t_start = time.perf_counter()
    errors: List[Any] = [None] * n
    K = extractor.n_mfcc
    stats = np.zeros((n, 4 * K + 3), np.float32)
    status = np.full(n, -1, np.int32)          # -1: not extracted (yet)
    nframes = np.zeros(n, np.int32)
    f0s = np.zeros((n, 4), np.float64)
    f0_done = np.zeros(n, bool)
    nsamp = np.zeros(n, np.int64)
    lanes = [(d, w) for d in devices for w in range(max(1, int(workers_per_gpu)))]
    if n < 4 * len(lanes):                             # small jobs: one worker per GPU
        lanes = [(d, 0) for d in devices]
    # shard by file size (a proxy for clip length that needs no decode): longest-processing-time-first
    def fsize(f):
        try:
            return max(1, os.path.getsize(str(f)) // 2)
        except OSError:
            return 1
    sizes = [fsize(f) for f in files]
    parts = lpt_partition(sizes, len(lanes))
    flags = _native.FLAG_PREEMPH | _native.FLAG_TRIM
    want = normalize_features(features_to_extract)
    want_f0, want_stats = "f0" in want, ("mfcc" in want or "energy" in want)
    host_cpus = usable_cpus()                   # the job's share of the host, not os.cpu_count()
    pool = ThreadPoolExecutor(max(1, min(host_cpus, DECODE_THREADS_PER_GPU * len(devices), n)))
    phase = {"decode_wait": 0.0, "device": 0.0}
    phase_lock = threading.Lock()

    def dec(i):
        try:
            return _decode(str(files[i]), extractor.sr)
        except Exception as e:          # load_audio: log + the file is dropped
            log.error(f"載入音頻文件失敗: {str(e)}")
            errors[i] = e
            return None

    # Files that need no conversion -- 16-bit PCM, mono, at the target rate: what a corpus of speech clips is -- never
    # pass through Python: libafx parses their headers and reads their samples straight into the packed int16 batch
    # buffer with native threads (afx_wav_probe / afx_wav_read_s16).  Everything else, and any file the native reader
    # cannot open or parse, goes through wavio as before (which also produces the error a bad file is logged with).
    native_threads = max(1, min(host_cpus, DECODE_THREADS_PER_GPU * len(devices)) // max(1, min(len(lanes), 4)))
    win_pool = ThreadPoolExecutor(max(1, len(lanes)))

    def load_window(win, pin=None):
        """-> (packed 16-bit group or None, indices decoded by wavio, their decoded clips); pin: the worker's pool of
        page-locked buffers (None: ordinary memory)"""
        rest = list(win)
        packed = None
        held = None
        try:
            paths = [str(files[i]) for i in win]
            pr = _native.wav_probe(paths, native_threads)
            ok = ((pr["status"] == 0) & (pr["tag"] == 1) & (pr["bits"] == 16) & (pr["channels"] == 1) &
                  (pr["rate"] == extractor.sr))
            sel = np.nonzero(ok)[0]
            if sel.size:
                lens = pr["frames"][sel].astype(np.int64)
                padded = (lens + 3) // 4 * 4                      # 4-element alignment, as _pack
                offs = np.zeros(sel.size, np.int64)
                offs[1:] = np.cumsum(padded)[:-1]
                nel = max(int(padded.sum()), 1)
                if pin is not None:
                    held = pin.get(nel * 2)
                    buf = held.array(np.int16, nel)
                else:
                    buf = np.empty(nel, np.int16)
                st = _native.wav_read_s16([paths[j] for j in sel], pr["data_off"][sel], lens, buf, offs, native_threads)
                for o, ln, pd in zip(offs[padded > lens], lens[padded > lens], padded[padded > lens]):
                    buf[o + ln: o + pd] = 0
                good = st == 0
                if good.any():
                    packed = ([win[j] for j in sel[good]], buf, offs[good], lens[good], held)
                    held = None
                    taken = set(packed[0])
                    rest = [i for i in win if i not in taken]
        except Exception:                                         # no native reader: the Python decoder takes the window
            rest, packed = list(win), None
        if held is not None:
            pin.put(held)
        decoded = list(pool.map(dec, rest)) if rest else []
        return packed, rest, decoded

    timeline: List[Any] = []                                      # (lane, clips, t_begin, t_uploaded, t_f0_done, t_collected) per sub-batch
    finished: Any = queue.SimpleQueue()                          # index lists of sub-batches whose results are in the arrays
    recs: List[Any] = [None] * n

    def run_group(plans, cur, buf, offs, lens, fmt, dev):
        plan, plan_f0 = plans
        tl = [None, len(cur), time.perf_counter() - t_start, 0.0, 0.0, 0.0]
        dbuf = dev.get(max(buf.nbytes, 16))                       # one PCIe copy for both passes
        try:
            dbuf.upload(buf)
            tl[3] = time.perf_counter() - t_start
            submitted = False
            if want_stats:                                        # MFCC / RMS pass: queued, runs beside the pYIN pass below
                plan.extract_submit(dbuf, offs, lens, flags=flags, fmt=fmt)
                submitted = True
            out = f0 = None
            try:
                if want_f0:
                    f0 = plan_f0.f0_batch(dbuf, offs, lens, extractor.f0_min, extractor.f0_max, flags=flags, fmt=fmt)
                tl[4] = time.perf_counter() - t_start
            finally:
                if submitted:                                     # always: the buffer must outlive the queued pass
                    out = plan.extract_collect()
            tl[5] = time.perf_counter() - t_start
            timeline.append(tl)
            nsamp[cur] = lens
            if out is not None:
                stats[cur] = out["stats"]
                nframes[cur] = out["nframes"]
            if f0 is not None:
                f0s[cur] = f0["stats"]
            f0_done[cur] = True
            # last: a file counts only with every requested pass done.  Without the MFCC / RMS pass the f0 pass's own
            # status (non-finite input) decides
            status[cur] = out["status"] if out is not None else f0["status"]
            finished.put(list(cur))
        finally:
            dev.put(dbuf)                                         # both passes are through (collect above): reusable

    def worker(lane, idxs):
        wins = _windows(sizes, idxs, max_batch_samples)
        plan = pin = dev = None
        try:                                                      # (MFCC / RMS plan, pYIN plan): own context and stream each
            plan = (extractor._plan(lane[0], lane[1]),
                    extractor._plan(lane[0], (lane[1], "f0")) if want_f0 else None)
            dpools = extractor.__dict__.setdefault("_dev_pools", {})
            dev = dpools.get(lane) or dpools.setdefault(lane, _DevPool(plan[0]))
            if hasattr(plan[0], "pinned_buffer"):                 # page-locked window buffers, kept with the extractor
                pools = extractor.__dict__.setdefault("_pin_pools", {})
                pin = pools.get(lane) or pools.setdefault(lane, _PinPool(plan[0]))
        except Exception as e:                                    # no device for this lane: its files are dropped, the batch goes on
            for i in idxs:
                errors[i] = e
            return
        pending = win_pool.submit(load_window, wins[0], pin) if wins else None
        for k, win in enumerate(wins):
            t0 = time.perf_counter()
            packed, rest, decoded = pending.result()
            # next window loads while this one is on the device
            pending = win_pool.submit(load_window, wins[k + 1], pin) if k + 1 < len(wins) else None
            t1 = time.perf_counter()
            try:
                if packed is not None:                            # the natively packed 16-bit clips, in budget-sized runs
                    ids, buf, offs, lens, _held = packed
                    pos = 0
                    while pos < len(ids):
                        tot, end = 0, pos
                        while end < len(ids) and (end == pos or tot + int(lens[end]) <= max_batch_samples):
                            tot += int(lens[end])
                            end += 1
                        lo = int(offs[pos])
                        hi = int(offs[end - 1] + (lens[end - 1] + 3) // 4 * 4)
                        run_group(plan, ids[pos:end], buf[lo:hi], offs[pos:end] - lo, lens[pos:end], _native.FMT_S16, dev)
                        pos = end
                for kind, fmt, dt in (("s16", _native.FMT_S16, np.int16), ("f32", _native.FMT_F32, np.float32)):
                    sel = [(i, d[1]) for i, d in zip(rest, decoded) if d is not None and d[0] == kind]
                    pos = 0
                    while pos < len(sel):          # a window may still exceed the budget (sizes were estimates)
                        tot, end = 0, pos
                        while end < len(sel) and (end == pos or tot + sel[end][1].size <= max_batch_samples):
                            tot += sel[end][1].size
                            end += 1
                        buf, offs, lens = _pack([y for _, y in sel[pos:end]], dt)
                        run_group(plan, [i for i, _ in sel[pos:end]], buf, offs, lens, fmt, dev)
                        pos = end
            except Exception as e:          # a device-level failure drops the files of the sub-batch it hit, and the
                for i in win:               # rest of this window; later windows are still attempted
                    if errors[i] is None and not (status[i] >= 0 and f0_done[i]):
                        errors[i] = e
                        status[i] = -1
            if packed is not None and packed[4] is not None:
                pin.put(packed[4])
            del decoded, packed
            with phase_lock:
                phase["decode_wait"] += t1 - t0
                phase["device"] += time.perf_counter() - t1

    threads = [threading.Thread(target=worker, args=(ln, p)) for ln, p in zip(lanes, parts) if p]
    # The calling thread builds the result dicts of finished sub-batches while the workers drive the device (55 Python floats
    # and four lists per file: 35-50 ms for 8192 files if left to the end).  The workers need the interpreter only for
    # microseconds between two native calls, but would wait a whole switch interval (5 ms) for it: shortened for the duration.
    def delivered(i):
        # fewer than nine frames fails the MFCC group only (the width-9 delta); extract_energy has its statistics
        return status[i] == _native.CLIP_OK or ("mfcc" not in want and status[i] == _native.CLIP_TOO_SHORT and
                                                 nsamp[i] >= 2 and nframes[i] >= 1)

    from .core.feature_extractor import AudioFeatureExtractor as _Stock
    stock = (getattr(type(extractor), "_stats_to_dicts", None) is _Stock._stats_to_dicts and
             getattr(type(extractor), "_f0_to_dict", None) is _Stock._f0_to_dict)      # else: the extractor's own methods, at the end
    K4 = 4 * K

    def build(i):                                                     # what _stats_to_dicts / _f0_to_dict build, key for key
        row, rec = stats[i].tolist(), {"file_path": str(files[i])}
        if want_f0:
            q = f0s[i].tolist()
            rec["f0_mean"], rec["f0_std"], rec["f0_missing_rate"], rec["f0_quality"] = q[0], q[1], q[2], q[3]
        if "mfcc" in want:
            rec["mfcc_mean"], rec["mfcc_std"] = row[0:K], row[K:2 * K]
            rec["mfcc_delta_mean"], rec["mfcc_delta2_mean"] = row[2 * K:3 * K], row[3 * K:K4]
        if "energy" in want:
            rec["energy_mean"], rec["energy_std"], rec["energy_range"] = row[K4], row[K4 + 1], row[K4 + 2]
        return rec

    def drain(block):
        try:
            cur = finished.get(timeout=0.002) if block else finished.get_nowait()
        except queue.Empty:
            return False
        if stock:
            for i in cur:
                if errors[i] is None and delivered(i):
                    recs[i] = build(i)
        return True

    _fast_switch(True)
    try:
        for t in threads:
            t.start()
        while any(t.is_alive() for t in threads):
            drain(True)
        while drain(False):
            pass
        for t in threads:
            t.join()
    finally:
        _fast_switch(False)
    win_pool.shutdown()
    pool.shutdown()

    t_gpu = time.perf_counter()
    results: List[Dict[str, Any]] = []
    from .core.feature_extractor import _status_error
    for i, f in enumerate(files):
        name = getattr(f, "name", str(f))
        err = errors[i]
        if err is None and not delivered(i):
            err = _status_error(int(status[i]), "extract_features", int(nframes[i]))
            log.error(f"特徵提取失敗: {str(err)}")
        if err is not None:
            log.error(f"處理文件 {name} 失敗: {str(err)}")
            continue
        if recs[i] is not None:
            results.append(recs[i])
            log.info(f"成功處理文件: {name}")
            continue
        mfcc, energy = extractor._stats_to_dicts(stats[i])
        rec: Dict[str, Any] = {"file_path": str(f)}
        if want_f0:
            rec.update(extractor._f0_to_dict(f0s[i]))
        if "mfcc" in want:
            rec.update(mfcc)
        if "energy" in want:
            rec.update(energy)
        results.append(rec)
        log.info(f"成功處理文件: {name}")
    LAST_TIMING.update(pipeline=t_gpu - t_start, decode_wait=phase["decode_wait"], device=phase["device"],
                       dicts=time.perf_counter() - t_gpu, files=n, workers=len(threads), timeline=sorted(timeline, key=lambda r: r[2]))
    return results
