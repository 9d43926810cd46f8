#!/usr/bin/env python3
"""bench.py -- MFCC frames/sec of the fused HIP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {2,3,5}]

Workload (default BASELINE.json configs[1]; configs[3] = the same per GPU when N > 1): per GPU, 1000 synthetic
10 s clips @22050 Hz, frame_length=1024, hop_length=256, n_mfcc=13 (SURVEY.md 8(d) generator), already resident in
HBM when the timed region starts.  --config 3 / 5 run BASELINE.json configs[2] / configs[4] (16 kHz 512/128/40 and
44.1 kHz 2048/512/20) the same way; they are parity cases with their own roofline lines, not the headline metric.
One "step" = one pass of preprocess_audio -> extract_mfcc + extract_energy over that batch (frame kernel before the trim
decision, trim decision, redo launch, then k_tail: clamp + DCT + statistics per clip -- timed in the `dct` slot; 4*n_mfcc+3
floats per clip land in pinned host memory).
Scaling is weak: every rank owns its own 1000 clips, no data-path collective; the only torch.distributed traffic is
the timing barrier and the max-over-ranks reduction.  `python bench.py --gpus N` without a launcher starts the N ranks
itself (children, before anything touches the GPU).
Within a rank, by default two whole-batch steps are in flight (--inflight 2): two plans, each on its own stream
(--queue own), driven from one host thread through afx_extract_submit / afx_extract_collect -- the frame kernels of
consecutive steps run back to back, the small kernels of two steps side by side, and the host's share of a step falls
under the other step's kernels.  --inflight 1 is one afx_extract_batch call at a time; --streams S > 1 is the older
form (the clips cut into S runs, one context / stream / host thread each -- how batch_process drives a GPU).
A step is always one pass over all of the rank's clips.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- the frame kernel (dominant): algorithmic bytes = 4*hop per frame (each input sample read once) /
                  average launch duration.  Durations come from HIP events recorded on the kernel's own stream around
                  every launch of the timed region; with several streams in flight the launches overlap, so the time
                  the GPU spends in the kernel is the UNION of their intervals (common device clock): avg_launch_ms =
                  union / launches, and kernel time per step <= ms_per_step by construction (with two steps in flight
                  on their own streams the second one's interval starts while the first still holds every CU, so the
                  union contains the hand-over).  `exclusive` times every kernel of 20 calls made one at a time on a
                  fresh context (what profiles/r02_*_kernel_stats.csv shows);
                  `fp32` the secondary vector-FLOP roofline;
  distinct_batches_frames_per_s -- the timed region repeated over --distinct different ragged batches (every submit uploads
                  new clip records; the device rebuilds its block list): what a window pipeline over real files sees;
  cpu_baseline -- the numpy/scipy oracle (a port of the reference's librosa path) timed on this box's host cores over a
                  bounded sample of the same workload: one core, and a pool over every core this process may use.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs (index) -> parameters; kflop = algorithmic FLOP per frame
CONFIGS = {
    2: dict(sr=22050, n_fft=1024, hop=256, n_mfcc=13, n_mels=128, kflop=36.0, baseline_index=1),
    3: dict(sr=16000, n_fft=512, hop=128, n_mfcc=40, n_mels=128, kflop=24.0, baseline_index=2),
    5: dict(sr=44100, n_fft=2048, hop=512, n_mfcc=20, n_mels=128, kflop=75.0, baseline_index=4),
}
SECONDS = 10.0
CLIPS_PER_GPU = 1000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 (vector)

_POOL_CLIPS = None
_POOL_CFG = None


def _pool_init(first: int, per_worker: int, cfg: dict) -> None:
    """Each pool worker pre-generates its own clips so that generation is not timed."""
    global _POOL_CLIPS, _POOL_CFG
    import multiprocessing as mp
    from audio_feature_extraction_amd.synth import make_clip
    from oracle import cpu_ref
    _POOL_CFG = cfg
    ident = (mp.current_process()._identity or (1,))[0]
    _POOL_CLIPS = [make_clip(first + (ident * 131 + i) % 1000, cfg["sr"], SECONDS) for i in range(per_worker)]
    cpu_ref.extract_stats(_POOL_CLIPS[0], sr=cfg["sr"], frame_length=cfg["n_fft"], hop_length=cfg["hop"], n_mfcc=cfg["n_mfcc"])   # warm


def _pool_job(k: int) -> int:
    from oracle import cpu_ref
    c = _POOL_CFG
    y = _POOL_CLIPS[k % len(_POOL_CLIPS)]
    out = cpu_ref.extract_stats(y, sr=c["sr"], frame_length=c["n_fft"], hop_length=c["hop"], n_mfcc=c["n_mfcc"])
    return 1 + (out["trim"][1] - out["trim"][0]) // c["hop"]


from audio_feature_extraction_amd.hostinfo import usable_cpus  # noqa: E402  (affinity mask and cgroup quota)


def cpu_baseline(samples, offsets, lengths, cfg: dict, n_single: int, pool_seconds: float) -> dict:
    """Times the CPU oracle on clips of the GPU workload itself: one core (BLAS pinned to one thread) over the first
    n_single clips, then a spawn-pool over every core this process may run on but one -- the reference's only parallel
    harness is a multiprocessing.Pool(cpu_count() - 1) over files
    (04_feature_extraction_experiment/feature_extraction_for_student.py:168-174)."""
    from oracle import cpu_ref
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    kw = dict(sr=cfg["sr"], frame_length=cfg["n_fft"], hop_length=cfg["hop"], n_mfcc=cfg["n_mfcc"])
    n_single = min(n_single, len(offsets))
    clips = [samples[offsets[i]: offsets[i] + lengths[i]] for i in range(n_single)]
    frames = 0

    def run():
        nonlocal frames
        frames = 0
        for y in clips:
            out = cpu_ref.extract_stats(y, **kw)
            frames += 1 + (out["trim"][1] - out["trim"][0]) // cfg["hop"]

    cpu_ref.extract_stats(clips[0], **kw)  # warm
    t0 = time.perf_counter()
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            run()
    else:
        run()
    dt = time.perf_counter() - t0
    res = {
        "value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"first {n_single} of the {len(offsets)} clips of this workload (10 s @{cfg['sr']} Hz, "
                  f"{cfg['n_fft']}/{cfg['hop']}/{cfg['n_mfcc']}), numpy/scipy oracle (pre-emphasis, trim, MFCC + deltas, RMS, "
                  f"statistics; no file load, no pYIN), single thread, {dt:.1f} s",
        "host_cpus": os.cpu_count(), "usable_cpus": usable_cpus(),
    }
    ncpu = usable_cpus()
    if pool_seconds > 0 and ncpu > 2:
        import multiprocessing as mp
        workers = max(1, ncpu - 1)
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
        per_clip = dt / max(n_single, 1)
        n_pool = int(max(workers * 4, min(workers * 4096, pool_seconds * workers / max(per_clip, 1e-4))))
        with mp.get_context("spawn").Pool(workers, initializer=_pool_init, initargs=(0, 4, cfg)) as pool:
            pool.map(_pool_job, range(workers))                 # make sure every worker is up
            t0 = time.perf_counter()
            fr = sum(pool.map(_pool_job, range(n_pool), chunksize=2))
            dt2 = time.perf_counter() - t0
        res["all_cores"] = {"value": fr / dt2, "unit": "frames/s", "cores": workers,
                            "sample": f"{n_pool} clip passes over a {workers}-process spawn pool "
                                      f"(every core this job may use but one: {ncpu} by affinity mask and cgroup quota; "
                                      f"os.cpu_count() = {os.cpu_count()}), {dt2:.1f} s"}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    res["cpu_model"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return res


def union_ms(spans) -> float:
    """Total length of the union of (start, end) intervals."""
    tot, cur_s, cur_e = 0.0, None, None
    for s, e in sorted(spans):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (this process has not touched the GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    return max(abs(rc) for rc in rcs)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (1-based as in BASELINE.md 4): 2 = the headline metric")
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU, help="clips per GPU (default: the BASELINE config)")
    ap.add_argument("--cpu-clips", type=int, default=1000, help="clips timed on one CPU core (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-pool-seconds", type=float, default=6.0, help="wall-clock length of the all-core pool sample (0 = skip)")
    ap.add_argument("--distinct", type=int, default=5,
                    help="after the contract's timed region, time the same number of steps again over this many DIFFERENT "
                         "ragged batches (every clip of the workload cut by up to 4096 samples at either end, another cut per "
                         "batch) cycled through the plans in flight, so that no submit meets the clip lengths of the plan's "
                         "previous batch: what a window pipeline over real files sees (feature_extractor.py:228-235). "
                         "Reported as distinct_batches_frames_per_s; 0 = skip")
    ap.add_argument("--no-timing-events", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="in-flight sub-batches per GPU: the rank's clips are cut into this many runs, each with its "
                         "own context/stream/host thread (what batch_process does), so that one run's bandwidth-bound "
                         "kernels and host round trip overlap another's frame kernel")
    ap.add_argument("--inflight", type=int, default=2,
                    help="steps in flight on ONE stream (afx_extract_submit / afx_extract_collect, one plan per step in "
                         "flight, one host thread): every step is a whole-batch pass and the passes run back to back on "
                         "the device; the host's share of a step (wait, hand-out, next submit) falls under the next "
                         "step's kernels.  Needs --streams 1")
    ap.add_argument("--queue", choices=("shared", "own"), default="own",
                    help="with --inflight > 1: the steps in flight share one stream (back to back) or have one each "
                         "(the tail of one step's frame kernel under the start of the next)")
    args = ap.parse_args()
    if args.streams > 1:
        args.inflight = 1          # sub-batches on their own streams: one step at a time each

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    distributed = world > 1
    cfg = CONFIGS[args.config]
    SR, N_FFT, HOP, N_MFCC, N_MELS = cfg["sr"], cfg["n_fft"], cfg["hop"], cfg["n_mfcc"], cfg["n_mels"]

    import numpy as np
    from audio_feature_extraction_amd.synth import make_batch

    n_clips = args.clips
    workers = max(1, min(16, usable_cpus() // max(1, world)))
    samples, offsets, lengths = make_batch(n_clips, SR, SECONDS, first_index=rank * n_clips, workers=workers)

    # CPU baseline first (rank 0, N=1 only), before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and args.cpu_clips > 0:
        cpu = cpu_baseline(samples, offsets, lengths, cfg, args.cpu_clips, args.cpu_pool_seconds)

    import torch
    from audio_feature_extraction_amd import _native as N

    if not torch.cuda.is_available() or N.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # one rank per GPU; AFX_BENCH_BACKEND=gloo lets the N>1 path be rehearsed on a 1-GPU box
    backend = os.environ.get("AFX_BENCH_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)

    import threading
    params = lambda: N.make_params(SR, N_FFT, HOP, N_MFCC, N_MELS)   # noqa: E731
    S = max(1, min(args.streams, n_clips))
    cut = [n_clips * i // S for i in range(S + 1)]
    D = max(1, args.inflight)
    lanes = []
    for i in range(S if D == 1 else 0):
        lo, hi = cut[i], cut[i + 1]
        base = int(offsets[lo])
        end = int(offsets[hi - 1] + lengths[hi - 1])
        ctx = N.Context(device)
        plan = N.Plan(ctx, params())
        dbuf = N.DeviceBuffer(ctx, (end - base) * 4)
        dbuf.upload(samples[base:end])
        lanes.append({"ctx": ctx, "plan": plan, "dbuf": dbuf, "offsets": offsets[lo:hi] - base,
                      "lengths": lengths[lo:hi], "out": None})

    if D > 1:                # D plans on one context (one stream), all over the whole batch of one device buffer
        ctx = N.Context(device)
        dbuf = N.DeviceBuffer(ctx, samples.nbytes)
        dbuf.upload(samples)
        ctxs = [ctx]
        for i in range(D):
            c = ctx if args.queue == "shared" else N.Context(device)
            if c is not ctx:
                ctxs.append(c)
            lanes.append({"ctx": c, "plan": N.Plan(c, params()), "dbuf": dbuf, "offsets": offsets, "lengths": lengths,
                          "out": None, "busy": False})

    stagger = [0.0]          # seconds between the first submissions of consecutive lanes

    def lane_steps(lane, k, delay=0.0):
        if delay > 0.0:
            time.sleep(delay)
        for _ in range(k):
            lane["out"] = lane["plan"].extract_batch(lane["dbuf"], lane["offsets"], lane["lengths"], out=lane["out"])

    # One host thread per lane for the whole run (a fresh thread pays HIP's per-thread set-up again, several
    # milliseconds on its second call): commands go through a queue, completion through an event.
    import queue
    errs = []

    def lane_main(lane):
        while True:
            cmd = lane["q"].get()
            if cmd is None:
                return
            try:
                lane_steps(lane, cmd[0], cmd[1])
            except BaseException as e:      # a lane that dies must fail the run, not shorten it
                errs.append(e)
            lane["done"].set()

    if S > 1:
        for ln in lanes:
            ln["q"], ln["done"] = queue.Queue(), threading.Event()
            ln["thread"] = threading.Thread(target=lane_main, args=(ln,), daemon=True)
            ln["thread"].start()

    step_no = [0]

    def run_steps(k):
        if D > 1:
            for _ in range(k):
                ln = lanes[step_no[0] % D]
                step_no[0] += 1
                if ln["busy"]:
                    ln["out"] = ln["plan"].extract_collect()
                ln["plan"].extract_submit(ln["dbuf"], ln["offsets"], ln["lengths"], out=ln["out"])
                ln["busy"] = True
            for ln in lanes:
                if ln["busy"]:
                    ln["out"] = ln["plan"].extract_collect()
                    ln["busy"] = False
            return
        if S == 1:
            lane_steps(lanes[0], k)
            return
        # lanes that start together stay in lockstep (their frame kernels co-run and finish together); a start
        # offset of 1/S of a lane's step keeps one lane's frame kernel over the others' small kernels
        for i, ln in enumerate(lanes):
            ln["done"].clear()
            ln["q"].put((k, i * stagger[0]))
        for ln in lanes:
            ln["done"].wait()
        if errs:
            raise errs[0]

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(D)
    t_w = time.perf_counter()
    if D > 1:
        pass
    elif S > 1:
        lanes[0]["done"].clear(); lanes[0]["q"].put((3, 0.0)); lanes[0]["done"].wait()
    else:
        lane_steps(lanes[0], 3)
    stagger[0] = (time.perf_counter() - t_w) / 3 / S
    # a first round of untimed passes: the first few calls of a process pay one-off costs (workspace growth, pinned staging,
    # clock ramp) worth several steps, and the result arrays below are read from them
    run_steps(8)
    for ln in lanes:
        assert int((ln["out"]["status"] != 0).sum()) == 0, "synthetic clips must all succeed"
    frames_per_step = int(sum(int(ln["out"]["nframes"].sum()) for ln in (lanes[:1] if D > 1 else lanes)))

    # PCIe-inclusive rate (host float32 -> stats), reported beside the HBM-resident value
    h0 = time.perf_counter()
    hres = lanes[0]["plan"].extract_batch(samples[:int(offsets[cut[1] - 1] + lengths[cut[1] - 1])], offsets[:cut[1]], lengths[:cut[1]])
    host_dt = time.perf_counter() - h0
    host_frames = int(hres["nframes"].sum())

    # The frame kernel alone on the GPU: the WHOLE batch on one stream, a few launches outside the timed region
    # (a separate context; the same figure a single-stream rocprofv3 kernel trace of this command gives).
    exclusive = None
    if not args.no_timing_events and rank == 0:
        xc = N.Context(device)
        xp = N.Plan(xc, params())
        xb = N.DeviceBuffer(xc, samples.nbytes)
        xb.upload(samples)
        xo = None
        for _ in range(10):          # a fresh context: workspace growth, first touches and the clock ramp take several calls
            xo = xp.extract_batch(xb, offsets, lengths, out=xo)
        xp.set_timing(True)
        xp.timings(reset=True)
        for _ in range(20):
            xo = xp.extract_batch(xb, offsets, lengths, out=xo)
        torch.cuda.synchronize()
        xt = xp.timings()
        ms, cnt = xt["frames"]
        if cnt:
            xf = int(xo["nframes"].sum())
            ex_gbs = xf * 4.0 * HOP / (ms / cnt * 1e-3) / 1e9
            exclusive = {"avg_launch_ms": ms / cnt, "achieved": ex_gbs, "frac": ex_gbs / HBM_PEAK_GBS, "frames_per_launch": xf,
                         "launches": cnt, "kernels_ms_per_launch": {k: v[0] / max(v[1], 1) for k, v in xt.items()}}
        xb.free(); xp.close(); xc.close()
    # the untimed warm-up steps sit directly in front of the timed region: the two measurements above ran on other plans /
    # from host memory, and the lanes' first steps after them pay for it (their workspace is out of the TLB and the caches)
    # (with the frame-kernel events already switched on: their first records are not free either)
    if not args.no_timing_events:
        for ln in lanes:
            ln["plan"].set_timing(True, frames_only=True)      # one event pair per step: the kernel the roofline is about
    # at least 32 of them whatever --warmup says (about 20 ms): after the idle gaps of the set-up the clocks take about that
    # long to come back -- a 20-step region behind 8 warm-up steps ran 4-5 % below the same region behind 20 (same-box,
    # profiles/r03_ab_runs.txt); a job that streams batches is in the warmed state
    run_steps(max(args.warmup, 32))
    if not args.no_timing_events:
        for ln in lanes:
            ln["plan"].timings(reset=True)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kt, spans = None, None
    if not args.no_timing_events:
        kt, spans = {}, []
        for ln in lanes:
            spans.extend(map(tuple, ln["plan"].intervals("frames")))
            for k, v in ln["plan"].timings().items():
                a = kt.get(k, (0.0, 0))
                kt[k] = (a[0] + v[0], a[1] + v[1])

    # ---- the same steps over DISTINCT ragged batches (no plan ever meets the clip lengths of its previous batch)
    distinct = None
    if D > 1 and args.distinct >= 2:
        rng = np.random.default_rng(4242)          # the same cuts on every rank: equal frame counts
        variants = []
        for _ in range(args.distinct):
            a = rng.integers(0, 4096, n_clips).astype(np.int64)
            b = rng.integers(0, 4096, n_clips).astype(np.int64)
            variants.append((np.ascontiguousarray(offsets + a), np.ascontiguousarray(lengths - a - b)))
        for ln in lanes:
            ln["plan"].set_timing(False)
        vstep = [0]

        def run_distinct(k):
            frames = 0
            for _ in range(k):
                ln = lanes[vstep[0] % D]
                vo, vl = variants[vstep[0] % len(variants)]
                vstep[0] += 1
                if ln["busy"]:
                    frames += int(ln["plan"].extract_collect()["nframes"].sum())
                ln["plan"].extract_submit(ln["dbuf"], vo, vl, out=ln["out"])
                ln["busy"] = True
            for ln in lanes:
                if ln["busy"]:
                    frames += int(ln["plan"].extract_collect()["nframes"].sum())
                    ln["busy"] = False
            return frames

        run_distinct(2 * len(variants) * D)
        fence()
        t0 = time.perf_counter()
        dframes = run_distinct(args.steps)
        fence()
        d_el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([d_el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d_el = float(t.item())
        distinct = {"frames": dframes, "elapsed": d_el, "batches": len(variants)}

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = frames_per_step * world * args.steps
        value = total_frames / elapsed
        roof = None
        if kt is not None and kt["frames"][1] > 0:
            launches = kt["frames"][1]
            launches_per_step = launches / args.steps          # one frame-kernel launch per stream and step
            busy = union_ms(spans) if len(spans) == launches else kt["frames"][0]   # GPU time inside the kernel
            avg_ms = busy / launches
            bytes_per_launch = frames_per_step * 4.0 * HOP / launches_per_step
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            tflops = frames_per_step / launches_per_step * cfg["kflop"] * 1e3 / (avg_ms * 1e-3) / 1e12
            # HBM bytes from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE) cannot be collected inside
            # this process; the committed profile of this same command supplies them (a step's launches summed)
            traffic, tsrc = None, None
            try:
                import glob
                cand = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_cfg{args.config}_traffic.json")))
                if cand and n_clips == CLIPS_PER_GPU:
                    with open(cand[-1]) as fh:
                        traffic = json.load(fh)["frame_kernel_hbm_bytes_per_step"] / launches_per_step
                    tsrc = os.path.relpath(cand[-1], ROOT)
            except Exception:
                traffic = None
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "kernel": {(1024, 256): "k_frames3", (2048, 512): "k_frames3s", (512, 128): "k_frames3d"}.get(
                        (N_FFT, HOP), f"k_frames<{N_FFT}>") + " (speculative launch)",
                    "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step, "streams": S,
                    "kernel_ms_per_step": busy / args.steps,
                    "sum_of_launch_ms_per_step": kt["frames"][0] / args.steps,
                    "exclusive": exclusive,
                    "fp32": {"flops_per_frame": cfg["kflop"] * 1e3, "achieved_tflops": tflops, "peak": FP32_PEAK_TFLOPS,
                             "frac": tflops / FP32_PEAK_TFLOPS,
                             "note": "algorithmic FLOP per frame (SURVEY.md 8(d)) / the same launch time; vector peak, no MFMA on this path"},
                    "note": ("avg_launch_ms = union of the launch intervals of all %d streams (HIP events, common device clock) "
                             "/ launches: GPU time inside the kernel, <= ms_per_step; sum_of_launch_ms counts overlapped time "
                             "once per stream; 'exclusive' = the whole batch on one stream" % S),
                    "inflight": D,
                    "kernels_ms_per_step": {k: v[0] / args.steps for k, v in kt.items() if v[1] > 0}}
        line = {
            "metric": f"audio frames/sec (sr={SR}, n_fft={N_FFT}, hop={HOP}, n_mfcc={N_MFCC})",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n_clips}x10s clips @{SR} Hz per GPU, frame_length={N_FFT}, hop_length={HOP}, "
                                   f"n_mfcc={N_MFCC}, n_mels={N_MELS}, hamming, pre-emphasis 0.97 + trim 30 dB + RMS "
                                   f"(BASELINE configs[{cfg['baseline_index'] if world == 1 or args.config != 2 else 3}])",
                       "clips_per_gpu": n_clips, "frames_per_gpu_step": frames_per_step,
                       "parallelism": (f"file-shard x{world}, no collective; {S} in-flight sub-batches per GPU" if D == 1 else
                                       f"file-shard x{world}, no collective; {D} whole-batch steps in flight on {'one stream' if args.queue == 'shared' else 'their own streams'} (submit / collect)"),
                       "input": "HBM-resident float32"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "host_to_result_frames_per_s": host_frames / host_dt,
        }
        if distinct is not None:
            dv = distinct["frames"] * world / distinct["elapsed"]
            line["distinct_batches_frames_per_s"] = dv
            line["distinct_batches"] = {
                "value": dv, "unit": "frames/s", "batches": distinct["batches"], "steps": args.steps,
                "ms_per_step": distinct["elapsed"] / args.steps * 1e3, "ratio_to_value": dv / value,
                "note": "the timed region repeated over different ragged batches of the same workload (each clip cut by up "
                        "to 4096 samples at either end, another cut per batch), cycled through the plans in flight: every "
                        "submit uploads new clip records and the device rebuilds its block list (no per-batch cache hit)"}
        print(json.dumps(line), flush=True)

    if S > 1:
        for ln in lanes:
            ln["q"].put(None)
        for ln in lanes:
            ln["thread"].join()
    for ln in lanes:
        ln["plan"].close()
    if D > 1:
        lanes[0]["dbuf"].free()
        for c in ctxs[1:] + ctxs[:1]:
            c.close()
    else:
        for ln in lanes:
            ln["dbuf"].free()
            ln["ctx"].close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
