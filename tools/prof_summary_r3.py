#!/usr/bin/env python3
"""Condenses tools/profile_r3.sh output: per config the kernel stats (single stream), the PMC averages per kernel and a
traffic summary (HBM bytes per step from FETCH_SIZE x2 + WRITE_SIZE over all kernels against the algorithmic bytes).
Writes profiles-ready files next to the raw output: cfgC_kernel_stats.csv, cfgC_pmc.txt, cfgC_traffic.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
ALGO = {2: (862000, 256), 3: (1251000, 128), 5: (862000, 512)}       # frames per step, hop


def kernel_stats(d):
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        with open(f) as fh:
            for i, row in enumerate(csv.reader(fh)):
                if i == 0 or "afx" in row[0]:
                    rows.append([c[:90] for c in row])
    return rows


def short(k):
    k = k.split("afx::", 1)[1] if "afx::" in k else k
    return k.split("(")[0][:48]


for cfg in (2, 3, 5):
    tr = os.path.join(root, f"cfg{cfg}_trace")
    if not os.path.isdir(tr):
        continue
    print(f"== config {cfg}: kernel stats, single stream (rocprofv3 --kernel-trace --stats; bench.py --config {cfg} --steps 20 --warmup 3 --cpu-clips 0 --streams 1 --inflight 1 --distinct 0) ==")
    rows = kernel_stats(tr)
    with open(os.path.join(root, f"cfg{cfg}_kernel_stats.csv"), "w") as fh:
        for r in rows:
            fh.write(",".join(r) + "\n")
            print(",".join(r))
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, f"cfg{cfg}_pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "afx" in k:
                    acc[short(k)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = [f"== config {cfg}: PMC, per-dispatch average (separate rocprofv3 --pmc passes; bench.py --config {cfg} --steps 5 --warmup 2 --cpu-clips 0 --streams 1 --inflight 1 --distinct 0) =="]
    for k in sorted(acc):
        lines.append(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            lines.append(f"   {c:28s} avg {sum(v)/len(v):16.1f}  n={len(v)}")
    with open(os.path.join(root, f"cfg{cfg}_pmc.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # traffic: FETCH_SIZE / WRITE_SIZE are in KB per dispatch; gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads
    frames, hop = ALGO[cfg]
    per_kernel = {}
    for k in acc:
        if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
            fe = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"]) * 1024 * 2
            wr = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"]) * 1024
            per_kernel[k] = {"fetch_bytes_x2": fe, "write_bytes": wr}
    algo = frames * 4.0 * hop
    tj = {"round": 3, "config": cfg,
          "source": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/profile_r3.sh), one step in flight: one launch of every kernel per step",
          "correction": "gfx950: FETCH_SIZE counts half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact; KB -> bytes",
          "per_kernel_bytes_per_launch": per_kernel,
          "algorithmic_bytes_per_step": algo}
    pipe = sum(v["fetch_bytes_x2"] + v["write_bytes"] for v in per_kernel.values())
    tj["pipeline_hbm_bytes_per_step"] = pipe
    tj["pipeline_over_algorithmic"] = pipe / algo
    frame_k = sorted((k for k in per_kernel if k.startswith("k_frames")), key=lambda k: -per_kernel[k]["fetch_bytes_x2"])   # the speculative launch reads the batch
    if frame_k:
        v = per_kernel[frame_k[0]]
        tj["frame_kernel"] = frame_k[0]
        tj["frame_kernel_hbm_bytes_per_step"] = v["fetch_bytes_x2"] + v["write_bytes"]
        tj["frame_kernel_over_algorithmic"] = tj["frame_kernel_hbm_bytes_per_step"] / algo
    with open(os.path.join(root, f"cfg{cfg}_traffic.json"), "w") as fh:
        json.dump(tj, fh, indent=1)
    print(json.dumps({k: tj[k] for k in ("pipeline_hbm_bytes_per_step", "algorithmic_bytes_per_step", "pipeline_over_algorithmic")}))
s3 = os.path.join(root, "cfg2_default_trace")
if os.path.isdir(s3):
    print("== config 2: kernel stats, the default run: two steps in flight on their own streams (bench.py --steps 20 --warmup 3 --cpu-clips 0) ==")
    rows = kernel_stats(s3)
    with open(os.path.join(root, "cfg2_default_kernel_stats.csv"), "w") as fh:
        for r in rows:
            fh.write(",".join(r) + "\n")
            print(",".join(r))

f0 = os.path.join(root, "f0_trace")
if os.path.isdir(f0):
    print("== extract_f0 (pYIN), 1000 x 10 s clips @22050 Hz 1024/256, four calls of afx_f0_batch (tools/f0_time.py 1000; the first touches the workspace) ==")
    rows = []
    for f in sorted(glob.glob(os.path.join(f0, "**", "*kernel_stats.csv"), recursive=True)):
        with open(f) as fh:
            for i, row in enumerate(csv.reader(fh)):
                if i == 0 or "afx" in row[0]:
                    rows.append([c[:90] for c in row])
    with open(os.path.join(root, "f0_kernel_stats.csv"), "w") as fh:
        for r in rows:
            fh.write(",".join(r) + "\n")
            print(",".join(r))
