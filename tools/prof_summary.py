#!/usr/bin/env python3
"""Condenses rocprofv3 csv output (kernel stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i == 0 or "afx" in row[0]:
                print(",".join(c[:60] for c in row))
print("== PMC (per-dispatch average over afx kernels) ==")
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "afx" not in k:
                continue
            short = k.split("afx")[1][:24] if "afx" in k else k[:24]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:28s} avg {sum(v)/len(v):16.1f}  n={len(v)}")
