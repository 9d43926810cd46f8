#!/bin/bash
# PMC passes over the pYIN stage (tools/f0_time.py 1000), each counter group in its own run, then a per-kernel summary.
# usage (on the GPU box): tools/profile_f0_pmc.sh <tag>   -> gpurun_out/prof_<tag>/f0_pmc.txt
set -u
TAG=${1:-r03f0}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/f0_pmc_$name -- python3 $ROOT/tools/f0_time.py 1000 > $OUT/f0_pmc_$name.log 2>&1
  echo "f0 pmc [$grp] rc=$?"
done
python3 - "$OUT" > $OUT/f0_pmc.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "f0_pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "afx::k_f0" in k:
            k = k.split("afx::", 1)[1].split("(")[0][:40]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== pYIN stage: PMC, per-dispatch average (separate rocprofv3 --pmc passes over tools/f0_time.py 1000: 1000 x 10 s clips, 862 000 frames; four calls per run) ==")
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:28s} avg {sum(v)/len(v):18.1f}  n={len(v)}")
PY
tail -30 $OUT/f0_pmc.txt
