#!/bin/bash
# kernel timeline of a multi-stream run (rocprofv3 --kernel-trace): tools/timeline.sh <streams> <waves> <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-2}; W=${2:-16}; TAG=${3:-tl}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AFX_F3_WAVES=$W rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 12 --warmup 3 --cpu-clips 0 --streams $S --no-timing-events ${EXTRA:-} > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], 'trace', '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        n = n[n.find('afx::') + 5:] if 'afx::' in n else n
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), n.split('(')[0][:28]))
rows.sort()
t0 = rows[-90][0] if len(rows) > 90 else rows[0][0]
with open(os.path.join(sys.argv[1], 'timeline.txt'), 'w') as fh:
    for s, e, q, n in rows[-90:]:
        fh.write(f"{(s - t0) / 1000:9.1f} {(e - t0) / 1000:9.1f} {(e - s) / 1000:8.1f} us  q{q:>3s}  {n}\n")
PY
tail -60 $OUT/timeline.txt
