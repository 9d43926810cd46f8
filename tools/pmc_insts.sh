#!/bin/bash
# instruction-mix counters of the frame kernel: tools/pmc_insts.sh <tag> [AFX_DEBUG_SKIP]
TAG=${1:-q}; export AFX_DEBUG_SKIP=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmci_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-clips 0 --streams 1 > $OUT/a.log 2>&1 || echo "pmc failed"
python3 - $OUT <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],'*','**','*counter_collection.csv'),recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_frames' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
print(' '.join(f"{c}={sum(acc[c])/len(acc[c])/53875/4:.0f}" for c in sorted(acc)), '(per wave per 16-frame block)')
PY
