#!/bin/bash
# quick PMC of k_frames only: tools/pmc_quick.sh <tag> [AFX_DEBUG_SKIP]
TAG=${1:-q}; export AFX_DEBUG_SKIP=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-clips 0 --streams 1"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1 || echo "pmc $name failed"
done
python3 - $OUT <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],'*','**','*counter_collection.csv'),recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_frames' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
for c in sorted(acc): print(f"{c:28s} {sum(acc[c])/len(acc[c]):16.0f}")
PY
