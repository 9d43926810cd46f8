#!/bin/bash
# per-frame-pair counters of the speculative k_frames3 launch: tools/pmc_pair.sh <tag> [lib]
TAG=${1:-q}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$2" ] && export AFX_LIB=$ROOT/$2
OUT=$ROOT/gpurun_out/pmcp_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-clips 0 --streams 1"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1 || echo "pmc $name failed"
done
python3 - $OUT <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],'*','**','*counter_collection.csv'),recursive=True):
    for row in csv.DictReader(open(f)):
        n=row['Kernel_Name']
        if 'k_frames3' in n and 'true>' in n.split('(')[0]: acc[row['Counter_Name']].append(float(row['Counter_Value']))
print(' '.join(f"{c[3:] if c.startswith('SQ_') else c}={sum(acc[c])/len(acc[c])/431000:.1f}" for c in sorted(acc)), '(per pair)')
PY
