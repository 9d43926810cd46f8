#!/bin/bash
# k_frames3 ablations (timing only): time + LDS counters per AFX_DEBUG_SKIP value.  Needs `make dbg`.
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$PWD
export AFX_LIB=$ROOT/audio_feature_extraction_amd/libafx_dbg.so AFX_F3_DEBUG=1 AFX_F3_WAVES=${AFX_F3_WAVES:-16}
for SK in 0 4 8 12 2 32 34 46; do
  export AFX_DEBUG_SKIP=$SK
  T=$(python bench.py --steps 10 --warmup 3 --cpu-clips 0 --streams 1 --inflight 1 --distinct 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f'%d['roofline']['kernels_ms_per_step']['frames'])")
  OUT=$ROOT/gpurun_out/f3abl_$SK; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-clips 0 --streams 1 --inflight 1 --distinct 0 > $OUT.log 2>&1)
  python3 - $OUT $SK $T <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1],'**','*counter_collection.csv'),recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_frames' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
print('skip=%s frames_ms=%s '%(sys.argv[2],sys.argv[3]) + ' '.join(f"{c[3:]}={sum(acc[c])/len(acc[c])/431000:.0f}" for c in sorted(acc)), '(per pair)')
PY
done
