#!/bin/bash
# A/B two builds on another BASELINE config: tools/ab_cfg.sh <config> <libA> <libB> [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do
  for L in $A $B; do
    AFX_LIB=$PWD/$L python bench.py --config $C --steps 20 --warmup 3 --cpu-clips 0 --streams 1 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'frames_ms=%.4f step_ms=%.4f'%(d['roofline']['kernels_ms_per_step']['frames'], d['ms_per_step']))"
  done
done
