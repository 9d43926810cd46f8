#!/bin/bash
# A/B two builds of libafx.so on one box: tools/ab.sh <libA> <libB> [rounds]   (frame-kernel ms per launch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    AFX_LIB=$PWD/$L python bench.py --steps 20 --warmup 3 --cpu-clips 0 --streams 1 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'frames_ms=%.4f step_ms=%.4f'%(d['roofline']['kernels_ms_per_step']['frames'], d['ms_per_step']))"
  done
done
