#!/bin/bash
# A/B two builds of libafx.so on one box, every kernel: tools/ab_step.sh <libA> <libB> [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    AFX_LIB=$PWD/$L python bench.py --steps 20 --warmup 3 --cpu-clips 0 --streams 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$L', 'step_ms=%.4f'%d['ms_per_step'], ' '.join('%s=%.4f'%(a,b) for a,b in k.items()))"
  done
done
