#!/bin/bash
# Same-box A/B of several builds of libafx.so, per-kernel: tools/ab_kernels.sh <rounds> <config> lib1 lib2 ...  (paths relative
# to the repo); one step at a time (--inflight 1), prints every kernel slot's ms per step and the whole step
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$1; C=$2; shift 2
for i in $(seq $R); do
  for L in "$@"; do
    AFX_LIB=$PWD/$L python bench.py --config $C --steps 30 --warmup 5 --cpu-clips 0 --streams 1 --inflight 1 --distinct 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', ' '.join('%s=%.4f'%(k,v) for k,v in d['roofline']['kernels_ms_per_step'].items()), 'step_ms=%.4f'%d['ms_per_step'])"
  done
done
