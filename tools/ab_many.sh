#!/bin/bash
# Same-box A/B of several builds of libafx.so: tools/ab_many.sh <rounds> <config> lib1 lib2 ...   (paths relative to the repo)
# prints the frame kernel's ms per launch and the whole step, one call at a time (--inflight 1), interleaved over the rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$1; C=$2; shift 2
for i in $(seq $R); do
  for L in "$@"; do
    AFX_LIB=$PWD/$L python bench.py --config $C --steps 30 --warmup 5 --cpu-clips 0 --streams 1 --inflight 1 --distinct 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'frames_ms=%.4f step_ms=%.4f'%(d['roofline']['kernels_ms_per_step']['frames'], d['ms_per_step']))"
  done
done
