#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for T in 1 0 1 0; do
  if [ $T = 1 ]; then export AFX_NO_DCT16L=1; else unset AFX_NO_DCT16L; fi
  python bench.py --config 3 --steps 30 --warmup 5 --cpu-clips 0 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 3 no_dct16l $T', 'value=%.4e step_ms=%.4f'%(d['value'], d['ms_per_step']), d['roofline']['exclusive']['kernels_ms_per_launch'])"
done
