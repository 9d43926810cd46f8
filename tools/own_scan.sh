#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for W in 16 12; do for Q in shared own; do for D in 2 3; do
  AFX_F3_WAVES=$W python bench.py --steps ${STEPS:-80} --warmup 5 --cpu-clips 0 --inflight $D --queue $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves $W queue $Q inflight $D', 'value=%.4e step_ms=%.4f frames_ms=%.4f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done; done
