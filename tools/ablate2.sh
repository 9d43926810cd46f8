#!/bin/bash
# frame-kernel ms per launch (single stream) for each AFX_DEBUG_SKIP value given (decimal: 1 loads, 2 FFT, 4 mel, 8 stores, 16/32 stagger)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  AFX_DEBUG_SKIP=$m python bench.py --steps 10 --warmup 2 --cpu-clips 0 --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('skip=$m', 'frames_ms=%.3f' % d['roofline']['avg_launch_ms'])"
done
