#!/bin/bash
# whole-step throughput vs steps in flight on one stream, with and without per-kernel timing events: tools/inflight_scan.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for D in 1 2 3; do for T in "" "--no-timing-events"; do
  python bench.py --steps ${STEPS:-60} --warmup 5 --cpu-clips 0 --streams 1 --inflight $D $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inflight $D $T', 'value=%.4e step_ms=%.4f'%(d['value'], d['ms_per_step']))"
done; done
