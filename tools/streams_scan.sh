#!/bin/bash
# whole-step throughput vs in-flight sub-batches: tools/streams_scan.sh [env assignments...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for E in "$@"; do
for S in 1 2 3 4; do
  env $E python bench.py --steps 50 --warmup 5 --cpu-clips 0 --streams $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$E streams', d['roofline']['streams'], 'value %.3e ms_per_step %.4f'%(d['value'], d['ms_per_step']), ' '.join('%s=%.3f'%(a,b) for a,b in k.items()))"
done; done
