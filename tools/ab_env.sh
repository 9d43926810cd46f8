#!/bin/bash
# frame-kernel ms per launch under different environment settings, same box: tools/ab_env.sh "VAR=1" "VAR=2" ...
# (STREAMS=n in the caller's environment: in-flight sub-batches, default 1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for E in "$@"; do
    env $E python bench.py --steps ${STEPS:-20} --warmup 3 --cpu-clips 0 --streams ${STREAMS:-1} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$E', 'frames_ms=%.4f step_ms=%.4f value=%.3e'%(d['roofline']['kernels_ms_per_step']['frames'], d['ms_per_step'], d['value']))"
  done
done
