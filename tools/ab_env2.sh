#!/bin/bash
# Same-box A/B of environment settings on the DEFAULT bench (two whole-batch steps in flight): tools/ab_env2.sh <rounds> <config> "VAR=a VAR2=b" "VAR=c" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$1; C=$2; shift 2
for i in $(seq $R); do
  for E in "$@"; do
    env $E python bench.py --config $C --steps 60 --warmup 10 --cpu-clips 0 --distinct 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', 'value=%.4g step_ms=%.4f frames_ms=%.4f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
