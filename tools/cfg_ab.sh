#!/bin/bash
# configs 3 and 5: ticketed vs equal shares, same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for C in 3 5; do for T in 1 0 1 0; do
  if [ $T = 1 ]; then export AFX_NO_TICKETS=1; else unset AFX_NO_TICKETS; fi
  python bench.py --config $C --steps 30 --warmup 5 --cpu-clips 0 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $C no_tickets $T', 'value=%.4e step_ms=%.4f frames_ms=%.4f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
