#!/bin/bash
# whole-step throughput over (waves per CU, streams, ticketed runs) on one box: tools/combo_scan.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for W in 16 12; do for S in 1 2 3; do for T in 0 1; do
  if [ $T = 1 ]; then export AFX_NO_TICKETS=1; else unset AFX_NO_TICKETS; fi
  AFX_F3_WAVES=$W python bench.py --steps ${STEPS:-60} --warmup 5 --cpu-clips 0 --streams $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves $W streams $S no_tickets $T', 'value=%.4e step_ms=%.4f frames_ms=%.4f'%(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step']['frames']))"
done; done; done
