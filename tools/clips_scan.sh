#!/bin/bash
# frame-kernel time vs batch size (fixed cost / tail of the persistent kernel): tools/clips_scan.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for C in 250 500 1000 2000 4000; do
  python bench.py --clips $C --steps 10 --warmup 3 --cpu-clips 0 --streams 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('clips=$C', ' '.join('%s=%.4f'%(a,b) for a,b in k.items()), 'step=%.4f'%d['ms_per_step'])"
done
