#!/bin/bash
# per-phase cycle stamps of the frame kernel for each AFX_DEBUG_SKIP mask given
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  echo "skip=$m"
  AFX_DEBUG_SKIP=$m AFX_DEBUG_STAMPS=1 python bench.py --steps 1 --warmup 1 --cpu-clips 0 --streams 1 2>&1 | grep "avg over waves" | tail -1
done
