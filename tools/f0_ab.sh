#!/bin/bash
# Same-box A/B of the pYIN stage between builds of libafx.so: tools/f0_ab.sh <rounds> <n_clips> lib1 lib2 ...  (paths relative
# to the repo); tools/f0_time.py per library, interleaved over the rounds.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$1; N=$2; shift 2
for i in $(seq $R); do
  for L in "$@"; do
    echo -n "$L: "; AFX_LIB=$PWD/$L python tools/f0_time.py $N
  done
done
