#!/bin/bash
# rocprofv3 recipe of round 3 (run on the GPU box): for BASELINE configs 2 / 3 / 5
#   * kernel-trace stats of a run with ONE step in flight (one launch of every kernel per step, nothing beside it: what
#     bench.py's `exclusive` times), and, for config 2, of the default run (two steps in flight on their own streams);
#   * PMC passes, each in its own run (SQ counters; FETCH_SIZE; WRITE_SIZE).
# usage: tools/profile_r3.sh <tag> [configs...]   -> gpurun_out/prof_<tag>/..., summaries copied to profiles/ by hand
set -u
TAG=${1:-r03}; shift
CFGS=${@:-2 3 5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in $CFGS; do
  BENCH="python3 $ROOT/bench.py --config $C --steps 20 --warmup 3 --cpu-clips 0 --streams 1 --inflight 1 --distinct 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg${C}_trace -- $BENCH > $OUT/cfg${C}_trace.log 2>&1
  echo "cfg$C trace rc=$?"
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --output-format csv -d $OUT/cfg${C}_pmc_$name -- python3 $ROOT/bench.py --config $C --steps 5 --warmup 2 --cpu-clips 0 --streams 1 --inflight 1 --distinct 0 > $OUT/cfg${C}_pmc_$name.log 2>&1
    echo "cfg$C pmc [$grp] rc=$?"
  done
done
if echo " $CFGS " | grep -q " 2 "; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg2_default_trace -- python3 $ROOT/bench.py --steps 20 --warmup 3 --cpu-clips 0 --distinct 0 > $OUT/cfg2_default_trace.log 2>&1
  echo "cfg2 default (two steps in flight) trace rc=$?"
fi
# extract_f0 (pYIN) of the same 1000 clips: the GPU time batch_process actually spends
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f0_trace -- python3 $ROOT/tools/f0_time.py 1000 > $OUT/f0_trace.log 2>&1
echo "f0 trace rc=$?"
python3 $ROOT/tools/prof_summary_r3.py $OUT > $OUT/summary.txt 2>&1
tail -5 $OUT/summary.txt
