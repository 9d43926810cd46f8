#!/bin/bash
# rocprofv3 recipe for the GPU box: kernel-trace stats, then PMC passes (each in its own run).
# usage: tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-clips 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_VALU"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1
  echo "pmc [$grp] rc=$?"
done
# summarise
python3 $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
