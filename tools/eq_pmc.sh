#!/bin/bash
# PMC passes over the equalizer at config C (tools/bench_extra.py, equalizer leg only): instruction fetch, wait and issue counters.
# usage (on the GPU box): tools/eq_pmc.sh TAG     -> gpurun_out/eqpmc_TAG/summary.csv
TAG=${1:-x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/eqpmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp JRC_BENCH_EXTRA_ONLY=${EQ_PMC_LEG:-equalizer}
DBS=""
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQC_ICACHE_HITS SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_INSTS_BRANCH"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o s -- python3 $REPO/tools/bench_extra.py > $OUT/p$i.log 2>&1 )
  DBS="$DBS $(find $OUT/p$i -name '*.db' | head -1)"
done
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $REPO/tools/bench_extra.py > $OUT/stats.log 2>&1 )
python3 profiles/summarize_rocpd.py $OUT/summary.csv "$TAG equalizer config C" $(find $OUT/stats -name '*.db' | head -1) $DBS > /dev/null
find $OUT -name '*.db' -delete
grep -i "equalizer_kernel" $OUT/summary.csv | cut -c1-60,200-
