#!/bin/bash
# rocprofv3 evidence for the A6+A7+A1 kernel (demod_chanest16_kernel) on the GPU box: stats + FETCH / WRITE + instruction and cycle counters
# usage: tools/demod_profile.sh TAG CONFIG FRAMES   -> gpurun_out/prof_TAG/summary.csv
TAG=$1; CFG=${2:-D}; FR=${3:-256}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
DBS=""
i=0
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $REPO/tools/td_kernel_probe.py $CFG $FR > $OUT/stats.log 2>&1 )
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o s -- python3 $REPO/tools/td_kernel_probe.py $CFG $FR > $OUT/p$i.log 2>&1 )
  DBS="$DBS $(find $OUT/p$i -name '*.db' | head -1)"
done
python3 profiles/summarize_rocpd.py $OUT/summary.csv "$TAG: tools/td_kernel_probe.py $CFG $FR" $(find $OUT/stats -name '*.db' | head -1) $DBS > /dev/null
find $OUT -name '*.db' -delete
grep "demod_chanest" $OUT/summary.csv | cut -c1-40,170-
