#!/bin/bash
# Round evidence for the secondary kernels (run on the GPU box via gpurun): for each leg of tools/bench_extra.py a
# `rocprofv3 --kernel-trace --stats` pass plus separate PMC passes (FETCH_SIZE, WRITE_SIZE, two sets of SQ counters); the program
# comes directly after `--`.  usage: tools/profile_secondary.sh TAG "leg[:ENV=VAL] ..."   -> gpurun_out/prof_TAG/<name>_rocprof_summary.csv
TAG=${1:-r03}
# default: one summary per secondary key of bench.py's line that has kernels of its own
LEGS=${2:-"demodB detectB detectB_noise powerB flowgraphB equalizer precoder rdD"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for SPEC in $LEGS; do
  NAME=${SPEC%%:*}; EXTRA=""; [ "$SPEC" != "$NAME" ] && EXTRA=${SPEC#*:}
  LEG=${NAME%%_unpruned}
  LEG=${LEG%%_sigfull}
  # a leg's ENV=VAL holds for that leg only: whatever the previous leg exported is unset again before the next one starts
  [ -n "$PREV_VAR" ] && unset "$PREV_VAR"
  PREV_VAR=""
  if [ -n "$EXTRA" ]; then export "$EXTRA"; PREV_VAR=${EXTRA%%=*}; fi
  export JRC_BENCH_EXTRA_ONLY=$LEG
  python3 tools/bench_extra.py > $OUT/$NAME.json 2> $OUT/$NAME.err
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/${NAME}_stats -o s -- python3 $REPO/tools/bench_extra.py > $OUT/${NAME}_stats.log 2>&1 )
  DBS=""
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    ( cd /tmp && rocprofv3 --kernel-trace --pmc $SET -d $OUT/${NAME}_p$i -o s -- python3 $REPO/tools/bench_extra.py > $OUT/${NAME}_p$i.log 2>&1 )
    DBS="$DBS $(find $OUT/${NAME}_p$i -name '*.db' | head -1)"
  done
  python3 profiles/summarize_rocpd.py $OUT/${NAME}_rocprof_summary.csv "$TAG $NAME: JRC_BENCH_EXTRA_ONLY=$LEG $EXTRA python3 tools/bench_extra.py -> $(cat $OUT/$NAME.json | cut -c1-600)" $(find $OUT/${NAME}_stats -name '*.db' | head -1) $DBS > /dev/null
  find $OUT -name '*.db' -delete
  rm -rf $OUT/${NAME}_stats $OUT/${NAME}_p[0-9]
  head -12 $OUT/${NAME}_rocprof_summary.csv | cut -c1-220
done
