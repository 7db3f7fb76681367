#!/bin/bash
# usage (on the GPU box): tools/profile_cmd.sh TAG script.py args...   -> gpurun_out/prof_TAG/summary.csv (stats + FETCH_SIZE + WRITE_SIZE passes)
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SCRIPT=$REPO/$1; shift
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $SCRIPT "$@" > $OUT/stats.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o s -- python3 $SCRIPT "$@" > $OUT/fetch.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o s -- python3 $SCRIPT "$@" > $OUT/write.log 2>&1 )
S=$(find $OUT/stats -name '*.db' | head -1); Fd=$(find $OUT/fetch -name '*.db' | head -1); W=$(find $OUT/write -name '*.db' | head -1)
python3 profiles/summarize_rocpd.py $OUT/summary.csv "$TAG: $SCRIPT $*" $S $Fd $W > /dev/null
find $OUT -name '*.db' -delete
cat $OUT/summary.csv; tail -2 $OUT/stats.log
