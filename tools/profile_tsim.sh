#!/bin/bash
# rocprofv3 kernel stats of the device target simulator (run on the GPU box).  usage: tools/profile_tsim.sh TAG [probe args]
TAG=${1:-r01}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_tsim_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $REPO/tools/tsim_probe.py --no-oracle "$@" > $OUT/probe.log 2>&1 )
S=$(find $OUT/stats -name '*.db' | head -1)
python3 profiles/summarize_rocpd.py $OUT/summary.csv "$TAG target simulator: tools/tsim_probe.py --no-oracle $*" $S > /dev/null
find $OUT -name '*.db' -delete
cat $OUT/probe.log | grep config; cat $OUT/summary.csv
