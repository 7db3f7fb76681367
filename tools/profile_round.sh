#!/bin/bash
# Round profiling pass (run on the GPU box via gpurun): stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of the
# default bench command for configs B and D; summaries land in gpurun_out/prof_<tag>/ as CSV.
# usage: tools/profile_round.sh TAG
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for CFG in B D; do
  python3 bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats_$CFG -o s -- python3 $REPO/bench.py --config $CFG --steps 20 --no-cpu-baseline --no-secondary --no-check --windows 1 > $OUT/stats_$CFG.log 2>&1 )
  ( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$CFG -o s -- python3 $REPO/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-check --windows 1 > $OUT/fetch_$CFG.log 2>&1 )
  ( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$CFG -o s -- python3 $REPO/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-check --windows 1 > $OUT/write_$CFG.log 2>&1 )
  S=$(find $OUT/stats_$CFG -name '*.db' | head -1); Fd=$(find $OUT/fetch_$CFG -name '*.db' | head -1); W=$(find $OUT/write_$CFG -name '*.db' | head -1)
  python3 profiles/summarize_rocpd.py $OUT/summary_$CFG.csv "$TAG config $CFG: python3 bench.py --config $CFG (default batch)" $S $Fd $W > /dev/null
  if [ $CFG = B ]; then SPEC_B="B:512:$Fd:$W"; else SPEC_D="D:256:$Fd:$W"; fi
done
python3 tools/make_pmc_traffic.py $OUT/pmc_traffic.json $TAG $SPEC_B $SPEC_D > /dev/null
find $OUT -name '*.db' -delete
# the bench line again, now with the traffic of THIS tree beside it
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
python3 bench.py --config B > $OUT/bench_B_with_traffic.json 2>> $OUT/bench_B.err
tail -n +1 $OUT/summary_B.csv $OUT/summary_D.csv
cat $OUT/bench_B.json $OUT/bench_D.json
