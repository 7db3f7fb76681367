#!/bin/bash
# usage (GPU box): tools/stats_cmd.sh script.py args...  -> per-kernel calls / average us of one `rocprofv3 --kernel-trace --stats` pass
REPO=$(pwd); OUT=$REPO/gpurun_out/stats_tmp; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
SCRIPT=$REPO/$1; shift
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o s -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1 )
tail -1 $OUT/run.log | cut -c1-200
S=$(find $OUT -name '*.db' | head -1)
python3 profiles/summarize_rocpd.py $OUT/summary.csv "$SCRIPT $*" $S > /dev/null
python3 - <<PY
import csv
rows = [r for r in csv.reader(l for l in open("$OUT/summary.csv") if not l.startswith("#")) if len(r) >= 5 and r[1].isdigit()]
for r in rows[:14]:
    if "at::native" in r[0] or "rocclr" in r[0]: continue
    print("%-60s calls %5s  avg %10.3f us  %5.1f %%" % (r[0][:60], r[1], float(r[3]), float(r[4])))
PY
find $OUT -name '*.db' -delete
