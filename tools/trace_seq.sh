#!/bin/bash
# per-dispatch durations in launch order (GPU box): tools/trace_seq.sh LEG [ENV=VAL]
export TMPDIR=/tmp; REPO=$(pwd); OUT=$REPO/gpurun_out/trace_seq; rm -rf $OUT; mkdir -p $OUT
[ -n "$2" ] && export "$2"
export JRC_BENCH_EXTRA_ONLY=$1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/bench_extra.py > $OUT/log.txt 2>&1 )
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])) for r in rows]
# print the last 24 dispatches with gaps
for i in range(max(1, len(seq) - 24), len(seq)):
    print("%-42s %8.1f us   gap before %6.1f us" % (seq[i][0], seq[i][1], (seq[i][2] - seq[i - 1][2]) / 1e3 - seq[i - 1][1]))
P
