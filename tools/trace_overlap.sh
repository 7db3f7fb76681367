#!/bin/bash
# start / end of every dispatch of one pipelined detect-only step relative to the step's first dispatch (GPU box): does A1 of slice i+1 run beside the
# detect kernel of slice i?   usage: tools/trace_overlap.sh SLICES   -> gpurun_out/trace_overlap_SLICES.txt
export TMPDIR=/tmp; REPO=$(pwd); OUT=$REPO/gpurun_out/trace_overlap; rm -rf $OUT; mkdir -p $OUT
export JRC_DETECT_SLICES=${1:-4} JRC_BENCH_EXTRA_ONLY=detectB
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/bench_extra.py > $OUT/log.txt 2>&1 )
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 - "$F" "$JRC_DETECT_SLICES" <<'P' | tee $REPO/gpurun_out/trace_overlap_$JRC_DETECT_SLICES.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
# the sliced steps are the ones where radar_chanest_x2 launches come n to a step with detect launches in between: take the last complete step that
# ran WITHOUT per-kernel events (the timed loop before set_timing): find runs of dispatches whose A1 launches cover 512/n frames each
names = [r["Kernel_Name"] for r in rows]
a1 = [i for i, k in enumerate(names) if "radar_chanest" in k]
# walk back from the middle of the trace to a dispatch pattern with overlap potential: print 3 consecutive steps' worth from 60 % into the trace
i0 = a1[int(len(a1) * 0.45)]
t0 = int(rows[i0]["Start_Timestamp"])
print("# JRC_DETECT_SLICES=%d, config B, 512 frames per step; times in us relative to the first dispatch shown; queue = stream" % n)
print("# %-44s %10s %10s %8s  %s" % ("kernel", "start", "end", "dur", "queue"))
prev_end = {}
for r in rows[i0:i0 + 8 * n + 8]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("  %-44s %10.1f %10.1f %8.1f  %s" % (r["Kernel_Name"].split("(")[0][:44], s, e, e - s, r.get("Queue_Id", "?")))
P
