#!/bin/bash
# usage (GPU box): tools/pmc_cmd.sh "CTR1 CTR2 ..." script.py args...  -> per-kernel averages of the counters
CTRS=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_tmp; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
SCRIPT=$REPO/$1; shift
( cd /tmp && rocprofv3 --kernel-trace --pmc $CTRS -d $OUT -o s -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1 )
tail -2 $OUT/run.log | cut -c1-200
python3 - <<PY
import sqlite3, glob
dbs = glob.glob("$OUT/**/*.db", recursive=True)
if not dbs: raise SystemExit("no db")
c = sqlite3.connect(dbs[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
q = """select k.kernel_name, p.counter_name, avg(p.value), count(*) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id group by k.kernel_name, p.counter_name"""
try:
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        print(r[0][:60], r[1], round(r[2],1), r[3])
except Exception as e:
    print("views:", [t for t in tabs if 'counter' in t.lower() or 'pmc' in t.lower()], e)
PY
find $OUT -name '*.db' -delete
