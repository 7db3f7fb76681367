#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts per access width (GPU box): two separate --pmc passes over tools/pmc_calib
# -> gpurun_out/pmc_calib.json  {kernel: {FETCH_SIZE_KiB, WRITE_SIZE_KiB, bytes, fetch_factor, write_factor}}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_calib; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
[ -x tools/pmc_calib ] || hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o s -- $REPO/tools/pmc_calib > $OUT/$C.log 2>&1 )
done
python3 - <<PY
import sqlite3, glob, json, re
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob("$OUT/%s/**/*.db" % c, recursive=True)
    if not dbs: continue
    cur = sqlite3.connect(dbs[0]).cursor()
    for name, ctr, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "calib" not in name: continue
        m = re.search(r"calib_(read|write)<(.*), *(\d), *\d+>\(", name)
        width = None
        if m: width = "16B" if "4" in m.group(2) else ("8B" if "2" in m.group(2) else "4B")      # float / float vector[2] / float vector[4]
        key = ("%s %s per lane %s" % (m.group(1), width, "non-temporal" if m.group(3) == "1" else "temporal")) if m else name.split("(")[0]
        res.setdefault(key, {"kernel": name[:100], "bytes": 1 << 30})[ctr + "_KiB"] = avg
for k, v in res.items():
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if v.get(c + "_KiB"): v[c.split("_")[0].lower() + "_bytes_over_counter"] = v["bytes"] / (v[c + "_KiB"] * 1024)
json.dump(res, open("$REPO/gpurun_out/pmc_calib.json", "w"), indent=1, sort_keys=True)
for k in sorted(res): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in res[k].items() if a != "kernel"})
PY
find $OUT -name '*.db' -delete
