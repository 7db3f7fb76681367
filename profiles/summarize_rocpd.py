#!/usr/bin/env python3
"""Turns the rocpd sqlite databases that rocprofv3 writes on this image into the small CSV summaries committed
under profiles/.   usage: summarize_rocpd.py OUT.csv "header comment" stats.db [pmc1.db pmc2.db ...]"""
import sqlite3
import sys


def main():
    out_path, header, stats_db = sys.argv[1], sys.argv[2], sys.argv[3]
    out = ["# " + header, "# top_kernels (rocprofv3 --kernel-trace --stats; durations in us)",
           "name,total_calls,total_duration_us,average_us,percentage"]
    cur = sqlite3.connect(stats_db).cursor()
    for r in cur.execute("select * from top_kernels"):
        out.append(",".join('"%s"' % x if isinstance(x, str) else str(x) for x in r))
    for db in sys.argv[4:]:
        cur = sqlite3.connect(db).cursor()
        out += ["", "# PMC pass %s (separate run; FETCH_SIZE/WRITE_SIZE in KiB per dispatch; on gfx950 bytes = FETCH_SIZE x 1024 x 2.0 "
                "and WRITE_SIZE x 1024 x 1.0 for 4 / 8 / 16 B per lane alike: profiles/r04_pmc_calibration.json, tools/pmc_calib.sh)" % db.split("/")[-2],
                "kernel,counter,dispatches,avg,min,max"]
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
             "from counters_collection group by kernel_name, counter_name")
        for r in cur.execute(q):
            out.append(",".join('"%s"' % x if isinstance(x, str) else str(x) for x in r))
    open(out_path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
