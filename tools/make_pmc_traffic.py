#!/usr/bin/env python3
"""Writes pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh: HBM bytes per launch of the
dominant kernel (range_angle_fused_kernel, or range_angle_wide_kernel at fft_len 1024), stamped with the hash of the kernel sources it was measured at, the kernel's
full name as rocprofv3 reports it, and the tag of the profile set.  bench.py reports `roofline.traffic` from this file only
when the stamp equals the hash of the tree it runs in (else traffic: null, traffic_stale: true).

usage: make_pmc_traffic.py OUT.json TAG B:frames:fetch.db:write.db [D:frames:fetch.db:write.db]"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jrc_amd  # noqa: E402,F401
from jrc_amd import build as jb  # noqa: E402


def avg_kib(db, counter):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like "
         "'%range_angle_%_kernel%' group by kernel_name")
    rows = list(cur.execute(q, (counter,)))
    rows.sort(key=lambda r: -r[2])
    return rows[0]


def main():
    out_path, tag = sys.argv[1], sys.argv[2]
    out = {"_comment": "HBM bytes per launch of range_angle_fused_kernel / range_angle_wide_kernel from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                       "runs; KiB per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 correction).",
           "source_hash": jb.source_hash(), "profile_set": tag}
    for spec in sys.argv[3:]:
        cfg, frames, fdb, wdb = spec.split(":")
        kname, fetch, n = avg_kib(fdb, "FETCH_SIZE")
        _, write, _ = avg_kib(wdb, "WRITE_SIZE")
        out[cfg] = {"frames_per_launch": int(frames), "kernel": kname, "dispatches": n, "fetch_kib": fetch, "write_kib": write,
                    "hbm_bytes_per_launch": int(round((2.0 * fetch + write) * 1024))}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
