#!/usr/bin/env python3
"""Static resource usage of every kernel of the library for gfx950, from the compiler's own report (hipcc -Rpass-analysis=kernel-resource-usage; no GPU
needed): VGPRs, AGPRs, SGPRs, scratch (spill) bytes per lane, occupancy in waves per SIMD, static LDS per workgroup.
usage: tools/kernel_resources.py OUT.csv"""
import csv
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from jrc_amd import build as jb
    csrc = jb.CSRC
    files = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    filt = "c++filt"

    def one(f):
        cmd = [jb.hipcc()] + jb.HIPCC_FLAGS + jb.EXTRA_FLAGS.get(f, []) + ["-Rpass-analysis=kernel-resource-usage", "--offload-device-only", "-c", os.path.join(csrc, f), "-o", os.devnull]
        r = subprocess.run(cmd, capture_output=True, text=True)
        rows, cur = [], None
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = {"file": f, "mangled": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
        return rows
    with ThreadPoolExecutor(max_workers=3) as ex:
        rows = [r for rs in ex.map(one, files) for r in rs]
    names = subprocess.run([filt], input="\n".join(r["mangled"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["kernel"] = re.sub(r"HIP_vector_type<float, 2u>", "float2", n)[:160]
    cols = ["file", "kernel", "VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize", "Occupancy", "LDS Size"]
    with open(sys.argv[1], "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["file", "kernel", "VGPRs", "AGPRs", "SGPRs", "scratch_bytes_per_lane", "occupancy_waves_per_SIMD", "static_LDS_bytes_per_workgroup"])
        for r in rows:
            w.writerow([r.get(c, "") for c in cols])
    spill = [r["kernel"] for r in rows if r.get("ScratchSize", 0) > 0]
    print("%d kernels, %d with scratch: %s" % (len(rows), len(spill), spill[:8]))


if __name__ == "__main__":
    main()
