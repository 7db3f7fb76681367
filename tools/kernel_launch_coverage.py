#!/usr/bin/env python3
"""Which kernel instantiations of the library does the GPU tier's suite launch?  Runs the suite on the emulated kernels (tests/hipcpu) with
HIPCPU_LAUNCH_LOG set — every launch records the symbol behind its function pointer — and compares with the kernels the compiler emits for gfx950
(profiles/rNN_kernel_resources.csv, tools/kernel_resources.py).  No GPU needed.
usage: tools/kernel_launch_coverage.py RESOURCES.csv OUT.json [pytest args ...]"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    res, out = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:] or ["tests", "-n", "6"]
    logdir = os.environ.get("KLC_LOG_DIR") or tempfile.mkdtemp()       # KLC_LOG_DIR: accumulate the launches of several runs (default pass, heavy pass, ...)
    os.makedirs(logdir, exist_ok=True)
    log = os.path.join(logdir, "launches")
    env = dict(os.environ, JRC_EMULATE="1", OMP_NUM_THREADS="1", HIPCPU_LAUNCH_LOG=log)
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "--timeout", "1800"] + extra, cwd=ROOT, env=env, capture_output=True, text=True)
    tail = [l for l in r.stdout.splitlines() if " passed" in l or " failed" in l][-1:]
    launched = set()
    for f in glob.glob(log + ".*"):
        launched.update(open(f).read().split())
    names = subprocess.run(["c++filt"], input="\n".join(sorted(launched)), capture_output=True, text=True).stdout.splitlines()
    norm = lambda n: n.replace("HIP_vector_type<float, 2u>", "float2").replace("HIP_vector_type<float, 4u>", "float4")
    launched_names = set(norm(n).split("(")[0].replace("void ", "").strip() for n in names)
    rows = list(csv.DictReader(open(res)))
    all_k = sorted(set(r_["kernel"].split("(")[0].replace("void ", "").strip() for r_ in rows))
    hit = [k for k in all_k if k in launched_names]
    miss = [k for k in all_k if k not in launched_names]
    rec = {"suite": " ".join(extra), "suite_result": tail[0].strip("= ") if tail else "", "kernel_instantiations": len(all_k), "launched_by_the_suite": len(hit),
           "never_launched": miss}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("suite_result", "kernel_instantiations", "launched_by_the_suite")}), len(miss), "never launched")


if __name__ == "__main__":
    main()
