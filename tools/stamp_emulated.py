#!/usr/bin/env python3
"""gpurun_out/<TAG>_emulated_{default,heavy,asan_ubsan}.log -> gpurun_out/<TAG>_emulated_suite.json, stamped with build.source_hash() of the tree
(tests/test_evidence_stamps.py holds the tracked copy under profiles/ against the tree's hash).  usage: stamp_emulated.py TAG RC_DEFAULT RC_HEAVY RC_ASAN RC_REVERSE RC_SHUFFLE RC_HEAVY_D_BATCHES"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stamp_suite  # noqa: E402


def main():
    tag, rcs = sys.argv[1], [int(x) for x in sys.argv[2:8]]
    from jrc_amd import build as jb
    rec = {"what": "the -m gpu tests on the library's kernel sources built for the host CPU under the emulated execution model of tests/hipcpu "
                   "(no GPU: the pool was closed to this repository in round 6) - NOT a device run", "source_hash": jb.source_hash(),
           "when": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "host": os.uname().machine, "passes": {}}
    for name, rc in zip(("default", "heavy", "asan_ubsan", "reverse", "shuffle", "heavy_config_d_batches"), rcs):
        p = os.path.join(ROOT, "gpurun_out", "%s_emulated_%s.log" % (tag, name))
        if not os.path.exists(p):
            continue
        text = open(p, errors="replace").read()
        r = stamp_suite.parse_log(text)
        r.pop("file_order", None)
        r["rc"] = rc
        r["sanitizer_reports"] = text.count("ERROR: AddressSanitizer") + text.count("runtime error:")
        rec["passes"][name] = r
    dst = os.path.join(ROOT, "gpurun_out", "%s_emulated_suite.json" % tag)
    json.dump(rec, open(dst, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
