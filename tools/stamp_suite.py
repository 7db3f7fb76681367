#!/usr/bin/env python3
"""Turns the log of a full `python -m pytest tests -m gpu -v` run into the stamped record profiles/rNN_gpu_suite.json
(VERDICT r5 item 3): {passed, failed, skipped, seconds, rc, source_hash, file_order}.  `source_hash` is build.source_hash() of the tree the
suite ran in — tests/test_evidence_stamps.py (CPU tier) fails when the tracked record's stamp is not the hash of the tree, so a kernel
change after the last full run cannot go unnoticed.
usage: tools/stamp_suite.py LOG RC OUT.json"""
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_log(text):
    out = {"passed": 0, "failed": 0, "skipped": 0, "errors": 0, "deselected": 0, "seconds": None}
    for line in reversed(text.splitlines()):
        if re.search(r"\b(passed|failed|error|errors|skipped|no tests ran)\b.* in [0-9.]+s", line):
            for n, what in re.findall(r"(\d+) (passed|failed|skipped|deselected|errors?)", line):
                out["errors" if what.startswith("error") else what] = int(n)
            out["seconds"] = float(re.search(r" in ([0-9.]+)s", line).group(1))
            break
    order = []
    for m in re.finditer(r"^(tests/[\w/]+\.py)::", text, re.M):
        if not order or order[-1] != m.group(1):
            order.append(m.group(1))
    out["file_order"] = order
    return out


def main():
    log, rc, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    from jrc_amd import build as jb
    rec = parse_log(open(log, errors="replace").read())
    rec.update(rc=rc, source_hash=jb.source_hash(), command="python3 -m pytest tests -m gpu -v -x", when=time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()))
    try:
        import torch
        rec["device"] = torch.cuda.get_device_name(0) if torch.cuda.is_available() else None
    except Exception:
        rec["device"] = None
    with open(dst, "w") as fh:
        json.dump(rec, fh, indent=1)
        fh.write("\n")
    print(json.dumps({k: rec[k] for k in ("passed", "failed", "skipped", "errors", "seconds", "rc", "source_hash")}))


if __name__ == "__main__":
    main()
