#!/usr/bin/env python3
"""Which kernel files compile to byte-identical DEVICE code in two revisions?  Each revision's gr-mimo-ofdm-jrc_amd/csrc (+ include/jrc.h) is exported to
the same scratch path (one after the other, so that path-derived compilation-unit ids agree), every .hip is compiled with the library's flags plus
--offload-device-only, and the outputs are compared byte for byte.  Needs no GPU.
usage: tools/device_code_diff.py REV_A [REV_B|WORKTREE] [OUT.json]     ->  one line per file on stdout, JSON summary last (and into OUT.json)"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def export(rev, dst):
    os.makedirs(os.path.join(dst, "gr-mimo-ofdm-jrc_amd", "csrc"))
    os.makedirs(os.path.join(dst, "include"))
    if rev == "WORKTREE":
        for f in os.listdir(os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd", "csrc")):
            shutil.copy(os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd", "csrc", f), os.path.join(dst, "gr-mimo-ofdm-jrc_amd", "csrc", f))
        shutil.copy(os.path.join(ROOT, "include", "jrc.h"), os.path.join(dst, "include", "jrc.h"))
        return
    names = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", rev, "gr-mimo-ofdm-jrc_amd/csrc/"], capture_output=True, text=True, check=True).stdout.split()
    for n in names + ["include/jrc.h"]:
        data = subprocess.run(["git", "-C", ROOT, "show", "%s:%s" % (rev, n)], capture_output=True, check=True).stdout
        open(os.path.join(dst, n), "wb").write(data)


def device_hashes(rev, scratch):
    from jrc_amd import build as jb
    if os.path.exists(scratch):
        shutil.rmtree(scratch)
    export(rev, scratch)
    csrc = os.path.join(scratch, "gr-mimo-ofdm-jrc_amd", "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))

    def one(f):
        out = os.path.join(csrc, f + ".devcode")
        cmd = [jb.hipcc()] + jb.HIPCC_FLAGS + jb.EXTRA_FLAGS.get(f, []) + ["--offload-device-only", "-c", os.path.join(csrc, f), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            return f, "compile failed: " + r.stderr[-300:]
        return f, hashlib.sha256(open(out, "rb").read()).hexdigest()[:16]
    with ThreadPoolExecutor(max_workers=4) as ex:
        out = dict(ex.map(one, files))
    out["__source_hash__"] = jb.source_hash(csrc, os.path.join(scratch, "include", "jrc.h"))
    return out


def main():
    a = sys.argv[1]
    b = sys.argv[2] if len(sys.argv) > 2 else "WORKTREE"
    scratch = os.path.join(tempfile.gettempdir(), "jrc_device_code_diff")
    ha = device_hashes(a, scratch)
    hb = device_hashes(b, scratch)
    sha, shb = ha.pop("__source_hash__"), hb.pop("__source_hash__")
    same, differ = [], []
    for f in sorted(set(ha) | set(hb)):
        x, y = ha.get(f, "absent"), hb.get(f, "absent")
        (same if x == y else differ).append(f)
        print("%-16s %s  %s  %s" % (f, x, y, "IDENTICAL device code" if x == y else "differs"))
    from jrc_amd import build as jb
    rec = {"a": a, "b": b, "identical_device_code": same, "differs": differ, "hashes_a": ha, "hashes_b": hb,
           "flags": "build.HIPCC_FLAGS + EXTRA_FLAGS + --offload-device-only", "compiler": subprocess.run([jb.hipcc(), "--version"], capture_output=True, text=True).stdout.splitlines()[0]}
    rec["source_hash_a"], rec["source_hash_b"] = sha, shb
    if b == "WORKTREE":
        rec["source_hash_of_worktree"] = jb.source_hash()
    print(json.dumps(rec))
    if len(sys.argv) > 3:
        json.dump(rec, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
