#!/usr/bin/env python3
"""Kernel-variant experiments: builds libjrc_hip.so variants with extra -D flags for one source file (chain.hip unless given) into
gr-mimo-ofdm-jrc_amd/lib/variants/<name>/ (CPU side: `build`), and benches each of them on the GPU box (`run`), one bench.py child per
variant (JRC_LIB_PATH selects the library).

  python tools/ra_variants.py build name1:"-DX=1 -DY" name2:radar.hip:"-DZ=2"      # here (hipcc cross-compiles)
  python tools/ra_variants.py run [--config B]                           # on the GPU box: every variant found + the default library
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd", "lib", "variants")


def build(specs):
    import importlib
    jb = importlib.import_module("gr-mimo-ofdm-jrc_amd.build")
    jb.build()
    for spec in specs:
        name, _, defs = spec.partition(":")
        src = "chain.hip"
        if defs.split(":")[0].endswith(".hip"):            # name:file.hip:defs
            src, _, defs = defs.partition(":")
        d = os.path.join(VDIR, name)
        os.makedirs(d, exist_ok=True)
        obj = os.path.join(d, os.path.splitext(src)[0] + ".o")
        cmd = [jb.hipcc()] + jb.HIPCC_FLAGS + jb.EXTRA_FLAGS.get(src, []) + defs.split() + ["-c", os.path.join(jb.CSRC, src), "-o", obj]
        subprocess.check_call(cmd)
        objs = [os.path.join(jb.OBJDIR, os.path.splitext(s)[0] + ".o") for s in jb.SOURCES if s != src] + [obj]
        subprocess.check_call([jb.hipcc(), "--offload-arch=" + jb.ARCH, "-shared", "-fPIC", "-o", os.path.join(d, "libjrc_hip.so")] + objs)
        open(os.path.join(d, "defs.txt"), "w").write(defs + "\n")
        print("built", name, defs)


def run(config, repeat):
    names = ["(default)"] + (sorted(os.listdir(VDIR)) if os.path.isdir(VDIR) else [])
    for rep in range(repeat):
        for n in names:
            env = dict(os.environ)
            if n != "(default)":
                env["JRC_LIB_PATH"] = os.path.join(VDIR, n, "libjrc_hip.so")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--no-cpu-baseline", "--no-secondary", "--steps", "40",
                                "--windows", "3", "--oracle-frames", "2"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", r.stderr[-300:])
                continue
            j = json.loads(line[-1])
            defs = open(os.path.join(VDIR, n, "defs.txt")).read().strip() if n != "(default)" else ""
            print("%-14s fused %.4f ms  frac %.3f  step %.4f ms  ok %s   %s" % (n, j["kernels_ms"]["range_angle_fused"], j["roofline"]["frac"],
                                                                                  j["windows"]["ms_per_step_median"], j["check"]["ok"], defs), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        cfg = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "B"
        rep = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 1
        run(cfg, rep)
