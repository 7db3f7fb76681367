"""Builds libjrc_hip.so (hand-written HIP kernels + the C ABI of include/jrc.h) for gfx950 with hipcc.

In-tree build: the .so lands in gr-mimo-ofdm-jrc_amd/lib/ (git-ignored, but it travels to the GPU box).
hipcc cross-compiles without a GPU, so this runs in the CPU-only container as the "does it build" check.
"""
import concurrent.futures
import contextlib
import fcntl
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libjrc_hip.so")
ARCH = "gfx950"

SOURCES = ["ctx.hip", "radar.hip", "fft.hip", "estimator.hip", "chain.hip", "feed.hip", "comm.hip", "tsim.hip", "codec.hip", "sync.hip", "onchip.hip"]
HEADERS = ["jrc_internal.h", "radar_kernels.h", "fft_device.h", "tsim_device.h", os.path.join("..", "..", "include", "jrc.h")]

HIPCC_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               "-fno-gpu-rdc"]
# per-file extras.  chain.hip: the SLP vectorizer packs the complex butterflies into v_pk_*_f32, which on
# gfx950 run at the scalar f32 rate but cost ~100 extra v_mov and ~50 VGPRs in the fused kernel.
EXTRA_FLAGS = {"chain.hip": ["-fno-slp-vectorize"], "radar.hip": ["-fno-slp-vectorize"]}


def source_hash(csrc=None, jrc_h=None):
    """hash of everything the device code is built from (kernel sources, headers, compiler flags): profiles/pmc_traffic.json is
    stamped with it, and bench.py drops `roofline.traffic` when the stamp is not the hash of the tree it runs in.
    (csrc / jrc_h: the same hash of another checkout's sources — tools/device_code_diff.py)"""
    import hashlib
    csrc = csrc or CSRC
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    for n in names:
        h.update(n.encode())
        h.update(open(os.path.join(csrc, n), "rb").read())
    h.update(open(jrc_h or os.path.join(HERE, "..", "include", "jrc.h"), "rb").read())
    h.update(repr((HIPCC_FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension is mandatory (no CPU fallback exists)")
    return exe


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    srcp = os.path.join(CSRC, src)
    deps = [srcp, os.path.abspath(__file__)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _mtime(obj) >= max(_mtime(d) for d in deps):
        return obj, False
    cmd = [hipcc()] + HIPCC_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


_lock_depth = 0


@contextlib.contextmanager
def build_lock():
    """one builder at a time across processes (N ranks started together on a tree without the .so, pytest-xdist workers): an flock on
    lib/.build.lock; whoever waited re-checks the time stamps afterwards and finds the work done.  Re-entrant within a process."""
    global _lock_depth
    if _lock_depth:
        _lock_depth += 1
        try:
            yield
        finally:
            _lock_depth -= 1
        return
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        _lock_depth = 1
        try:
            yield
        finally:
            _lock_depth = 0
            fcntl.flock(fh, fcntl.LOCK_UN)


def build(force=False, verbose=False):
    with build_lock():
        return _build_locked(force, verbose)


def _build_locked(force, verbose):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(ch for _, ch in results)
    if rebuilt or not os.path.exists(LIB) or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(LIBDIR, "libjrc_blocks.so")
HOST_SOURCES = ["jrc_blocks.cc", "jrc_blocks_capi.cc"]


def build_host(force=False, verbose=False):
    """host-side C++ blocks (reference block interface over the C ABI) + their C test harness, with g++ against the
    stand-alone runtime (no GNU Radio in this image); links libjrc_hip.so"""
    with build_lock():
        return _build_host_locked(force, verbose)


def _build_host_locked(force, verbose):
    build()
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST_DIR, h) for h in ("jrc_blocks.h", "jrc_block_runtime.h")] + [os.path.join(HERE, "..", "include", "jrc.h")]
    if not force and os.path.exists(HOST_LIB) and _mtime(HOST_LIB) >= max(_mtime(d) for d in deps):
        return HOST_LIB
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++14", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB] + srcs + \
          ["-L" + LIBDIR, "-ljrc_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host block build failed:\n%s\n%s" % (r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    if verbose:
        print("built", HOST_LIB)
    return HOST_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_host(force="--force" in sys.argv, verbose=True)
