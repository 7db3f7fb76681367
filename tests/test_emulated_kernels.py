"""CPU tier: the `-m gpu` tests against the library's own kernel sources on a machine WITHOUT a GPU (tests/hipcpu: the kernels built for the host
CPU under an emulated wavefront / workgroup execution model; tests/hipcpu/README.md says what that shows and what it does not).

Round 6 had no GPU — the pool was closed to this repository before the round's first call — and the reference ships no tests: without this file the
round would have no execution of the kernels at all.  With it, every CPU-tier run (the driver's `-m "not gpu"` run included) executes the GPU tier's
parity tests — the same test functions, the same oracle, the same tolerances, bit-exact where they ask for it — in a child pytest in emulation mode:

* the whole suite minus what needs the device itself (bench.py rank launches, tests that re-import the library in child processes, timing guards) and
  minus the shapes that take the emulation minutes (tools/emulated_suite.sh runs those);
* a second pass over the cross-lane-heavy files with the order of waves and lanes REVERSED: results must not depend on the order in which
  the waves of a workgroup (and the lanes between two rendezvous) happen to run — a missing barrier does;
* the emulation's own semantics (DPP control words, permlane swaps, shuffles, ballots under divergence, barriers, refused launches) against values
  worked out by hand.

This is a checker, like oracle/: nothing under gr-mimo-ofdm-jrc_amd/ refers to it (tests/test_abi_symbols.py)."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCPU = os.path.join(ROOT, "tests", "hipcpu")


def _have_clang():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hipcpu import build as eb
    try:
        eb.compiler()
        return True
    except RuntimeError:
        return False


pytestmark = pytest.mark.skipif(not _have_clang() or os.uname().machine != "x86_64", reason="the emulation needs clang++ and an x86-64 host")


def _workers():
    return max(1, min(7, (os.cpu_count() or 2) - 1))


def _run_emulated(args, env=None, timeout=2400):
    e = dict(os.environ, JRC_EMULATE="1", OMP_NUM_THREADS="1")
    e.pop("JRC_LIB_PATH", None)
    e.update(env or {})
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "-n", str(_workers()), "--timeout", "900"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=e, timeout=timeout)
    tail = (r.stdout + r.stderr)[-4000:]
    m = None
    for line in reversed(r.stdout.splitlines()):
        if re.search(r"\b(passed|failed|error)\b.* in [0-9.]+s", line):
            m = line
            break
    assert m is not None, tail
    counts = {k: 0 for k in ("passed", "failed", "skipped", "error")}
    for n, what in re.findall(r"(\d+) (passed|failed|skipped|errors?)", m):
        counts["error" if what.startswith("error") else what] = int(n)
    counts["seconds"] = float(re.search(r" in ([0-9.]+)s", m).group(1))
    return r, counts, tail


def test_emulation_semantics_selftest(tmp_path):
    from hipcpu import build as eb
    exe = str(tmp_path / "selftest")
    r = subprocess.run([eb.compiler(), "-x", "c++", "-std=c++17", "-O1", "-g", "-I" + os.path.join(HIPCPU, "include"), "-Wno-unused-value",
                        os.path.join(HIPCPU, "selftest.cc"), os.path.join(HIPCPU, "hipcpu_runtime.cc"), "-o", exe, "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    for sched in ("natural", "reverse", "shuffle:11"):
        r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, HIPCPU_SCHEDULE=sched), timeout=120)
        assert r.returncode == 0 and "selftest ok" in r.stdout, (sched, r.stdout[-2000:], r.stderr[-2000:])


def test_the_emulated_sources_are_the_tracked_sources():
    """what the emulation compiles is each tracked .hip with exactly three kinds of line rewritten — `extern __shared__ T x[];` (the running workgroup's
    dynamic LDS), `asm volatile("s_sleep" / "s_waitcnt" / "")` (scheduling hints without a data effect) and static `__shared__` declarations, which get a
    call appended that fills the object with 0xFF once per workgroup (LDS is garbage at workgroup start on the device, not zero) — and nothing else"""
    from hipcpu import build as eb
    csrc = os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd", "csrc")
    n_dyn = n_asm = n_static = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        a = open(os.path.join(csrc, f)).read().splitlines()
        b = eb.rewrite(open(os.path.join(csrc, f)).read()).splitlines()
        assert len(a) == len(b), f
        for x, y in zip(a, b):
            if x == y:
                continue
            if "__shared__" in x and "extern" not in x:
                head, _, tail = y.partition("; ::hipcpu::poison_static_lds(")
                assert tail and x.startswith(head.rstrip()) or head.split() == x[:x.index(";")].split(), (f, x, y)
                rest = y
                while "::hipcpu::poison_static_lds(" in rest:
                    i = rest.index("::hipcpu::poison_static_lds(")
                    j = rest.index(");", i) + 2
                    rest = rest[:i] + rest[j:]
                assert rest.split() == x.split(), (f, x, y)          # the declaration and its comment are untouched
                n_static += 1
            elif "extern __shared__" in x:
                assert "::hipcpu::dyn_lds()" in y and "extern" not in y, (f, x, y)
                n_dyn += 1
            else:
                assert "asm volatile" in x and "asm" not in y and "((void)0);" in y, (f, x, y)
                assert any(h in x for h in ('"s_sleep 1"', '"s_waitcnt vmcnt(1)"', '""')), (f, x)
                n_asm += 1
    assert n_dyn >= 18 and n_asm == 5 and n_static >= 50, (n_dyn, n_asm, n_static)


def test_gpu_tier_under_emulation():
    """the GPU tier's tests on the emulated kernels: nothing may fail, and the count says the parity files really ran"""
    r, c, tail = _run_emulated(["tests"])
    if r.returncode != 0 and 0 < c["failed"] + c["error"] <= 3:
        # six workers on a loaded host: the block layer's tests start threads and hold age bounds in wall-clock time.  Up to three failures are run
        # again ALONE (no workers beside them); they must pass then, and the record says that a second attempt was needed and for what.
        again = [l.split(" ")[1] for l in r.stdout.splitlines() if l.startswith(("FAILED ", "ERROR ")) and "::" in l]
        r2, c2, tail2 = _run_emulated(again + ["-n", "0"])
        c["rerun_alone"] = {"tests": again, "passed": c2["passed"], "failed": c2["failed"] + c2["error"]}
        assert r2.returncode == 0 and c2["failed"] == 0 and c2["error"] == 0 and c2["passed"] == len(again), (again, tail, tail2)
        c["failed"] = c["error"] = 0
        r = r2
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(c, open(os.path.join(ROOT, "gpurun_out", "emulated_suite_cpu_tier.json"), "w"))
    except OSError:
        pass
    assert r.returncode == 0 and c["failed"] == 0 and c["error"] == 0, tail
    assert c["passed"] >= 850, c                      # 878 at the end of round 6 (931 collected; heavy shapes and device-only tests skipped)


def test_results_do_not_depend_on_the_order_waves_and_lanes_are_run_in():
    """the same kernels with the scheduler walking waves and lanes BACKWARDS: a kernel whose waves communicate through LDS without a barrier, or
    whose lanes rely on running in index order, gives different results — the tests' own assertions catch it.  (This pass over the whole suite is
    what found the detect-only chain re-reading its shared running maximum after lane 0's store with nothing in between: fine on the device, marked
    JRC_LOCKSTEP() since.  tools/emulated_suite.sh runs the whole suite both backwards and in a shuffled order; here: the cross-lane-heavy files.)"""
    files = ["tests/test_gpu_blocks.py", "tests/test_gpu_chain_modes.py", "tests/test_gpu_comm.py", "tests/test_gpu_sync.py",
             "tests/test_gpu_flowgraph.py", "tests/test_golden_fixtures.py", "tests/test_golden_flowgraphs.py"]
    r, c, tail = _run_emulated(files + ["-k", "not 174080 and not long_bursts and not B-40 and not D-9 and not A-300"], env={"HIPCPU_SCHEDULE": "reverse"})
    assert r.returncode == 0 and c["failed"] == 0 and c["error"] == 0 and c["passed"] >= 280, (c, tail)
