"""CPU tier: the evidence files under profiles/ that the documents cite.

* no tracked JSON record under profiles/ is empty (round 5 committed `{}` as `r05_flowgraph_parity.json` while DESIGN.md cited it);
* a stamped full-suite record (`profiles/rNN_gpu_suite.json`, written by tools/final_check.sh through tools/stamp_suite.py) says green AND carries
  build.source_hash() of the tree it sits in — a kernel-source change after the last full `-m gpu` run makes this test fail until the suite has
  been re-run and re-stamped (rule of the round: no kernel-file commit after the last stamped full run);
* tools/stamp_suite.py reads pytest's summary line and the file order of a verbose log."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def test_no_tracked_json_record_is_empty():
    empty = []
    for p in sorted(glob.glob(os.path.join(PROFILES, "*.json"))):
        try:
            d = json.loads(open(p).read().strip().splitlines()[-1]) if open(p).read().strip() else None
        except ValueError:
            d = json.load(open(p))
        if not d:
            empty.append(os.path.basename(p))
    assert not empty, "empty evidence files: %s" % empty


def test_round5_flowgraph_parity_record_is_the_real_run():
    d = json.load(open(os.path.join(PROFILES, "r05_flowgraph_parity.json")))
    assert len(d) >= 20 and any(k.startswith("comm/") for k in d) and any(k.startswith("radar/") for k in d)
    assert "eq_out" in json.dumps(d) or "metric" in json.dumps(d)


def _latest_suite_record():
    recs = sorted(glob.glob(os.path.join(PROFILES, "r[0-9][0-9]*_gpu_suite.json")))
    return recs[-1] if recs else None


def test_stamped_full_suite_record_matches_this_tree():
    rec = _latest_suite_record()
    if rec is None:
        pytest.skip("no stamped full-suite record under profiles/ (the GPU was closed to this repository in round 6: tools/final_check.sh could not run)")
    from jrc_amd import build as jb
    d = json.load(open(rec))
    assert d["rc"] == 0 and d["failed"] == 0 and d["errors"] == 0 and d["passed"] >= 800, d
    tier0 = ("tests/test_gpu_blocks.py", "tests/test_gpu_chain.py", "tests/test_gpu_comm.py", "tests/test_gpu_tsim.py")
    order = d["file_order"]
    assert order.index("tests/test_bench_launch.py") > max(order.index(f) for f in tier0)          # parity first, process-spawning tests last
    assert d["source_hash"] == jb.source_hash(), \
        "%s was stamped at kernel sources %s, the tree is at %s: re-run tools/final_check.sh on the GPU box" % (os.path.basename(rec), d["source_hash"], jb.source_hash())


def test_stamped_emulated_suite_record_matches_this_tree():
    """profiles/rNN_emulated_suite.json (tools/emulated_suite.sh): every pass green, no sanitizer report, stamped at the kernel sources of this tree"""
    recs = sorted(glob.glob(os.path.join(PROFILES, "r[0-9][0-9]*_emulated_suite.json")))
    if not recs:
        pytest.skip("no stamped emulated-suite record under profiles/")
    from jrc_amd import build as jb
    d = json.load(open(recs[-1]))
    assert "NOT a device run" in d["what"]
    assert set(d["passes"]) >= {"default", "asan_ubsan"}
    for name, p in d["passes"].items():
        assert p["rc"] == 0 and p["failed"] == 0 and p["errors"] == 0 and p["sanitizer_reports"] == 0, (name, p)
    assert d["passes"]["default"]["passed"] >= 750
    assert d["source_hash"] == jb.source_hash(), \
        "%s was stamped at kernel sources %s, the tree is at %s: re-run tools/emulated_suite.sh" % (os.path.basename(recs[-1]), d["source_hash"], jb.source_hash())


def test_device_code_record_says_round5s_kernels_are_unchanged_and_is_of_this_tree():
    """profiles/r06_device_code_vs_r05.json (tools/device_code_diff.py 02f2fef WORKTREE): every kernel file round 5 ended on compiles to byte-identical
    DEVICE code in this tree — the defaults launch round 5's machine code, whose GPU runs stand — and the only file with new device code is
    onchip.hip (opt-in kernels); the record is of THIS tree's kernel sources"""
    p = os.path.join(PROFILES, "r06_device_code_vs_r05.json")
    if not os.path.exists(p):
        pytest.skip("no device-code record")
    from jrc_amd import build as jb
    d = json.load(open(p))
    assert d["differs"] == ["onchip.hip"] and "onchip.hip" not in d["hashes_a"]
    assert set(d["identical_device_code"]) == {"chain.hip", "codec.hip", "comm.hip", "ctx.hip", "estimator.hip", "feed.hip", "fft.hip", "radar.hip", "sync.hip", "tsim.hip"}
    assert d["source_hash_of_worktree"] == jb.source_hash(), "kernel sources changed since the record was taken: re-run tools/device_code_diff.py 02f2fef WORKTREE profiles/r06_device_code_vs_r05.json"


def test_stamp_suite_parses_a_verbose_log():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stamp_suite
    log = "\n".join([
        "tests/test_gpu_blocks.py::test_a[1] PASSED   [  0%]",
        "tests/test_gpu_blocks.py::test_b PASSED   [  1%]",
        "tests/test_gpu_chain.py::test_c SKIPPED (x)   [  2%]",
        "tests/test_bench_launch.py::test_two PASSED   [100%]",
        "============ slowest 12 durations ============",
        "===== 829 passed, 2 skipped, 331 deselected in 91.42s (0:01:31) =====",
    ])
    r = stamp_suite.parse_log(log)
    assert (r["passed"], r["skipped"], r["failed"], r["deselected"], r["seconds"]) == (829, 2, 0, 331, 91.42)
    assert r["file_order"] == ["tests/test_gpu_blocks.py", "tests/test_gpu_chain.py", "tests/test_bench_launch.py"]
    r = stamp_suite.parse_log("tests/test_gpu_blocks.py::test_a FAILED\n=== 1 failed, 3 passed in 4.00s ===")
    assert r["failed"] == 1 and r["passed"] == 3


def test_suite_order_puts_parity_first_and_process_spawning_last():
    """tests/conftest.py pytest_collection_modifyitems, checked on the real collection"""
    import subprocess
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"], capture_output=True, text=True, cwd=ROOT, timeout=300)
    ids = [l for l in r.stdout.splitlines() if "::" in l]
    assert len(ids) >= 800
    files = []
    for i in ids:
        f = i.split("::")[0]
        if not files or files[-1] != f:
            files.append(f)
    first_spawn = min(k for k, i in enumerate(ids) if i.startswith(("tests/test_gpu_switches.py", "tests/test_bench_launch.py")) or "pacing_word" in i or "experiment_switches" in i)
    parity = [k for k, i in enumerate(ids) if i.startswith(("tests/test_gpu_blocks.py", "tests/test_gpu_chain.py", "tests/test_gpu_comm.py", "tests/test_gpu_tsim.py",
                                                             "tests/test_gpu_flowgraph_parity.py", "tests/test_gpu_sync.py", "tests/test_gpu_codec.py"))
              and "pacing_word" not in i]
    assert max(parity) < first_spawn
    assert files[-2:] == ["tests/test_bench_launch.py", "tests/test_gpu_unvetted.py"]      # rank launches, then the kernels that have never run on hardware
    assert ids[0].startswith(("tests/test_golden", "tests/test_gpu_blocks.py"))
