"""bench.py's N>1 launch path (BASELINE.json config 5: frame stream sharded over GPUs).

CPU tier: `python bench.py --gpus N` without WORLD_SIZE starts its ranks itself; without a GPU every rank refuses to run
(there is no CPU fallback) and the parent must relay that as a non-zero exit, with no JSON line and without hanging.
GPU tier: two ranks on the one GPU of the box (--same-device, gloo) go through the whole multi-rank path — process group,
frame shards, barriers, MAX-reduce of the timings, all-gather of results and maps — and every frame must come out bit for bit
as in a single-rank run of the same frame stream."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout=300, env=None, verbose=None):
    """runs bench.py; returns (completed process, the JSON lines of its stdout).  With `verbose` = a path, bench.py writes its full record there
    and the last element of `lines` is replaced by that record (the stdout line itself is the compact form, checked on its own below)."""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    if verbose:
        args = args + ["--verbose-out", verbose]
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if verbose and r.returncode == 0 and lines:
        assert len(lines[-1]) < 6000                       # the contract's line stays short whatever the record holds
        c = json.loads(lines[-1])
        full = json.loads(open(verbose).read())
        assert c["value"] == pytest.approx(full["value"], rel=1e-5) and c["n_gpus"] == full["n_gpus"] and c["check"]["ok"] == full["check"]["ok"]
        lines[-1] = json.dumps(full)
    return r, lines


def test_self_launch_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-tier check of the failure path")
    r, lines = _run(["--gpus", "2", "--config", "A", "--frames", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary",
                     "--no-check"], timeout=300)
    assert r.returncode != 0
    assert not lines                                   # no result line from a run that did not happen
    assert "needs a GPU" in r.stderr


def test_drivers_eight_gpu_launch_line_fails_loudly_without_a_gpu():
    """the 8-GPU scaling run is launched by the driver as `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W` (backend nccl = RCCL).  Without a GPU every one of the 8 ranks
    must get as far as the device check and say "needs a GPU" — before init_process_group, so nothing waits on a rendezvous — and the launcher must
    come back non-zero, with no JSON line, well inside its limit."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-tier check of the failure path")
    import socket
    import time
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    e = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=420, env=e, cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.stderr.count("needs a GPU") >= 1 and "Traceback (most recent call last):\n  File \"%s\"" % BENCH not in r.stderr
    assert time.time() - t0 < 400


def test_contract_line_stays_short_and_parses():
    """round 5's line grew to 21 KB and the driver could not parse it.  The recorded full record of that run
    (tests/golden/bench_verbose_r05.json) through the same compaction bench.py prints: < 6000 bytes, strict JSON, headline + roofline +
    cpu_baseline + every secondary leg as {value, unit, frac_of_hbm_peak}; an 8-rank record and a record with runaway legs stay short too."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.loads(open(os.path.join(ROOT, "tests", "golden", "bench_verbose_r05.json")).read())
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, "gpurun_out/bench_verbose.json")
    assert len(line) < bench.LINE_LIMIT == 6000 and "\n" not in line
    c = json.loads(line)
    assert json.loads(json.dumps(c)) == c
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in c, k
    assert c["value"] == pytest.approx(full["value"], rel=1e-5) and c["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert c["config"]["workload"] == full["config"]["workload"]
    r = c["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert r["traffic"] == full["roofline"]["traffic"] and r["algorithmic_bytes_per_launch"] == 512 * 4227072
    assert r["chain"]["frac"] > 0 and r["chain"]["a1"]["frac"] > 0
    b = c["cpu_baseline"]
    assert b["kind"] == "port" and b["cores"] == 1 and b["value"] > 0 and b["unit"] == "frames/s" and len(b["sample"]) <= 120
    assert c["check"]["ok"] is True
    assert set(c["secondary"]) == set(full["secondary"])
    for name, leg in c["secondary"].items():
        if name == "per_block_drop_in":
            assert leg["blocks"]["mimo_ofdm_radar"]["B"][0] > 0
            continue
        assert set(leg) <= {"value", "unit", "frac_of_hbm_peak", "error"} and "value" in leg, (name, leg)
    # eight ranks: the per-rank entries are cut to what identifies a rank
    eight = dict(full, n_gpus=8, ranks=[dict(full["ranks"][0], rank=k, pid=1000 + k) for k in range(8)])
    l8 = bench.compact_line(eight)
    assert len(l8) < 6000 and [r["rank"] for r in json.loads(l8)["ranks"]] == list(range(8))
    # legs that run away (a secondary leg returning pages of prose, hundreds of legs) cannot take the headline with them
    fat = dict(full, secondary=dict(full["secondary"], **{"leg_%d" % k: {"frames_per_s": 1.0 * k, "what": "x" * 4000} for k in range(400)}))
    lf = bench.compact_line(fat)
    cf = json.loads(lf)
    assert len(lf) < 6000 and cf["value"] == c["value"] and cf["roofline"]["frac"] == c["roofline"]["frac"] and "secondary" in cf["dropped_to_fit"]
    # a NaN from a failed leg must not produce a line json.loads rejects
    bad = dict(full, secondary={"x": {"frames_per_s": float("nan"), "frac_of_hbm_peak": float("inf")}})
    assert json.loads(bench.compact_line(bad))["secondary"]["x"]["value"] is None


def test_traffic_of_another_tree_counts_only_with_identical_device_code(tmp_path, monkeypatch):
    """bench.pmc_traffic: the PMC figure of profiles/pmc_traffic.json was measured at round 5's kernel sources; this tree's sources differ (markers, host code)
    but profiles/r06_device_code_vs_r05.json shows chain / radar / estimator compiling to byte-identical device code, stamped with both trees' hashes:
    the figure is reported.  With the record's hashes not matching (another tree, a file that differs) it is withheld and flagged stale."""
    sys.path.insert(0, ROOT)
    import bench
    from jrc_amd import build as jb
    rec_path = os.path.join(ROOT, "profiles", "r06_device_code_vs_r05.json")
    if not os.path.exists(rec_path):
        pytest.skip("no device-code record")
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    rec = json.load(open(rec_path))
    if rec.get("source_hash_b") != jb.source_hash() or rec.get("source_hash_a") != pmc["source_hash"]:
        pytest.skip("the record is not of this pair of trees (test_evidence_stamps.py says so loudly)")
    got, stale = bench.pmc_traffic("B", pmc["B"]["frames_per_launch"])
    assert got == pmc["B"]["hbm_bytes_per_launch"] and not stale
    # a tree with other kernel sources and no record for it: withheld
    monkeypatch.setattr(jb, "source_hash", lambda *a, **k: "0000000000000000")
    got, stale = bench.pmc_traffic("B", pmc["B"]["frames_per_launch"])
    assert got is None and stale


def test_self_launch_is_bounded(tmp_path):
    """bench.py's launcher must not wait for its ranks for ever: with ranks that never finish (a stand-in script that
    sleeps) the launcher kills the children it started and exits 124 within --launch-timeout"""
    sys.path.insert(0, ROOT)
    import bench
    import time
    a = bench.parse(["--gpus", "2", "--launch-timeout", "3"])
    real_file = bench.__file__
    sleeper = tmp_path / "sleeper.py"
    sleeper.write_text("import time\ntime.sleep(600)\n")
    bench.__file__ = str(sleeper)
    try:
        t0 = time.time()
        rc = bench.self_launch(a, [])
        assert rc == 124 and time.time() - t0 < 30
    finally:
        bench.__file__ = real_file
    assert bench.same_device_hw_queues(8) == 2 and bench.same_device_hw_queues(2) == 4 and bench.same_device_hw_queues(64) == 1


def test_argument_surface():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(["--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert (a.gpus, a.steps, a.warmup, a.config) == (4, 7, 2, "B")
    a = bench.parse([])
    assert a.gpus == 1 and a.windows >= 5


COMMON = ["--config", "A", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--windows", "2", "--prewarm-seconds", "0"]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    F = 48
    one = str(tmp_path / "one.npz")
    two = str(tmp_path / "two.npz")
    r1, l1 = _run(COMMON + ["--frames", str(2 * F), "--distinct", str(2 * F), "--dump", one, "--dump-maps", str(2 * F), "--oracle-frames", "4"],
                  verbose=str(tmp_path / "v1.json"))
    assert r1.returncode == 0, r1.stderr[-2000:]
    j1 = json.loads(l1[-1])
    assert j1["n_gpus"] == 1 and j1["check"]["ok"] and j1["config"]["launcher"] == "direct"
    assert len(j1["ranks"]) == 1 and j1["collective_world"] == 1 and j1["backend"] is None and j1["distinct_devices"] == 1
    r2, l2 = _run(COMMON + ["--gpus", "2", "--same-device", "--backend", "gloo", "--frames", str(F), "--distinct", str(F), "--dump", two,
                            "--dump-maps", "3", "--oracle-frames", "4", "--gather-results", "--gather-maps", "2"], verbose=str(tmp_path / "v2.json"))
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert len(l2) == 1                                # ONE JSON line, from rank 0
    j2 = json.loads(l2[-1])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["check"]["ok"] and j2["check"]["ok_all_ranks"]
    assert j2["config"]["launcher"] == "self-launched children" and j2["config"]["gather_results"] and j2["config"]["gather_maps_per_gpu"] == 2
    assert j2["value"] > 0 and abs(j2["value"] - 2 * F * 3 / (j2["ms_per_step"] * 3e-3)) < 1e-6 * j2["value"]
    # the line proves who took part: one entry per rank (device identity, own clock), the backend and the size of the process group
    rk = j2["ranks"]
    assert [r["rank"] for r in rk] == [0, 1] and j2["collective_world"] == 2 and j2["backend"] == "gloo"
    assert all(r["device_name"] and r["pci_bus_id"] and r["pid"] > 0 and r["ms_per_step_window0"] > 0 for r in rk)
    assert rk[0]["pid"] != rk[1]["pid"] and rk[0]["pci_bus_id"] == rk[1]["pci_bus_id"] and j2["distinct_devices"] == 1   # --same-device
    assert max(r["ms_per_step_window0"] for r in rk) <= j2["ms_per_step"] * (1 + 1e-9)      # the line's time is the MAX over ranks
    assert 0.3 < j2["value_over_n_times_slowest_rank"] <= 1.0 + 1e-9
    a, b = np.load(one), np.load(two)
    # rank 0 owns frames [0, F), rank 1 frames [F, 2F) of the same seeded stream: gathered in frame order
    assert np.array_equal(a["results"], b["results"])
    assert np.array_equal(a["chanest"], b["chanest"])
    want_maps = np.concatenate([a["maps"][0:3], a["maps"][F:F + 3]])
    assert np.array_equal(want_maps, b["maps"])


EIGHT_ON_ONE = os.environ.get("JRC_TEST_EIGHT_RANKS_ON_ONE_GPU", "") not in ("", "0")


@pytest.mark.gpu
@pytest.mark.skipif(not EIGHT_ON_ONE, reason="8 processes on ONE device ended in `HW Exception ... GPU Hang` on the driver's box in round 5 (passed on the builder's); "
                    "the GPU was closed to this repository in round 6 before the root cause could be studied on hardware (tools/hang_bisect.sh is the "
                    "prepared study), so the unverified case is opt-in: JRC_TEST_EIGHT_RANKS_ON_ONE_GPU=1.  Two ranks on one device (above) ran green "
                    "on the driver's box in rounds 3-5; world 8 with ragged and empty shards is covered over gloo in tests/test_multi_gpu_gloo.py")
def test_eight_ranks_with_ragged_shards_equal_one_rank(tmp_path):
    """BASELINE config 5's world size on the one GPU there is: 8 ranks (--same-device, gloo) split ONE stream of 203 frames into the ragged
    blocks of shard.frame_shard (26 / 25 frames), gather the per-frame records and two maps per rank inside every step, and the dump —
    records, channel estimates, maps, in frame order — must equal the single-rank run of the same 203 frames bit for bit"""
    sys.path.insert(0, ROOT)
    from jrc_amd import shard
    M, W = 203, 8
    one, eight = str(tmp_path / "one.npz"), str(tmp_path / "eight.npz")
    r1, l1 = _run(COMMON + ["--frames", str(M), "--distinct", str(M), "--dump", one, "--dump-maps", str(M), "--oracle-frames", "4"], timeout=240)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r8, l8 = _run(COMMON + ["--gpus", str(W), "--same-device", "--backend", "gloo", "--stream-frames", str(M), "--distinct", str(M), "--dump", eight,
                            "--dump-maps", "2", "--oracle-frames", "4", "--gather-results", "--gather-maps", "2", "--launch-timeout", "200"],
                  timeout=240, verbose=str(tmp_path / "v8.json"), env={"JRC_LOG_CALLS": os.environ.get("JRC_LOG_CALLS", "0")})
    assert r8.returncode == 0, r8.stderr[-3000:]
    assert len(l8) == 1
    j = json.loads(l8[-1])
    sizes = shard.shard_sizes(M, W)
    assert sorted(set(sizes)) == [25, 26] and sum(sizes) == M                 # ragged
    assert j["n_gpus"] == W and j["scaling"] == "strong" and j["collective_world"] == W and j["backend"] == "gloo" and j["distinct_devices"] == 1
    assert j["check"]["ok"] and j["check"]["ok_all_ranks"]
    assert j["config"]["stream_frames_per_step"] == M and j["config"]["gather_results"] and j["config"]["gather_maps_per_gpu"] == 2
    assert [r["rank"] for r in j["ranks"]] == list(range(W)) and len(set(r["pid"] for r in j["ranks"])) == W
    assert abs(j["value"] - M * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    a, b = np.load(one), np.load(eight)
    assert a["results"].shape[0] == b["results"].shape[0] == M
    assert np.array_equal(a["results"], b["results"]) and np.array_equal(a["chanest"], b["chanest"])
    firsts = [shard.frame_shard(M, r, W)[0] for r in range(W)]
    assert np.array_equal(np.concatenate([a["maps"][f:f + 2] for f in firsts]), b["maps"])


@pytest.mark.gpu
def test_two_ranks_on_one_device_without_the_test_flag_is_refused():
    """an N-GPU line whose ranks share a physical device is not an N-GPU measurement: without --same-device (where every rank is
    pinned to GPU 0 on purpose) bench.py must exit non-zero.  On this one-GPU box LOCAL_RANK 1 has no device of its own, so the ranks are
    started by hand with LOCAL_RANK 0 twice - what a mis-set launcher would do."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH] + COMMON + ["--gpus", "2", "--backend", "gloo", "--frames", "16", "--no-check"],
                                      env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=240) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                   # the children this test started, nothing else
    assert all(p.returncode == 4 for p in procs), [(p.returncode, o[1][-500:]) for p, o in zip(procs, outs)]
    assert "same device" in outs[0][1]


@pytest.mark.gpu
def test_self_launched_single_rank_matches_direct():
    args = ["--config", "B", "--frames", "512", "--steps", "30", "--warmup", "3", "--no-cpu-baseline", "--no-secondary", "--oracle-frames", "2"]
    rd, ld = _run(args)
    rs, ls = _run(args + ["--spawn"])
    assert rd.returncode == 0 and rs.returncode == 0, (rd.stderr[-1500:], rs.stderr[-1500:])
    jd, js = json.loads(ld[-1]), json.loads(ls[-1])
    assert jd["config"]["launcher"] == "direct" and js["config"]["launcher"] == "self-launched children"
    assert js["check"]["ok"] and jd["check"]["ok"]
    md, ms = jd["windows"]["ms_per_step_median"], js["windows"]["ms_per_step_median"]
    # same work per step whichever way the rank was started (measured: within 1-2 %).  A sanity bound, not a performance assertion: this suite is
    # the correctness evidence and runs on a shared box; the figures themselves are bench.py's business
    assert 0.5 < ms / md < 2.0, (md, ms)


@pytest.mark.gpu
def test_rccl_initialises_and_runs_the_collectives_the_bench_uses(tmp_path):
    """the box has one GPU, and RCCL refuses two ranks on one device, so the N > 1 tests above run over gloo; this one brings up the
    `nccl` (= RCCL) backend itself with a world of one — process group with device_id as bench.py creates it, the MAX / MIN all-reduce of
    the timing contract, all_gather_into_tensor of shard.gather_results, barrier — on device tensors"""
    script = tmp_path / "rccl_one.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import jrc_amd
from jrc_amd import shard
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
assert dist.get_backend() == "nccl"
t = torch.tensor([0.25, 3.0], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
assert t.tolist() == [0.25, 3.0]
assert shard.max_over_ranks_vec([1.0, 2.0], "cuda:0") == [1.0, 2.0] and shard.min_over_ranks(0.5, "cuda:0") == 0.5
x = torch.arange(48 * 5, dtype=torch.uint8, device="cuda:0").reshape(5, 48)
out = torch.empty_like(x)
dist.all_gather_into_tensor(out, x)
assert torch.equal(out, x) and torch.equal(shard.gather_results(x, 5), x)
objs = [None]
dist.all_gather_object(objs, {"rank": 0, "device_key": "x"})      # bench.gather_identities: pickled through RCCL's own device tensors
assert objs == [{"rank": 0, "device_key": "x"}]
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
''' % ROOT)
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240, env=e, cwd=ROOT)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-3000:]
