"""CPU tier: bench.py's main() end to end — process group aside, everything from the device buffers to the printed contract line — on the EMULATED kernels
(tests/hipcpu), as a check of the code path that prints the line the driver parses (round 5's line was not parsed; round 6 could not run bench.py on a
device at all).  The numbers of this run mean nothing and are not looked at; what is: the run completes, rank 0 prints exactly one JSON line under 6000
bytes carrying the contract's fields, `roofline`, `check.ok` (every frame verified against the oracle inside bench.py), and the full record beside it.
bench.py itself knows nothing of the emulation: a launcher script in this test points the package at the emulated library, stands in for the three
torch.cuda calls that ask about the device, and then runs bench.main() unchanged."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LAUNCHER = r'''
import os, sys, types
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from hipcpu import build as eb
os.environ["JRC_LIB_PATH"] = eb.build()
from hipcpu import torch_redirect
torch_redirect.install()
import torch
torch.cuda.is_available = lambda: True
torch.cuda.get_device_properties = lambda i: types.SimpleNamespace(name="hipcpu emulation (no GPU)", pci_domain_id=0, pci_bus_id=0, pci_device_id=0, uuid="emulated", multi_processor_count=256)
torch.cuda.is_initialized = lambda: False
import bench
sys.argv = ["bench.py"] + %(args)r
bench.main()
'''


def _have_clang():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hipcpu import build as eb
    try:
        eb.compiler()
        return os.uname().machine == "x86_64"
    except RuntimeError:
        return False


@pytest.mark.skipif(not _have_clang(), reason="the emulation needs clang++ and an x86-64 host")
@pytest.mark.parametrize("cfg,frames", [("A", 24), ("B", 3)])
def test_bench_main_prints_one_short_contract_line(tmp_path, cfg, frames):
    verbose = str(tmp_path / "verbose.json")
    args = ["--config", cfg, "--frames", str(frames), "--distinct", str(frames), "--steps", "2", "--warmup", "1", "--windows", "2", "--prewarm-seconds", "0",
            "--no-cpu-baseline", "--no-secondary", "--oracle-frames", "2", "--verbose-out", verbose]
    e = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "JRC_LIB_PATH"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", LAUNCHER % dict(root=ROOT, args=args)], capture_output=True, text=True, cwd=ROOT, env=e, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "check"):
        assert k in j, k
    assert j["metric"] == "ofdm_frames_per_sec" and j["unit"] == "frames/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert j["config"]["workload"].startswith("config %s:" % cfg) and j["config"]["frames_per_gpu_per_step"] == frames and j["config"]["launcher"] == "direct"
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["achieved"] > 0 and rf["algorithmic_bytes_per_launch"] > 0 and "chain" in rf
    assert j["check"]["ok"] is True and j["check"]["chanest_bit_exact"] is True and j["check"]["estimator_fields_exact"] is True and j["check"]["frames_checked"] == frames
    assert j["check"]["map_max_rel_err"] <= 1e-4
    assert j["value"] > 0 and abs(j["value"] - frames * 2 / (j["ms_per_step"] * 2e-3)) < 1e-3 * j["value"]
    full = json.load(open(verbose))
    assert full["check"]["ok"] and len(json.dumps(full)) > len(lines[0]) and full["device"].startswith("hipcpu emulation")
    assert "bench.py verbose record:" in r.stderr


def _spawn_ranks(world, args, timeout=600):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        e = dict(os.environ, OMP_NUM_THREADS="1", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), JRC_BENCH_CHILD="1")
        e.pop("JRC_LIB_PATH", None)
        procs.append(subprocess.Popen([sys.executable, "-c", LAUNCHER % dict(root=ROOT, args=args)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=e))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                   # the children this test started
    return procs, outs


@pytest.mark.skipif(not _have_clang(), reason="the emulation needs clang++ and an x86-64 host")
@pytest.mark.parametrize("world,stream", [(2, 96), (8, 203)])
def test_n_ranks_sharing_one_emulated_device_equal_one_rank(tmp_path, world, stream):
    """the N > 1 path of bench.py on the emulated kernels: `world` ranks (gloo, every rank on "GPU 0") split ONE stream of frames into the blocks of
    shard.frame_shard — ragged 26 / 25 for 203 frames over 8 ranks, the run that hung the device on the driver's box in round 5 — gather the records and
    two maps per rank inside every step, and the dump (records, channel estimates, maps, in frame order) must equal the single-rank run of the same
    stream bit for bit; one JSON line from rank 0, short, with one entry per rank"""
    import numpy as np
    sys.path.insert(0, ROOT)
    from jrc_amd import shard
    common = ["--config", "A", "--steps", "2", "--warmup", "1", "--windows", "2", "--prewarm-seconds", "0", "--no-cpu-baseline", "--no-secondary", "--oracle-frames", "2"]
    one, many = str(tmp_path / "one.npz"), str(tmp_path / "many.npz")
    p1, o1 = _spawn_ranks(1, common + ["--frames", str(stream), "--distinct", str(stream), "--dump", one, "--dump-maps", str(stream), "--verbose-out", str(tmp_path / "v1.json")])
    assert p1[0].returncode == 0, o1[0][1][-3000:]
    pn, on = _spawn_ranks(world, common + ["--gpus", str(world), "--same-device", "--backend", "gloo", "--stream-frames", str(stream), "--distinct", str(stream),
                                           "--dump", many, "--dump-maps", "2", "--gather-results", "--gather-maps", "2", "--verbose-out", str(tmp_path / "vn.json")])
    assert all(p.returncode == 0 for p in pn), [(p.returncode, o[1][-1500:]) for p, o in zip(pn, on) if p.returncode]
    lines = [l for o in on for l in o[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000                     # ONE line, from rank 0
    j = json.loads(lines[0])
    sizes = shard.shard_sizes(stream, world)
    assert sum(sizes) == stream and (world == 2 or sorted(set(sizes)) == [25, 26])
    assert j["n_gpus"] == world and j["scaling"] == "strong" and j["collective_world"] == world and j["backend"] == "gloo" and j["distinct_devices"] == 1
    assert j["check"]["ok"] and j["check"]["ok_all_ranks"] and [r["rank"] for r in j["ranks"]] == list(range(world)) and len(set(r["pid"] for r in j["ranks"])) == world
    assert j["config"]["stream_frames_per_step"] == stream and j["config"]["gather_results"] and j["config"]["gather_maps_per_gpu"] == 2
    a, b = np.load(one), np.load(many)
    assert a["results"].shape[0] == b["results"].shape[0] == stream
    assert np.array_equal(a["results"], b["results"]) and np.array_equal(a["chanest"], b["chanest"])
    firsts = [shard.frame_shard(stream, r, world)[0] for r in range(world)]
    assert np.array_equal(np.concatenate([a["maps"][f:f + 2] for f in firsts]), b["maps"])
