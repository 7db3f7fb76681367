#!/usr/bin/env python3
"""bench.py — OFDM frames/s of the radar hot path (A1 -> A5) on MI355X, with the roofline of the dominant
kernel and a CPU baseline beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config B|D|A] [--frames F]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the fused chain (mimo_ofdm_radar -> range IFFT -> transpose/pad -> angle FFT ->
range_angle_estimator) over one batch of F synthetic frames already resident in HBM.  Frames are
independent, so N GPUs each process their own batch (weak scaling, no data-path collective); the only
collectives are the barriers / MAX-reduce of the timing contract (and the optional --gather-* exchange).
Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself: this
process — which never touches the GPU — starts N fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set),
relays rank 0's line and exits non-zero if any child does.  Nothing is ever exec'ed from a process that holds a GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MAP_TOL = 1e-4                 # north_star: outputs within 1e-4 (relative, complex float) of the reference
N_WINDOWS = 5                  # timed windows of --steps steps each; the first one is the contract's timed region


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="B", choices=["A", "B", "D"])
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (0 = per-config default)")
    ap.add_argument("--interp-range", type=int, default=8, help="interp_factor of the range axis (the flowgraph's 8; other values are for kernel experiments)")
    ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic frames generated per GPU (tiled to --frames)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures of SURVEY §8(d) (config-D roofline, host-fed rate, chain incl. RX demod, detect-only "
                         "mode, equalizer / precoder config C) that rank 0 appends at N=1 after the timed region")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run verification of all F results (tiles bit-equal, distinct frames vs the oracle)")
    ap.add_argument("--oracle-frames", type=int, default=-1, help="distinct frames compared with the oracle after the timed region (-1 = per config)")
    ap.add_argument("--windows", type=int, default=N_WINDOWS, help="timed windows of --steps steps (>= 1; the first is the contract's timed region)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.3,
                    help="untimed runs before the W warm-up steps so a cold GPU has its clocks up (0 = none)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests of the N>1 path)")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--spawn", action="store_true", help="launch the ranks from this process even for --gpus 1 (the self-launch path)")
    ap.add_argument("--gather-results", action="store_true",
                    help="also RCCL all-gather the per-frame results every step (optional exchange, off by default)")
    ap.add_argument("--gather-maps", type=int, default=0, metavar="K",
                    help="also RCCL all-gather the range-angle maps of the first K frames of every rank each step (optional exchange, off by default)")
    ap.add_argument("--stream-frames", type=int, default=0, metavar="M",
                    help="testing: split ONE stream of M frames over the ranks (shard.frame_shard: contiguous blocks whose sizes may differ by one) "
                         "instead of --frames per rank; the line then says scaling: strong")
    ap.add_argument("--dump", default="", metavar="PATH.npz",
                    help="rank 0 writes the frame-ordered results of all ranks (and the first --dump-maps maps of every rank) after the timed region")
    ap.add_argument("--dump-maps", type=int, default=0)
    ap.add_argument("--verbose-out", default="", metavar="PATH.json",
                    help="where rank 0 writes the full record (every leg with its description); default gpurun_out/bench_verbose.json")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="self-launch path: seconds the ranks may take before the launcher kills the children it started and exits 124")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# self-launch: N ranks as fresh child processes of a parent that never initialises the GPU
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def same_device_hw_queues(world):
    """--same-device (testing only) puts `world` processes on ONE GPU.  HIP multiplexes a process's streams onto up to GPU_MAX_HW_QUEUES (default 4)
    hardware queues, and the device's compute scheduler holds 24 user queues per XCC before it has to time-slice the run list with wave
    save/restore — the regime in which the 8-rank run of round 5 ended in `HW Exception ... GPU Hang` on the driver's box (8 x 4 = 32 queues).
    A rank of this bench needs two (torch's stream and the context's), so the ranks of a shared device are capped to fit with room to spare (16 // world,
    at least 1, at most HIP's default 4).  One process per GPU — the real N-GPU launch — is not touched."""
    return max(1, min(4, 16 // max(1, world)))


def self_launch(a, argv):
    # the ranks start together: build (or find built) the extension ONCE here, before any of them looks for it.  Compiling is not a GPU
    # call; the build is also flock-guarded (build.build_lock) for ranks that a launcher other than this one starts on an unbuilt tree.
    from jrc_amd import build as jb
    jb.build()
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_WORLD_SIZE=str(a.gpus), JRC_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.same_device and a.gpus > 1:
            env.setdefault("GPU_MAX_HW_QUEUES", str(same_device_hw_queues(a.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    # bounded: a rank that never comes back (a device fault under it, a rendezvous that never completes) must not hold the launcher — and
    # whatever started the launcher — for ever.  On expiry the children THIS process started are killed (never anything else) and the exit is non-zero.
    deadline = time.time() + a.launch_timeout
    timed_out, out0 = False, ""
    try:
        out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        timed_out = True
    rc = procs[0].returncode or 0
    if not timed_out:
        for p in procs[1:]:
            try:
                p.wait(timeout=max(1.0, min(deadline, time.time() + 120) - time.time()))
            except subprocess.TimeoutExpired:
                timed_out = True
                break
            rc = rc or p.returncode
    if timed_out:
        for p in procs:
            if p.poll() is None:
                p.kill()                  # exactly the children this process started
        for p in procs:
            try:
                p.wait(timeout=15)        # a child stuck under a device fault may not even take SIGKILL at once: do not wait for it for ever
            except subprocess.TimeoutExpired:
                pass
        sys.stderr.write("bench.py: the ranks did not finish within --launch-timeout %.0f s: children killed\n" % a.launch_timeout)
        return 124
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    for l in (out0 or "").splitlines():
        if not l.startswith("{"):
            sys.stderr.write(l + "\n")
    for l in lines[:-1]:                  # a rank's earlier lines (the verbose record) pass through in front of the contract's line
        print(l)
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    if rc == 0 and not lines:
        rc = 1
    return rc


def rank_identity(torch, rank, local_rank, dev_index, my_windows, steps):
    """what makes the N > 1 line self-proving: which physical device this rank ran on and how long ITS OWN timed windows took"""
    p = torch.cuda.get_device_properties(dev_index)
    have_pci = all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    bus = "%04x:%02x:%02x" % tuple(int(getattr(p, k)) & 0xffff for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")) if have_pci else "unknown"
    uuid = str(getattr(p, "uuid", ""))
    host = socket.gethostname()
    # one physical device = (host, PCI address), or (host, uuid) where torch does not expose the address, or (host, device index) as a last resort:
    # two hosts with the same PCI topology are different devices, and missing properties must not make every rank look like the same one
    device_key = "%s/%s" % (host, bus if have_pci else (uuid if uuid else "index%d" % dev_index))
    return {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device_name": p.name, "pci_bus_id": bus,
            "uuid": uuid, "device_key": device_key, "compute_units": int(getattr(p, "multi_processor_count", 0)),
            "host": host, "pid": os.getpid(),
            "ms_per_step_window0": 1e3 * my_windows[0] / steps, "ms_per_step_windows": [1e3 * w / steps for w in my_windows]}


def gather_identities(dist, world, mine):
    if world == 1:
        return [mine]
    allr = [None] * world
    dist.all_gather_object(allr, mine)          # pickled through the backend's own tensors (RCCL: on this rank's device)
    return sorted(allr, key=lambda r: r["rank"])


def scenario(name):
    from jrc_amd import synth
    return {"A": synth.config_A, "B": synth.config_B, "D": synth.config_D}[name]()


# ----------------------------------------------------------------------------------------------------------------------
# CPU leg (oracle): baseline timing and the expected outputs of the distinct frames
# ----------------------------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    """one host core: the oracle's block-by-block chain on its share of the sample"""
    import oracle
    sc_args, Ir, Ia, frames, axes, budget_s = args
    from jrc_amd import synth
    sc = synth.Scenario(*sc_args)
    rb, ab, ndr, nda = axes
    rad = oracle.Radar(sc.N, sc.T, sc.R, sc.S, sc.Npre, interp_factor=Ir)
    done, t0 = 0, time.perf_counter()
    while True:
        fr = frames[done % len(frames)]
        m = rad.chain([fr[t] for t in range(sc.T)], [fr[sc.T + r] for r in range(sc.R)], Ia)
        oracle.ra_estimate(m, rb, ab, ndr, nda, 15.0, 0.0)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 4096:
            return done, el


def _cpu_stage_times(sc_args, Ir, Ia, frames, axes):
    """seconds per frame of each block of the radar branch (oracle port, float32 FFTs), one thread"""
    import oracle
    from jrc_amd import synth
    sc = synth.Scenario(*sc_args)
    rb, ab, ndr, nda = axes
    P = sc.T * sc.R
    rad = oracle.Radar(sc.N, sc.T, sc.R, sc.S, sc.Npre, interp_factor=Ir)
    acc = {"mimo_ofdm_radar": 0.0, "fft_range": 0.0, "matrix_transpose": 0.0, "fft_angle": 0.0, "range_angle_estimator": 0.0}
    for fr in frames:
        t = [time.perf_counter()]
        H = rad.work([fr[k] for k in range(sc.T)], [fr[sc.T + r] for r in range(sc.R)]); t.append(time.perf_counter())
        rng = oracle.fft_vcc(H, False, False, f32=True); t.append(time.perf_counter())
        tr = oracle.matrix_transpose(rng, sc.N * Ir, P, Ia); t.append(time.perf_counter())
        m = oracle.fft_vcc(tr, True, True, f32=True); t.append(time.perf_counter())
        oracle.ra_estimate(m, rb, ab, ndr, nda, 15.0, 0.0); t.append(time.perf_counter())
        for k, name in enumerate(acc):
            acc[name] += t[k + 1] - t[k]
    out = {k: v / len(frames) for k, v in acc.items()}
    # The reference's two FFT stages are FFTW3f behind gr::fft::fft_vcc, not a scalar radix-2 loop.  FFTW is not in this image; scipy's
    # pocketfft (C++, SIMD, native float32) is the closest tuned library here: the same two padded transforms through it, for scale.
    try:
        import scipy.fft as sfft
        Hs = [rad.work([fr[k] for k in range(sc.T)], [fr[sc.T + r] for r in range(sc.R)]) for fr in frames]
        sfft.fft(oracle.matrix_transpose(sfft.ifft(Hs[0], axis=-1, norm="forward", workers=1).astype(np.complex64, copy=False), sc.N * Ir, P, Ia),
                 axis=-1, workers=1)                                   # plans and twiddles: not timed
        t0 = time.perf_counter()
        rngs = [sfft.ifft(H, axis=-1, norm="forward", workers=1).astype(np.complex64, copy=False) for H in Hs]
        t1 = time.perf_counter()
        trs = [oracle.matrix_transpose(r_, sc.N * Ir, P, Ia) for r_ in rngs]
        t2 = time.perf_counter()
        for tr in trs:
            sfft.fftshift(sfft.fft(tr, axis=-1, workers=1), axes=-1)
        t3 = time.perf_counter()
        out["fft_range_pocketfft"] = (t1 - t0) / len(frames)
        out["fft_angle_pocketfft"] = (t3 - t2) / len(frames)
    except Exception:
        pass
    return out


def cpu_baseline(sc, Ir, Ia, frames, axes, budget_s):
    """the oracle's block-by-block chain (A1..A5, float32) timed on a bounded sample of the same frames:
    `value` = one thread (the reference's per-block regime); `all_cores` = frame-parallel over every host core."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    sc_args = (sc.N, sc.T, sc.R, sc.S, sc.Npre)
    done, el = _cpu_worker((sc_args, Ir, Ia, frames, axes, budget_s * 0.6))
    out = {"value": done / el, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same workload, oracle C port (float32 radix-2 FFTs, not FFTW: the reference's stock fft_vxx blocks are "
                     "faster than this port), 1 thread, %.1f s" % (done, el)}
    try:
        # (b) pipeline-ideal: GNU Radio runs one thread per block, so a saturated flowgraph moves at the pace of its slowest block
        st = _cpu_stage_times(sc_args, Ir, Ia, frames[:3], axes)
        five = {k: v for k, v in st.items() if not k.endswith("_pocketfft")}
        out["pipeline_ideal"] = {"value": 1.0 / max(five.values()), "unit": "frames/s", "cores": len(five),
                                 "stage_ms": {k: 1e3 * v for k, v in st.items()},
                                 "sample": "1 / slowest block of the five (thread-per-block scheduling), %d frames block by block" % len(frames[:3])}
        if "fft_angle_pocketfft" in st:      # the same pipeline with a tuned FFT library in the two FFT blocks (the reference uses FFTW3f there)
            tuned = dict(five, fft_range=st["fft_range_pocketfft"], fft_angle=st["fft_angle_pocketfft"])
            out["pipeline_ideal"]["value_with_tuned_fft"] = 1.0 / max(tuned.values())
            out["pipeline_ideal"]["single_thread_with_tuned_fft"] = 1.0 / sum(tuned.values())
            out["pipeline_ideal"]["tuned_fft"] = "scipy.fft (pocketfft, float32, 1 worker) in place of the port's radix-2 loops for the two FFT stages"
    except Exception as e:
        out["pipeline_ideal"] = {"error": str(e)}
    try:
        # every host core this process may run on; the only cap is memory: a worker holds the oracle's double-precision
        # work buffers for one frame (a few map sizes), so leave each 64 map-sizes of the MemAvailable the box reports
        ncpu = len(os.sched_getaffinity(0))
        per_worker = 64 * (sc.N * Ir) * (sc.T * sc.R * Ia) * 16
        try:
            avail = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
        except (OSError, IndexError, ValueError):
            avail = 8 << 30
        n = max(1, min(ncpu, int(avail * 0.5 // per_worker)))
        with mp.get_context("fork").Pool(n) as pool:
            res = pool.map(_cpu_worker, [(sc_args, Ir, Ia, frames[:4], axes, budget_s * 0.3)] * n)
        out["all_cores"] = {"value": sum(d / e for d, e in res), "unit": "frames/s", "cores": n,
                            "sample": "%d frames over %d processes" % (sum(d for d, _ in res), n)}
    except Exception as e:      # the single-thread figure above is the contract; this one is extra
        out["all_cores"] = {"error": str(e)}
    return out


def oracle_expect(sc, Ir, Ia, frames):
    """the checker's outputs for the given distinct frames: channel estimate H (A1) and range-angle map (A2..A4)"""
    import oracle
    oracle.build()
    exp = []
    for fr in frames:
        rad = oracle.Radar(sc.N, sc.T, sc.R, sc.S, sc.Npre, interp_factor=Ir)
        tx = [fr[t] for t in range(sc.T)]
        rx = [fr[sc.T + r] for r in range(sc.R)]
        H = rad.work(tx, rx)
        m = oracle.fft_vcc(oracle.matrix_transpose(oracle.fft_vcc(H, False, False), sc.N * Ir, sc.T * sc.R, Ia), True, True)
        exp.append((H[:, :sc.N].copy(), m.astype(np.complex64)))
    return exp


def verify(torch, bufs, res_bytes, res, F, n_distinct, expect, axes, sc):
    """after the timed region: every one of the F frames of the batch is checked.
    (1) the batch is n_distinct frames tiled: frame f must equal frame f % n_distinct BIT FOR BIT in all three outputs
        (channel estimate, map, result record) — so the frames compared with the oracle stand for every position of the launch
        geometry that was timed (first/last workgroup of an XCD group, first/last frame of the batch);
    (2) the distinct frames in `expect` against the oracle: A1 bit-exact, map within MAP_TOL, A5 record exact on the same map."""
    import oracle
    rb, ab, ndr, nda = axes
    out = {"frames_checked": int(F), "tiles": int(-(-F // n_distinct))}
    ok = True
    tiled = True
    for name in ("chanest", "map"):
        t = bufs[name]
        base = t[:n_distinct]
        for f0 in range(n_distinct, F, n_distinct):
            n = min(n_distinct, F - f0)
            if not torch.equal(t[f0:f0 + n], base[:n]):
                tiled = False
    rb_all = res_bytes.cpu().numpy()
    for f0 in range(n_distinct, F, n_distinct):
        n = min(n_distinct, F - f0)
        if not np.array_equal(rb_all[f0:f0 + n], rb_all[:n]):
            tiled = False
    out["tiled_bit_equal"] = tiled
    ok = ok and tiled
    if expect:
        import ctypes
        k = len(expect)
        h_exact, est_exact, worst = True, True, 0.0
        tiles = -(-F // n_distinct)
        for i, (H, m) in enumerate(expect):
            # take distinct frame i from a tile that moves through the batch (first tile, last full tile, ...)
            pos = i + n_distinct * ((i * 7) % tiles)
            if pos >= F:
                pos = i
            gH = bufs["chanest"][pos].cpu().numpy().view(np.complex64)[..., 0]
            gm = bufs["map"][pos].cpu().numpy().view(np.complex64)[..., 0]
            if not np.array_equal(gH, H):
                h_exact = False
            d = float(np.abs(m).max())
            worst = max(worst, float(np.abs(gm - m).max() / (d if d > 0 else 1.0)))
            ro = oracle.ra_estimate(gm, rb, ab, ndr, nda, 15.0, 0.0)          # the oracle's estimator on the SAME map
            # jrc_ra_result and the oracle's record have the same twelve fields; `res` is the host-finished record (snr_est, published)
            want = ctypes.string_at(ctypes.byref(ro), ctypes.sizeof(ro))
            got = ctypes.string_at(ctypes.byref(res[pos]), ctypes.sizeof(res[pos]))
            if got != want:
                est_exact = False
        out.update(oracle_frames=k, chanest_bit_exact=h_exact, map_max_rel_err=worst, map_tol=MAP_TOL, estimator_fields_exact=est_exact)
        ok = ok and h_exact and est_exact and worst <= MAP_TOL
    else:
        out["oracle_frames"] = 0
    out["ok"] = bool(ok)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# secondary figures (rank 0, N = 1, after the timed region; never part of `value`)
# ----------------------------------------------------------------------------------------------------------------------
def pmc_traffic(cfg, frames_per_launch):
    """PMC-measured HBM bytes per launch of the dominant kernel, collected by tools/profile_round.sh in separate rocprofv3 --pmc passes and
    stamped with the hash of the kernel sources it was measured at: returns (bytes or None, stale flag)"""
    from jrc_amd import build as jb
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        pt = d.get(cfg)
        if not pt or pt.get("frames_per_launch") != frames_per_launch:
            return None, False
        if d.get("source_hash") != jb.source_hash():
            # measured at other kernel SOURCES.  It still describes this tree's kernel if the DEVICE CODE of the files the chain's kernels live in is
            # byte-identical between the tree it was measured at and this one: profiles/rNN_device_code_vs_*.json (tools/device_code_diff.py — both trees
            # compiled with the library's flags and --offload-device-only, outputs compared byte for byte), itself stamped with both source hashes
            import glob
            for rec in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_device_code_vs_*.json")), reverse=True):
                dc = json.load(open(rec))
                if (dc.get("source_hash_a") == d.get("source_hash") and dc.get("source_hash_b") == jb.source_hash()
                        and all(f in dc.get("identical_device_code", []) for f in ("chain.hip", "radar.hip", "estimator.hip"))):
                    return pt["hbm_bytes_per_launch"], False
            return None, True
        return pt["hbm_bytes_per_launch"], False
    except (OSError, ValueError, KeyError):
        return None, False


def roofline_of(chain, kt, cfg, sc, Ir, Ia, F):
    P, NR, NA = sc.T * sc.R, sc.N * Ir, sc.T * sc.R * Ia
    # a batch beyond one resident wave of workgroups runs as several launches of the dominant kernel: roofline per launch
    n_launch = max(1, chain.launches_per_run(F))
    f_launch = F // n_launch if F % n_launch == 0 else None
    k_ms = kt["range_angle_fused"] / n_launch
    # algorithmic bytes of the dominant kernel: unpadded H in, complex map out
    alg_bytes = F * (P * sc.N * 8 + NR * NA * 8) / n_launch
    traffic, stale = pmc_traffic(cfg, f_launch if f_launch is not None else F)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    wide = P in (8, 16) and Ia == 16 and not os.environ.get("JRC_NO_WIDE") and sc.N in (256, 512, 1024) and NR >= 256       # chain.hip: jrc_chain::wide
    r = {"bound": "hbm", "kernel": ("range_angle_wide_kernel<%d>" if wide else "range_angle_fused_kernel<%d>") % P, "achieved": achieved,
         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
         "avg_launch_ms": k_ms, "launches_timed": kt["launches"] * n_launch,
         "launches_per_step": n_launch, "frames_per_launch": F / n_launch}
    if stale:
        r["traffic_stale"] = True      # profiles/pmc_traffic.json was measured at other kernel sources: re-run tools/profile_round.sh
    return r


def chain_roofline(sc, Ir, Ia, F, ms_step, kt):
    """the whole step against the HBM roofline (SURVEY.md §8(d) 'full radar chain A1-A5 with map written': inputs read once, H written and
    read once is NOT compulsory, map written once), next to the dominant kernel's own figure above; and A1 alone on its read stream"""
    P, NR, NA = sc.T * sc.R, sc.N * Ir, sc.T * sc.R * Ia
    per_frame = (sc.T + sc.R) * sc.S * sc.N * 8 + P * sc.N * 8 + NR * NA * 8            # B: 5,275,648  D: 25,296,896
    a1_bytes = (sc.T + sc.R) * sc.S * sc.N * 8 + P * sc.N * 8                            # inputs read, H written
    out = {"compulsory_bytes_per_frame": per_frame, "compulsory_bytes_per_step": per_frame * F,
           "achieved": per_frame * F / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": per_frame * F / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_step": ms_step,
           "kernels_sum_ms": kt["radar_chanest"] + kt["range_angle_fused"] + kt["ra_finalize"]}
    if kt["radar_chanest"] > 0:
        g = a1_bytes * F / (kt["radar_chanest"] * 1e-3) / 1e9
        out["a1"] = {"kernel": "radar_chanest_x2_kernel", "bound": "hbm (read)", "algorithmic_bytes_per_step": a1_bytes * F, "avg_ms": kt["radar_chanest"],
                     "achieved": g, "frac": g / HBM_PEAK_GBS}
    return out


def config_d_roofline(ctx, steps=30, warm=8):
    """Metric 2's home configuration (4x4, 1024 subcarriers, 128 symbols, 8 targets): the same chain on a resident batch of 256 frames"""
    import torch
    import jrc_amd
    from jrc_amd import synth
    sc = synth.config_D()
    Ir, Ia, F = 8, 16, 256
    P = sc.T * sc.R
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    chain = jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 14.36, 15.0, 0.0, max_frames=F, ctx=ctx)
    bufs = chain.alloc(F, "cuda:%d" % ctx.device)
    fr = synth.make_frames(sc, 8)
    hf = torch.from_numpy(fr.view(np.float32).reshape((8,) + tuple(bufs["frames"].shape[1:])))
    for f0 in range(0, F, 8):
        bufs["frames"][f0:f0 + 8].copy_(hf)
    torch.cuda.synchronize()
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < 0.3:       # the same untimed pre-warm as the headline run
        chain.run(bufs, F)
        ctx.sync()
    for _ in range(warm):
        chain.run(bufs, F)
    ctx.sync()
    chain.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        chain.run(bufs, F)
    ctx.sync()
    el = time.perf_counter() - t0
    kt = chain.get_timing()
    r = roofline_of(chain, kt, "D", sc, Ir, Ia, F)
    r.update(frames_per_s=F * steps / el, ms_per_step=1e3 * el / steps, frames_per_step=F,
             what="config D (4x4, 1024 sc, 128 sym, 8 targets): chain A1..A5 on %d resident frames, %d steps" % (F, steps))
    chain.close()
    del bufs
    torch.cuda.empty_cache()
    return r


def host_fed_rate(ctx, cfg, sc, axes, seconds=1.5):
    """PCIe-inclusive rate (SURVEY §8(d) 'with and without H2D/D2H'): frames start in pinned HOST memory, results (48 B/frame) come back
    to the host; three batches in flight, hipGraph replay"""
    import jrc_amd
    from jrc_amd import synth
    rb, ab, ndr, nda = axes
    fps = 64 if cfg != "D" else 16
    slots = 3
    src = synth.make_frames(sc, 8)
    src = np.concatenate([src] * (fps // 8))[:fps].copy()
    feed = jrc_amd.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, 8, 16, rb, ab, ndr, nda, 15.0, 0.0, ctx=ctx, n_slots=slots, frames_per_slot=fps, graph=True)
    for _ in range(slots):
        feed.acquire()[:] = src
        feed.submit(None, fps)
    for _ in range(slots):
        feed.collect()
    done = 0
    t0 = time.perf_counter()
    while True:
        while feed.pending() < slots:
            feed.acquire()
            feed.submit(None, fps)
        r, _ = feed.collect()
        done += len(r)
        if time.perf_counter() - t0 > seconds:
            break
    while feed.pending():
        done += len(feed.collect()[0])
    el = time.perf_counter() - t0
    g = done * feed.frame_bytes / el / 1e9
    out = {"frames_per_s": done / el, "host_GBps": g, "frames_per_batch": fps, "batches_in_flight": slots,
           # the three fields every secondary leg carries; this leg's bound is the PCIe link (~56 GB/s in practice), not HBM
           "algorithmic_bytes_per_step": int(fps * feed.frame_bytes), "GBps_algorithmic": g, "frac_of_hbm_peak": g / HBM_PEAK_GBS, "bound": "pcie",
           "what": "config %s, frames in pinned host memory -> H2D -> chain -> 48-byte results D2H, hipGraph replay (jrc_chain_feed_*)" % cfg}
    feed.close()
    return out


def host_fed_tx_resident_rate(ctx, cfg, sc, axes, seconds=1.5):
    """the same pipeline when the T reference ports repeat from frame to frame (the reference flowgraph's radar block is pointed at the MIMO-LTF
    rows, …radar_sim.grc:1292-1295): they are handed over once (jrc_chain_feed_set_tx), every batch uploads its receive ports only
    (jrc_chain_feed_submit_rx) and leaves out the N_pre preamble symbols mimo_ofdm_radar never reads.  Results checked against a full upload."""
    import jrc_amd
    from jrc_amd import synth
    rb, ab, ndr, nda = axes
    fps = 64 if cfg != "D" else 16
    slots = 3
    src = synth.make_frames(sc, 8)
    src = np.concatenate([src] * (fps // 8))[:fps].copy()
    src[:, :sc.T] = src[0, :sc.T]
    used = np.ascontiguousarray(src[:, :, sc.Npre:])                          # the N_sym symbols the radar block reads
    feed = jrc_amd.ChainFeed(sc.N, sc.T, sc.R, sc.S, 0, 8, 16, rb, ab, ndr, nda, 15.0, 0.0, ctx=ctx, n_slots=slots, frames_per_slot=fps)
    feed.submit(used)
    want, _ = feed.collect()
    feed.set_tx(used[0, :sc.T])
    for _ in range(slots):
        feed.acquire()[:, sc.T:] = used[:, sc.T:]
        feed.submit(None, fps, rx_only=True)
    got = []
    for _ in range(slots):
        got = feed.collect()[0]
    same = all((a.peak_range_idx, a.peak_angle_idx, a.snr_est, a.noise_power) == (b.peak_range_idx, b.peak_angle_idx, b.snr_est, b.noise_power)
               for a, b in zip(got, want))
    done = 0
    t0 = time.perf_counter()
    while True:
        while feed.pending() < slots:
            feed.acquire()
            feed.submit(None, fps, rx_only=True)
        done += len(feed.collect()[0])
        if time.perf_counter() - t0 > seconds:
            break
    while feed.pending():
        done += len(feed.collect()[0])
    el = time.perf_counter() - t0
    rx_bytes = sc.R * sc.S * sc.N * 8
    g = done * rx_bytes / el / 1e9
    out = {"frames_per_s": done / el, "host_GBps": g, "bytes_per_frame_over_pcie": rx_bytes,
           "algorithmic_bytes_per_step": int(fps * rx_bytes), "GBps_algorithmic": g, "frac_of_hbm_peak": g / HBM_PEAK_GBS, "bound": "pcie",
           "bytes_per_frame_full_upload": (sc.T + sc.R) * (sc.Npre + sc.S) * sc.N * 8, "results_equal_full_upload": bool(same),
           "frames_per_batch": fps, "batches_in_flight": slots,
           "what": "config %s with the TX reference ports resident on the device (jrc_chain_feed_set_tx / _submit_rx): receive ports of the "
                   "N_sym used symbols in pinned host memory -> H2D -> chain -> results D2H" % cfg}
    feed.close()
    return out


def flowgraph_shape_host_fed(ctx, seconds=1.0):
    """the only shape the reference's flowgraphs run (…radar_sim.grc: 4 TX x 2 RX, fft_len 64, N_pre 5, N_sym 4, interp 8 / 16): 27 KB per
    packet, a latency regime.  Through jrc_chain_feed_* — one packet per batch for latency (p50 / p99 of submit -> collect with an idle
    pipeline), 16 per batch and three batches in flight for throughput — through the radar_chain block (host/jrc_blocks.cc, one scheduler turn
    of 64 packets at a time), and the CPU port (the oracle's A1..A5 on one core) on the same packets."""
    import ctypes as C
    import jrc_amd
    from jrc_amd import synth
    sc = synth.Scenario(64, 4, 2, 4, targets=[(10.0, 20.0, 0.0, 100.0)])
    Ir, Ia, P = 8, 16, 8
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ndr, nda = 2 * 3e8 / (2 * sc.fs), 2 * float(np.rad2deg(np.arcsin(2 / P)))
    frames = synth.make_frames(sc, 64)
    frames[:, :sc.T] = frames[0, :sc.T]                                       # the MIMO-LTF rows: the same in every packet
    n_items = sc.Npre + sc.S
    out = {"shape": "4 TX x 2 RX, fft_len 64, N_pre 5, N_sym 4, interp 8 x 16 (512 x 128 map), %d B per packet on the block's ports" % (6 * n_items * 64 * 8)}
    # (1) latency, idle pipeline, one packet per batch, map not stored (what the radar_chain block asks for)
    feed = jrc_amd.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, ndr, nda, 15.0, 0.0, ctx=ctx, n_slots=2, frames_per_slot=1, graph=True)
    feed.set_write_map(False)
    lat = []
    for i in range(600):
        st = feed.acquire()
        st[0] = frames[i % 64]
        t0 = time.perf_counter()
        feed.submit(None, 1)
        feed.collect()
        lat.append(time.perf_counter() - t0)
    lat = np.sort(np.array(lat[100:])) * 1e6
    out["feed_latency_us_one_packet"] = {"p50": float(lat[len(lat) // 2]), "p99": float(lat[int(len(lat) * 0.99)]), "min": float(lat[0]),
                                         "what": "submit -> collect of a single packet, pinned staging, hipGraph replay, idle pipeline"}
    feed.close()
    # (2) throughput, 16 packets per batch, three in flight
    feed = jrc_amd.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, ndr, nda, 15.0, 0.0, ctx=ctx, n_slots=3, frames_per_slot=16)
    feed.set_write_map(False)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        while feed.pending() < 3:
            feed.acquire()[:] = frames[:16]
            feed.submit(None, 16)
        done += len(feed.collect()[0])
    while feed.pending():
        done += len(feed.collect()[0])
    out["feed_frames_per_s"] = done / (time.perf_counter() - t0)
    feed.close()
    # (3) the radar_chain block: 64 packets per scheduler turn on its T+R ports, tags on ports 0 and T
    lib = os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd", "lib", "libjrc_blocks.so")
    if os.path.exists(lib):
        L = C.CDLL(lib)
        fp, vp = C.POINTER(C.c_float), C.c_void_p
        L.jrcb_make_radar_chain.restype = vp
        L.jrcb_make_radar_chain.argtypes = [C.c_int] * 8 + [fp, C.c_int, fp, C.c_int] + [C.c_float] * 4 + [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.jrcb_add_in_tag.argtypes = [vp, C.c_int, C.c_uint64, C.c_char_p, C.c_int, C.c_long, C.c_double]
        L.jrcb_run.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.jrcb_call_setter.argtypes = [vp, C.c_char_p, C.c_double]
        L.jrcb_destroy.argtypes = [vp]
        L.jrcb_profile_ns.restype = C.c_longlong
        L.jrcb_profile_ns.argtypes = [vp, C.c_int]
        rbf, abf = np.ascontiguousarray(rb, np.float32), np.ascontiguousarray(ab, np.float32)
        L.jrcb_add_in_tags.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.c_int, C.c_long, C.c_double]
        ports = [np.ascontiguousarray(np.concatenate([frames[f, p] for f in range(64)])) for p in range(sc.T + sc.R)]
        nin = (C.c_int * 6)(*[64 * n_items] * 6)
        pin = (vp * 6)(*[a.ctypes.data for a in ports])
        pout = (vp * 1)()

        def run_block(fpb, slots, pre):
            h = L.jrcb_make_radar_chain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, 0, rbf.ctypes.data_as(fp), len(rbf), abf.ctypes.data_as(fp), len(abf),
                                        ndr, nda, 15.0, 0.0, b"", 0, fpb, slots)
            if not h:
                return
            # the length tags of a turn arrive with its items, attached by the upstream blocks on THEIR threads: added here in two calls per
            # turn (jrcb_add_in_tags), outside the block's work() — one ctypes call per tag costs the harness more than the block spends per packet
            turns, pos, t0, t_turns, t_work, prof0 = 0, 0, None, 0, 0.0, None
            while True:
                L.jrcb_add_in_tags(h, 0, pos, n_items, 64, b"packet_len", 0, n_items, 0.0)
                L.jrcb_add_in_tags(h, sc.T, pos, n_items, 64, b"packet_len", 0, n_items, 0.0)
                tw = time.perf_counter()
                rc_turn = L.jrcb_run(h, 0, nin, 6, pin, 0, pout)
                if t0 is not None:
                    t_work += time.perf_counter() - tw
                if rc_turn < 0:
                    out[pre + "error"] = "jrcb_run returned %d after %d turns" % (rc_turn, turns)
                    break
                pos += 64 * n_items
                turns += 1
                if turns == 20:
                    t0, t_turns = time.perf_counter(), turns
                    prof0 = [L.jrcb_profile_ns(h, k) / 1e3 for k in range(4)]
                if t0 is not None and time.perf_counter() - t0 > seconds:
                    break
            if t0 is not None and turns > t_turns:
                el = time.perf_counter() - t0                          # (before the final flush: the rate of the steady state)
                npk = (turns - t_turns) * 64
                prof = [L.jrcb_profile_ns(h, k) / 1e3 - prof0[k] for k in range(4)]
                out[pre + "frames_per_s"] = npk / el
                out[pre + "us_per_packet_inside_work"] = 1e6 * t_work / npk
                out[pre + "us_per_packet_breakdown"] = {"staging_tx_compare_and_copies": prof[0] / npk, "feed_submit_calls": prof[1] / npk,
                                                        "collect_and_publish": prof[2] / npk, "general_work_total": prof[3] / npk}
            L.jrcb_call_setter(h, b"flush", 0.0)
            out[pre + "rx_only_batches"] = int(L.jrcb_call_setter(h, b"rx_only_batches", 0.0))
            out[pre + "frames_done"] = int(L.jrcb_call_setter(h, b"frames_done", 0.0))
            out[pre + "every_packet_accounted_for"] = out[pre + "frames_done"] == turns * 64
            L.jrcb_destroy(h)

        run_block(0, 0, "radar_chain_block_")                        # the block's defaults (make(): frames_per_batch, batches_in_flight)
        for combo in os.environ.get("JRC_BENCH_BLOCK_COMBOS", "16x3,32x3").split(","):   # other (frames per batch) x (batches in flight)
            fpb, slots = (int(v) for v in combo.split("x"))
            run_block(fpb, slots, "radar_chain_block_%d_per_batch_%d_in_flight_" % (fpb, slots))
    # (4) the CPU port on the same packets, one core
    import oracle
    n_cpu, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min(seconds, 1.0):
        f = frames[n_cpu % 64]
        rad = oracle.Radar(sc.N, sc.T, sc.R, sc.S, sc.Npre, interp_factor=Ir)
        m = rad.chain([f[t] for t in range(sc.T)], [f[sc.T + r] for r in range(sc.R)], Ia)
        oracle.ra_estimate(m, rb, ab, ndr, nda, 15.0, 0.0)
        n_cpu += 1
    out["cpu_port_frames_per_s_one_core"] = n_cpu / (time.perf_counter() - t0)
    return out


def per_block_cpu_blocks(point, calls=30):
    """cpu_baseline-kind leg: the oracle's restatement of each hot block (the reference's CPU block) timed one call at a time on the shapes
    tools/per_block_probe.py times the drop-in blocks on — p50 in microseconds, one core"""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import per_block_probe as pbp
    rng = np.random.default_rng(7)
    N, cp, T, R, Npre, S, n_data, Ir, Ia = pbp.shapes(point)
    o = pbp.tables(point)
    P, NR, NA = T * R, N * Ir, T * R * Ia
    n_sync = len(o["l_stf_ltf_64"])
    n_total = n_sync + 1 + T + n_data
    nd = len(o["data_subcarriers"])
    import jrc_amd
    rb, ab = jrc_amd.radar_axes(N, 125e6, Ir, P, Ia)

    def p50(fn):
        fn()
        t = []
        for _ in range(calls):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return 1e6 * sorted(t)[len(t) // 2]
    cr = pbp.crandn
    out = {}
    rad = oracle.Radar(N, T, R, S, Npre, interp_factor=Ir)
    tx, rx = [cr(rng, n_total, N) for _ in range(T)], [cr(rng, n_total, N) for _ in range(R)]
    out["mimo_ofdm_radar"] = p50(lambda: rad.work(tx, rx))
    x = cr(rng, P, NR)
    out["matrix_transpose"] = p50(lambda: oracle.matrix_transpose(x, NR, P, Ia))
    m = cr(rng, NR, NA, scale=0.05)
    m[NR // 8, NA // 2 + 9] += 3.0
    nda = 28.96 if P == 8 else 14.36
    out["range_angle_estimator"] = p50(lambda: oracle.ra_estimate(m, rb, ab, 2.4, nda, 15.0, 0.0))
    sg = cr(rng, n_total * (N + cp))
    out["ofdm_cyclic_prefix_remover"] = p50(lambda: oracle.cp_remove(sg, N, cp))
    spec = cr(rng, 40000, scale=0.001)
    spec[3100] = 2 * np.exp(-0.7j)
    out["fft_peak_detect"] = p50(lambda: oracle.fft_peak_detect(spec, 125000000, 8.0, -20.0, 10))
    pre = oracle.Precoder(N, T, 1, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"], o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"])
    pdu_len = (n_data * nd - 22) // 8
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    sym = pts[rng.integers(0, 4, n_data * nd)].astype(np.complex64)
    out["mimo_precoder"] = p50(lambda: pre.work(sym, 2, 2, pdu_len))
    txf = pre.work(sym, 2, 2, pdu_len)
    y = np.tensordot(cr(rng, T), txf, axes=(0, 0))
    y = np.ascontiguousarray(np.concatenate([y[n_sync - 1:n_sync], y[n_sync - 1:]]), np.complex64)
    eq = oracle.Equalizer(0, 24e9, 125e6, N, cp, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"], o["ltf_64"], o["ltf_mapped_sc__ss_sym"], T)
    out["mimo_ofdm_equalizer"] = p50(lambda: eq.general_work(y, [(0, 0.0)]))
    # the two stock fft_vxx blocks of the radar branch on the CPU, as the per-block probe runs them (scipy pocketfft)
    import scipy.fft as sfft
    out["stock_fft_vxx_range_cpu"] = p50(lambda: (sfft.ifft(x, axis=1) * np.float32(NR)).astype(np.complex64))
    out["stock_fft_vxx_angle_cpu"] = p50(lambda: np.ascontiguousarray(sfft.fftshift(sfft.fft(m, axis=1), axes=1), np.complex64))
    return out


def per_block_drop_in():
    """VERDICT r4 missing 4: the literal per-block drop-ins (tools/per_block_probe.py, in its own process: it opens its own context) beside
    the oracle's CPU blocks on the same shapes (this process, one core)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "per_block_probe.py"), "--json", "--calls", "200", "--seconds", "0.7"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "per_block_probe rc %d: %s" % (r.returncode, r.stderr[-400:])}
    out = json.loads(lines[-1])
    for point in out:
        cpu = per_block_cpu_blocks(point)
        for name, b in out[point]["blocks"].items():
            b["cpu_block_p50_us"] = cpu.get(name)
            b["gpu_us_over_cpu_us"] = (b["p50_us"] / cpu[name]) if cpu.get(name) else None     # > 1: the GPU call takes LONGER than the CPU block
        out[point]["stock_fft_vxx_cpu_p50_us"] = {"range": cpu["stock_fft_vxx_range_cpu"], "angle": cpu["stock_fft_vxx_angle_cpu"]}
    out["what"] = ("one work() call of each hot block through host/jrc_blocks.cc with host buffers (H2D + kernels + D2H + sync per call) — p50 / p99 us and the "
                   "bytes over PCIe — beside the oracle's CPU block on the same shapes (cpu_block_p50_us, one core); radar_branch_unchanged_grc = GPU radar -> CPU "
                   "fft -> GPU transpose -> CPU fft -> GPU estimator as wired in the reference's .grc, packets/s")
    return out


def secondary_figures(cfg, ctx, sc, axes):
    """Metric 2 on its home config D, the PCIe-inclusive rate, the same chain with the RX OFDM demod in front (time-domain RX in),
    and the comm-side config C — measured after the timed region (their own batches, a few seconds in total); never part of `value`"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    out = {}

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as ex:            # a secondary figure must never take the headline line down with it
            out[name] = {"error": repr(ex)}

    if cfg != "D":
        leg("roofline_config_d", lambda: config_d_roofline(ctx))
    leg("host_fed_chain", lambda: host_fed_rate(ctx, cfg, sc, axes))
    leg("host_fed_chain_tx_resident", lambda: host_fed_tx_resident_rate(ctx, cfg, sc, axes))
    leg("host_fed_reference_flowgraph_shape", lambda: flowgraph_shape_host_fed(ctx))
    leg("per_block_drop_in", per_block_drop_in)
    try:
        import bench_extra as be
    except Exception as ex:
        out["error"] = repr(ex)
        return out

    ROOF = ("algorithmic_bytes_per_step", "GBps_algorithmic", "frac_of_hbm_peak")   # the three fields every secondary leg with kernels of its own carries

    def _demod():
        r = be.radar_with_demod(cfg if cfg in ("B", "D") else "B", 512 if cfg != "D" else 256)
        return dict({"frames_per_s": r["frames_per_s"], "ms_per_step": r["ms_per_step"], "frames_per_step": r["frames_per_step"],
                     "what": "A6+A7+A1 as one kernel (time-domain RX in) -> A2..A5, config " + (cfg if cfg in ("B", "D") else "B")}, **{k: r[k] for k in ROOF})

    def _eq():
        e = be.equalizer_config_c()
        return dict({"frames_per_s": e["frames_per_s"], "lane_frames_per_s": e["lane_frames_per_s"], "GBps": e["GBps"], "what": e["what"]}, **{k: e[k] for k in ROOF})

    def _comm():
        import comm_rx_probe
        c = comm_rx_probe.run(4096, 200, 5)
        sec = c["frames"] / c["frames_per_s"]
        g = c["M_samples_per_s"] * 1e6 * 8 / 1e9
        return {"frames_per_s": c["frames_per_s"], "M_samples_per_s": c["M_samples_per_s"], "crc_ok": c["crc_ok"],
                "frames": c["frames"], "payloads_intact": c["payloads_intact"], "what": c["what"] + " (200-byte PDUs, QPSK 1/2)",
                "algorithmic_bytes_per_step": int(c["M_samples_per_s"] * 1e6 * sec * 8), "GBps_algorithmic": g, "frac_of_hbm_peak": g / HBM_PEAK_GBS}

    def _pre():
        p = be.precoder_config_c()
        q = p["per-subcarrier_steering_+_radar_streams"]
        return dict({"packets_per_s_dft": p["dft"]["frames_per_s"], "packets_per_s_steering_and_radar_streams": q["frames_per_s"], "what": p["what"],
                     "roofline_is_of": "the per-subcarrier steering + radar streams form", "dft_form": {k: p["dft"][k] for k in ROOF}}, **{k: q[k] for k in ROOF})

    def _rd():
        d = be.range_doppler("D", 64)      # a device-resident batch of >= 64 config-D frames (8.6 GB of map)
        return dict({"frames_per_s": d["frames_per_s"], "frames_per_step": d["frames_per_step"], "what": d["what"]}, **{k: d[k] for k in ROOF})

    leg("chain_with_rx_demod", _demod)
    if hasattr(be, "detect_only"):
        leg("detect_only_chain", lambda: be.detect_only(cfg if cfg in ("B", "D") else "B"))
        # the same mode on frames of noise alone: the case its exact pruning of the angle stage gains least from
        leg("detect_only_chain_noise_only_frames", lambda: be.detect_only(cfg if cfg in ("B", "D") else "B", noise_only=True))
    if hasattr(be, "power_map"):
        leg("power_map_chain", lambda: be.power_map(cfg if cfg in ("B", "D") else "B"))
    if hasattr(be, "device_resident_flowgraph"):
        def _drf():
            r = be.device_resident_flowgraph(64)
            r["packets_per_s_at_256_per_pass"] = be.device_resident_flowgraph(256)["frames_per_s"]     # the same leg when a pass holds four times the packets
            g = be.device_resident_flowgraph_grc(512)                                                  # and at the reference flowgraph's own geometry (4x2, 64 subcarriers)
            r["grc_geometry"] = {k: g[k] for k in ("what", "frames_per_step", "ms_per_step", "frames_per_s", "packet0")}
            return r
        leg("device_resident_sim_flowgraph", _drf)
    leg("equalizer_config_c", _eq)
    if hasattr(be, "sync_front_end"):
        leg("sync_front_end", lambda: be.sync_front_end(4096))
    leg("comm_rx_chain", _comm)
    leg("precoder_config_c", _pre)
    leg("range_doppler_config_d", _rd)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# the contract's line.  The full record (`out` of main(): every secondary leg with its prose, per-window times, per-rank identities) is written
# to --verbose-out and to stderr; the LAST stdout line is this compact form, held under LINE_LIMIT bytes whatever the legs grow to — round 5's
# 21 KB line was not parsed by the driver, and took the headline down with it.
# ----------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _rnd(v, sig=6):
    if isinstance(v, float):
        if v != v or abs(v) == float("inf"):
            return None
        if v.is_integer() and abs(v) < 2.0 ** 53:
            return int(v)                    # byte and frame counts stay exact
        return float("%.*g" % (sig, v))
    if isinstance(v, dict):
        return {k: _rnd(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_rnd(x, sig) for x in v]
    return v


_LEG_VALUE_KEYS = ("frames_per_s", "packets_per_s_steering_and_radar_streams", "packets_per_s_dft", "radar_chain_block_frames_per_s", "M_samples_per_s")


def compact_secondary(sec):
    """{leg: {value, unit, frac_of_hbm_peak}} — numbers only"""
    out = {}
    for name, leg in (sec or {}).items():
        if not isinstance(leg, dict):
            continue
        if "error" in leg:
            out[name] = {"error": str(leg["error"])[:80]}
            continue
        if name == "per_block_drop_in":       # GPU call / CPU block, p50 microseconds, at the .grc point and at config B
            e = {}
            for point in ("grc", "B"):
                for blk, b in (leg.get(point, {}).get("blocks", {}) or {}).items():
                    e.setdefault(blk, {})[point] = [b.get("p50_us"), b.get("cpu_block_p50_us")]
            out[name] = {"unit": "p50 us [gpu call, cpu block]", "blocks": e}
            continue
        c = {}
        for k in _LEG_VALUE_KEYS:
            if k in leg:
                c["value"] = leg[k]
                c["unit"] = {"M_samples_per_s": "Msamples/s"}.get(k, "packets/s" if "packets" in k or "block" in k else "frames/s")
                break
        f = leg.get("frac_of_hbm_peak", leg.get("frac"))
        if f is not None:
            c["frac_of_hbm_peak"] = f
        out[name] = c
    return out


def compact_line(out, verbose_path=None):
    """the ONE stdout line of the contract from the full record: headline fields, roofline (+ chain), cpu_baseline (numbers), check, and the
    secondary legs as {value, unit, frac_of_hbm_peak}.  Guaranteed to serialise to fewer than LINE_LIMIT bytes: if it ever would not, the
    optional parts are dropped, largest first, and the line says which."""
    c = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = dict(out.get("config", {}))
    c["config"] = cfg
    r = out.get("roofline", {})
    c["roofline"] = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "algorithmic_bytes_per_launch",
                              "avg_launch_ms", "launches_timed", "launches_per_step", "frames_per_launch"))
    ch = r.get("chain")
    if ch:
        c["roofline"]["chain"] = dict(_pick(ch, ("compulsory_bytes_per_frame", "achieved", "frac", "ms_per_step", "kernels_sum_ms")),
                                      **({"a1": _pick(ch["a1"], ("kernel", "achieved", "frac", "avg_ms"))} if "a1" in ch else {}))
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "host_cpus"))
        c["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:120]
        if isinstance(cb.get("all_cores"), dict):
            c["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
        if isinstance(cb.get("pipeline_ideal"), dict):
            c["cpu_baseline"]["pipeline_ideal"] = _pick(cb["pipeline_ideal"], ("value", "cores", "value_with_tuned_fft"))
    c["windows"] = _pick(out.get("windows", {}), ("n", "ms_per_step_median", "ms_per_step_min", "ms_per_step_max"))
    c["kernels_ms"] = out.get("kernels_ms")
    c["check"] = _pick(out.get("check", {}), ("ok", "ok_all_ranks", "frames_checked", "oracle_frames", "tiled_bit_equal", "chanest_bit_exact",
                                             "map_max_rel_err", "map_tol", "estimator_fields_exact", "range_m", "angle_deg", "snr_db"))
    if out.get("with_results_on_host"):
        c["with_results_on_host"] = _pick(out["with_results_on_host"], ("frames_per_s", "ms_per_step", "records_complete_on_host", "records_equal_headline_run"))
    c.update(_pick(out, ("device", "backend", "collective_world", "distinct_devices", "value_over_n_times_slowest_rank")))
    c["ranks"] = [_pick(rk, ("rank", "device_key", "pid", "ms_per_step_window0")) for rk in out.get("ranks", [])]
    if "secondary" in out:
        c["secondary"] = compact_secondary(out["secondary"])
    if verbose_path:
        c["verbose_record"] = verbose_path
    c = _rnd(c)
    dropped = []
    for victim in (None, "secondary.per_block_drop_in", "ranks", "secondary", "with_results_on_host", "windows", "kernels_ms"):
        if victim:
            top, _, sub = victim.partition(".")
            if sub:
                if isinstance(c.get(top), dict) and sub in c[top]:
                    del c[top][sub]
                    dropped.append(victim)
            elif top in c:
                del c[top]
                dropped.append(victim)
            if dropped:
                c["dropped_to_fit"] = dropped
        line = json.dumps(c, separators=(",", ":"), allow_nan=False)
        if len(line) < LINE_LIMIT:
            return line
    raise RuntimeError("compact_line: %d bytes after dropping every optional part" % len(line))


# ----------------------------------------------------------------------------------------------------------------------
def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.spawn):
        sys.exit(self_launch(a, [x for x in argv if x != "--spawn"]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    a.gpus = world

    import torch
    import torch.distributed as dist
    import jrc_amd
    from jrc_amd import shard, synth

    sc = scenario(a.config)
    Ir, Ia = a.interp_range, 16
    P, NR, NA = sc.T * sc.R, sc.N * Ir, sc.T * sc.R * Ia
    F = a.frames or {"A": 4096, "B": 512, "D": 256}[a.config]
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ndr = 2 * 3e8 / (2 * sc.fs)
    nda = 2 * float(np.rad2deg(np.arcsin(2 / P))) if P > 2 else 30.0
    axes = (rb, ab, ndr, nda)
    n_stream = a.stream_frames if a.stream_frames > 0 else world * F   # frames of the global stream per step
    if a.stream_frames > 0:
        if a.stream_frames < world:
            sys.exit("bench.py: --stream-frames must give every rank a frame")
        lo, hi = shard.frame_shard(n_stream, rank, world)
        F = hi - lo                                                 # ragged: the blocks differ by at most one frame
    n_distinct = min(a.distinct, F)
    first_frame = shard.frame_shard(n_stream, rank, world)[0]       # this rank's block of the global frame stream
    host_frames = synth.make_frames(sc, n_distinct, first_frame=first_frame)
    cpu_base = None
    expect = []
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # CPU leg first: it forks worker processes, which must happen before this process initialises the GPU
        cpu_base = cpu_baseline(sc, Ir, Ia, host_frames, axes, a.cpu_seconds)
        cpu_base["host_cpus"] = os.cpu_count()
    if rank == 0 and not a.no_check:
        k = a.oracle_frames if a.oracle_frames >= 0 else {"A": 32, "B": 32, "D": 8}[a.config]
        expect = oracle_expect(sc, Ir, Ia, host_frames[:min(k, n_distinct)])

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path is mandatory; there is no CPU fallback)")
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(a.backend)
    coll_dev = dev if a.backend == "nccl" else "cpu"

    ctx = jrc_amd.Context(local_rank)
    chain = jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, ndr, nda, 15.0, 0.0,
                               max_frames=F, ctx=ctx)
    bufs = chain.alloc(F, dev)
    hf = torch.from_numpy(host_frames.view(np.float32).reshape((n_distinct,) + tuple(bufs["frames"].shape[1:])))
    for f0 in range(0, F, n_distinct):
        n = min(n_distinct, F - f0)
        bufs["frames"][f0:f0 + n].copy_(hf[:n])
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    gathered = [None]
    do_gather = a.gather_results and world > 1
    k_maps = min(a.gather_maps, n_stream // world) if world > 1 else 0

    def step():
        chain.run(bufs, F)
        if do_gather or k_maps:
            # optional exchange step (RCCL / xGMI).  The chain runs on the context's stream, the collective on torch's: both are
            # drained around it so that the exchange reads finished maps and its cost is part of the step.
            ctx.sync()
            if do_gather:                     # per-frame result records
                gathered[0] = shard.gather_results(bufs["results"], n_stream)
            if k_maps:                        # K maps per rank
                gathered[0] = shard.gather_maps(bufs["map"][:k_maps], world * k_maps)
            torch.cuda.synchronize()

    if a.prewarm_seconds > 0:                 # untimed: bring a freshly booted GPU out of its idle power state
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < a.prewarm_seconds:
            chain.run(bufs, F)
            ctx.sync()
    for _ in range(a.warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    # Window 0 is the contract's timed region: exactly K steps between barrier + synchronize, with NO instrumentation inside it.  The
    # per-kernel hipEvents the roofline record is made of (two events around every launch, on the chain's own stream) are switched on for
    # windows 1.. — the same K steps on the same resident batch, directly behind window 0 — so `roofline.avg_launch_ms` is measured live in
    # this run, but not inside the region `value` is taken from.  (With --windows 1 there is only window 0, and it carries the events.)
    n_windows = max(1, a.windows)
    events_in_window0 = n_windows == 1
    chain.set_timing(events_in_window0)
    windows = []
    for w in range(n_windows):
        if w == 1:
            chain.set_timing(True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        ctx.sync()
        torch.cuda.synchronize()
        barrier()
        windows.append(time.perf_counter() - t0)
    kt = chain.get_timing()
    res = chain.results(bufs, F)

    # The same steps with A5 finished INSIDE the timed region (a frame counts "through A5"): the records of step n are
    # copied on the chain's copy stream while step n+1 runs (two record buffers alternate), and the host completes snr_est / published
    # of step n-1 (one log10f per frame, the reference's own libm call) — every step's records are on the host, complete, when the
    # window closes.  Timed without the per-kernel events.
    with_results = None
    if world == 1 and not (do_gather or k_maps):
        chain.set_timing(False)
        rbuf = [bufs["results"], torch.empty_like(bufs["results"])]
        views = [dict(bufs, results=rbuf[0]), dict(bufs, results=rbuf[1])]
        host = [(jrc_amd.RaResult * F)() for _ in range(2)]              # the host's record arrays, alternating like the device's
        counts = []

        def step_r(i):
            chain.run(views[i & 1], F)
            chain.results_begin(rbuf[i & 1], F)
            if i >= 1:
                counts.append(chain.results_end(into=host[(i - 1) & 1]))

        for i in range(4):
            step_r(i)
        chain.results_end(into=host[1])
        ctx.sync()
        torch.cuda.synchronize()
        wr = []
        for w in range(max(1, a.windows)):
            counts.clear()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step_r(i)
            counts.append(chain.results_end(into=host[(a.steps - 1) & 1]))   # the last step's records: the window ends with everything on the host
            wr.append(time.perf_counter() - t0)
        last = host[(a.steps - 1) & 1]
        same = all((r.peak_range_idx, r.peak_angle_idx, r.snr_est, r.published, r.noise_power) ==
                   (q.peak_range_idx, q.peak_angle_idx, q.snr_est, q.published, q.noise_power) for r, q in zip(last, res))
        wr_ms = sorted(1e3 * w / a.steps for w in wr)
        with_results = {"what": "the same %d steps with every step's records copied to the host and A5's snr_est / published completed there "
                                "INSIDE the timed region (copy of step n beside the kernels of step n+1; jrc_chain_fetch_results_begin / _end)" % a.steps,
                        "frames_per_s": F * a.steps / wr[0], "ms_per_step": 1e3 * wr[0] / a.steps, "ms_per_step_median": wr_ms[len(wr_ms) // 2],
                        "ms_per_step_min": wr_ms[0], "ms_per_step_max": wr_ms[-1],
                        "records_complete_on_host": len(counts) == a.steps and all(c == F for c in counts),
                        "records_equal_headline_run": bool(same)}
        chain.set_timing(True)

    ranks = gather_identities(dist, world, rank_identity(torch, rank, local_rank, torch.cuda.current_device(), windows, a.steps))
    windows = shard.max_over_ranks_vec(windows, coll_dev)
    elapsed = windows[0]
    same_bus = sorted(r["device_key"] for r in ranks)
    shared_device = any(x == y for x, y in zip(same_bus, same_bus[1:]))

    check = None
    if not a.no_check:
        check = verify(torch, bufs, bufs["results"], res, F, n_distinct, expect, axes, sc)
        ok_all = shard.min_over_ranks(1.0 if check["ok"] else 0.0, coll_dev)
        check["ok_all_ranks"] = bool(ok_all > 0.5)

    if a.dump:
        ctx.sync()
        allr = shard.gather_results(bufs["results"], n_stream)
        km = min(a.dump_maps, n_stream // world)
        allm = shard.gather_maps(bufs["map"][:km].contiguous(), world * km) if km else None
        allh = shard.gather_results(bufs["chanest"], n_stream)
        if rank == 0:
            np.savez(a.dump, results=allr.cpu().numpy(), chanest=allh.cpu().numpy(),
                     maps=(allm.cpu().numpy() if allm is not None else np.zeros(0, np.float32)))

    rc = 0
    if rank == 0:
        total_frames = n_stream * a.steps
        ms_step = 1e3 * elapsed / a.steps
        per = sorted(1e3 * w / a.steps for w in windows)
        out = {
            "metric": "ofdm_frames_per_sec", "value": total_frames / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if a.stream_frames > 0 else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "config %s: %dTx x %dRx, %d subcarriers, %d symbols, Ir=%d, Ia=%d -> %dx%d range-angle map + "
                                   "peak/SNR estimate (mimo_ofdm_radar -> range IFFT -> transpose -> angle FFT -> "
                                   "range_angle_estimator), %d frames/GPU/step resident in HBM"
                                   % (a.config, sc.T, sc.R, sc.N, sc.S, Ir, Ia, NR, NA, F),
                       "frames_per_gpu_per_step": F, "stream_frames_per_step": n_stream, "parallelism": "frame-sharded x%d, no data-path collective" % world,
                       "gather_results": bool(do_gather), "gather_maps_per_gpu": int(k_maps),
                       "launcher": "self-launched children" if os.environ.get("JRC_BENCH_CHILD") else ("torch.distributed.run" if world > 1 else "direct")},
            "windows": {"n": len(per), "steps_each": a.steps, "ms_per_step_median": per[len(per) // 2], "ms_per_step_min": per[0],
                        "ms_per_step_max": per[-1], "ms_per_step_each": [1e3 * w / a.steps for w in windows],
                        "per_kernel_events_in_window0": events_in_window0,
                        "note": "`value` / `ms_per_step` are window 0 (the contract's timed region, no per-kernel events inside it); windows 1.. repeat "
                                "the same steps WITH two hipEvents around every kernel launch (the roofline's avg_launch_ms comes from those), "
                                "which is why they, not window 0, may be the slower ones; max over ranks per window"},
            "roofline": dict(roofline_of(chain, kt, a.config, sc, Ir, Ia, F), chain=chain_roofline(sc, Ir, Ia, F, ms_step, kt)),
            "kernels_ms": {k: kt[k] for k in ("radar_chanest", "range_angle_fused", "ra_finalize")},
            "host_epilogue": "snr_est = 10 log10(peak/noise) and the publish decision of A5 (one log10f per frame) are finished on the host in "
                             "jrc_chain_fetch_results, after the timed region of `value`; `with_results_on_host` times the same steps with that "
                             "inside the region",
            "with_results_on_host": with_results,
            "check": dict(check or {}, range_m=res[0].range_val, angle_deg=res[0].angle_val, snr_db=res[0].snr_est),
            "device": ctx.device_name(),
            # who took part: one entry per rank with the physical device it ran on and ITS OWN clock over the same windows
            "ranks": ranks,
            "backend": (dist.get_backend() if world > 1 else None),
            "collective_world": (dist.get_world_size() if world > 1 else 1),
            "distinct_devices": len(set(r["device_key"] for r in ranks)),
        }
        slowest = max(r["ms_per_step_window0"] for r in ranks)
        # consistency of the line with its parts (barrier cost + start skew), NOT the scaling efficiency (the driver computes that from the
        # per-N lines): whole-job value over N x the rate the slowest rank measured on its own clock
        out["value_over_n_times_slowest_rank"] = (total_frames / elapsed) / (n_stream / (slowest * 1e-3))
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        if world == 1 and not a.no_secondary:
            chain.set_timing(False)
            chain.close()
            bufs.clear()                         # the secondary legs allocate their own batches (config D: 6 GiB)
            torch.cuda.empty_cache()
            out["secondary"] = secondary_figures(a.config, ctx, sc, axes)
        # full record: to the file and to stderr; the contract's ONE stdout line is the compact form (< LINE_LIMIT bytes)
        vpath = a.verbose_out or os.path.join(ROOT, "gpurun_out", "bench_verbose.json")
        try:
            os.makedirs(os.path.dirname(os.path.abspath(vpath)), exist_ok=True)
            with open(vpath, "w") as fh:
                json.dump(out, fh)
                fh.write("\n")
        except OSError as ex:
            sys.stderr.write("bench.py: verbose record not written to %s: %s\n" % (vpath, ex))
            vpath = None
        sys.stderr.write("bench.py verbose record: " + json.dumps(out) + "\n")
        sys.stderr.flush()
        print(compact_line(out, os.path.relpath(vpath, ROOT) if vpath else None))
        sys.stdout.flush()
    if shared_device and not a.same_device:
        sys.stderr.write("bench.py: two ranks ran on the same device (%s): not an N-GPU measurement\n" % ", ".join(same_bus))
        rc = 4
    if check is not None and not check.get("ok_all_ranks", check["ok"]):
        sys.stderr.write("bench.py: result check FAILED on rank %d: %s\n" % (rank, json.dumps(check)))
        rc = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
