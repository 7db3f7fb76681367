#!/usr/bin/env python3
"""bench.py — OFDM frames/s of the radar hot path (A1 -> A5) on MI355X, with the roofline of the dominant
kernel and a CPU baseline beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config B|D|A] [--frames F]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the fused chain (mimo_ofdm_radar -> range IFFT -> transpose/pad -> angle FFT ->
range_angle_estimator) over one batch of F synthetic frames already resident in HBM.  Frames are
independent, so N GPUs each process their own batch (weak scaling, no data-path collective); the only
collectives are the barriers / MAX-reduce of the timing contract.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="B", choices=["A", "B", "D"])
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (0 = per-config default)")
    ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic frames generated per GPU (tiled to --frames)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures of SURVEY §8(d) (chain incl. RX demod, equalizer / precoder config C) that rank 0 "
                         "appends at N=1 after the timed region")
    ap.add_argument("--prewarm-seconds", type=float, default=0.3,
                    help="untimed runs before the W warm-up steps so a cold GPU has its clocks up (0 = none)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests of the N>1 path)")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--gather-results", action="store_true",
                    help="also RCCL all-gather the per-frame results every step (optional exchange, off by default)")
    ap.add_argument("--gather-maps", type=int, default=0, metavar="K",
                    help="also RCCL all-gather the range-angle maps of the first K frames of every rank each step (optional exchange, off by default)")
    return ap.parse_args()


def scenario(name):
    from jrc_amd import synth
    return {"A": synth.config_A, "B": synth.config_B, "D": synth.config_D}[name]()


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
    return {k: v / len(frames) for k, v in acc.items()}


def cpu_baseline(sc, Ir, Ia, frames, axes, budget_s):
    """the oracle's block-by-block chain (A1..A5, float32) timed on a bounded sample of the same frames:
    `value` = one thread (the reference's per-block regime); `all_cores` = frame-parallel over every host core."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    sc_args = (sc.N, sc.T, sc.R, sc.S, sc.Npre)
    done, el = _cpu_worker((sc_args, Ir, Ia, frames, axes, budget_s * 0.6))
    out = {"value": done / el, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same workload, oracle C port (float32 radix-2 FFTs), 1 thread, %.1f s" % (done, el)}
    try:
        # (b) pipeline-ideal: GNU Radio runs one thread per block, so a saturated flowgraph moves at the pace of its slowest block
        st = _cpu_stage_times(sc_args, Ir, Ia, frames[:3], axes)
        out["pipeline_ideal"] = {"value": 1.0 / max(st.values()), "unit": "frames/s", "cores": len(st),
                                 "stage_ms": {k: 1e3 * v for k, v in st.items()},
                                 "sample": "1 / slowest block of the five (thread-per-block scheduling), %d frames block by block" % len(frames[:3])}
    except Exception as e:
        out["pipeline_ideal"] = {"error": str(e)}
    try:
        ncpu = len(os.sched_getaffinity(0))
        n = max(1, min(ncpu, 64))
        with mp.get_context("fork").Pool(n) as pool:
            res = pool.map(_cpu_worker, [(sc_args, Ir, Ia, frames[:4], axes, budget_s * 0.3)] * n)
        out["all_cores"] = {"value": sum(d / e for d, e in res), "unit": "frames/s", "cores": n,
                            "sample": "%d frames over %d processes" % (sum(d for d, _ in res), n)}
    except Exception as e:      # the single-thread figure above is the contract; this one is extra
        out["all_cores"] = {"error": str(e)}
    return out


def secondary_figures(cfg):
    """the same chain with the RX OFDM demod in front (time-domain RX in), and the comm-side config C — measured after
    the timed region by tools/bench_extra.py's routines (their own batches, a few seconds in total); never part of `value`"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    out = {}
    try:
        import bench_extra as be
        r = be.radar_with_demod(cfg if cfg in ("B", "D") else "B", 512 if cfg != "D" else 256)
        out["chain_with_rx_demod"] = {"frames_per_s": r["frames_per_s"], "ms_per_step": r["ms_per_step"], "frames_per_step": r["frames_per_step"],
                                      "what": "A6+A7+A1 as one kernel (time-domain RX in) -> A2..A5, config " + (cfg if cfg in ("B", "D") else "B")}
        e = be.equalizer_config_c()
        out["equalizer_config_c"] = {"frames_per_s": e["frames_per_s"], "lane_frames_per_s": e["lane_frames_per_s"], "GBps": e["GBps"],
                                     "what": e["what"]}
        import comm_rx_probe
        c = comm_rx_probe.run(4096, 200, 5)
        out["comm_rx_chain"] = {"frames_per_s": c["frames_per_s"], "M_samples_per_s": c["M_samples_per_s"], "crc_ok": c["crc_ok"],
                                "frames": c["frames"], "payloads_intact": c["payloads_intact"], "what": c["what"] + " (200-byte PDUs, QPSK 1/2)"}
        p = be.precoder_config_c()
        out["precoder_config_c"] = {"packets_per_s_dft": p["dft"]["frames_per_s"],
                                    "packets_per_s_steering_and_radar_streams": p["per-subcarrier_steering_+_radar_streams"]["frames_per_s"],
                                    "what": p["what"]}
        d = be.range_doppler("D", 16)
        out["range_doppler_config_d"] = {"frames_per_s": d["frames_per_s"], "GBps_algorithmic": d["GBps_algorithmic"],
                                         "frames_per_step": d["frames_per_step"], "what": d["what"]}
    except Exception as ex:            # secondary figures must never take the headline line down with them
        out["error"] = repr(ex)
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world

    import torch
    import torch.distributed as dist
    import jrc_amd
    from jrc_amd import shard, synth

    sc = scenario(a.config)
    Ir, Ia = 8, 16
    P, NR, NA = sc.T * sc.R, sc.N * Ir, sc.T * sc.R * Ia
    F = a.frames or {"A": 4096, "B": 512, "D": 256}[a.config]
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ndr = 2 * 3e8 / (2 * sc.fs)
    nda = 2 * float(np.rad2deg(np.arcsin(2 / P))) if P > 2 else 30.0
    n_distinct = min(a.distinct, F)
    first_frame = shard.frame_shard(world * F, rank, world)[0]      # this rank's block of the global frame stream
    host_frames = synth.make_frames(sc, n_distinct, first_frame=first_frame)
    cpu_base = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # CPU leg first: it forks worker processes, which must happen before this process initialises the GPU
        cpu_base = cpu_baseline(sc, Ir, Ia, host_frames, (rb, ab, ndr, nda), a.cpu_seconds)
        cpu_base["host_cpus"] = os.cpu_count()

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
    k_maps = min(a.gather_maps, F) if world > 1 else 0

    def step():
        chain.run(bufs, F)
        if do_gather:                         # optional exchange step: per-frame result records over RCCL
            ctx.sync()
            gathered[0] = shard.gather_results(bufs["results"], world * F)
        if k_maps:                            # optional exchange step: K maps per rank over RCCL / xGMI
            ctx.sync()
            gathered[0] = shard.gather_maps(bufs["map"][:k_maps], world * k_maps)

    if a.prewarm_seconds > 0:                 # untimed: bring a freshly booted GPU out of its idle power state
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < a.prewarm_seconds:
            chain.run(bufs, F)
            ctx.sync()
    for _ in range(a.warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    chain.set_timing(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kt = chain.get_timing()
    res = chain.results(bufs, F)

    elapsed = shard.max_over_ranks(elapsed, dev if a.backend == "nccl" else "cpu")

    if rank == 0:
        total_frames = F * a.steps * world
        ms_step = 1e3 * elapsed / a.steps
        # algorithmic bytes of the dominant kernel: unpadded H in, complex map out
        alg_bytes = F * (P * sc.N * 8 + NR * NA * 8)
        traffic = None
        try:   # PMC-measured HBM bytes of the same kernel/config, collected in separate rocprofv3 --pmc passes
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(a.config)
            if pt and pt["frames_per_launch"] == F:
                traffic = pt["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        # a batch beyond one resident wave of workgroups runs as several launches of the dominant kernel: roofline per launch
        n_launch = max(1, chain.launches_per_run(F))
        f_launch = F // n_launch if F % n_launch == 0 else None
        k_ms = kt["range_angle_fused"] / n_launch
        alg_bytes = alg_bytes / n_launch
        if traffic is None and f_launch is not None:
            try:
                pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(a.config)
                if pt and pt["frames_per_launch"] == f_launch:
                    traffic = pt["hbm_bytes_per_launch"]
            except (OSError, ValueError, KeyError):
                pass
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "ofdm_frames_per_sec", "value": total_frames / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "config %s: %dTx x %dRx, %d subcarriers, %d symbols, Ir=%d, Ia=%d -> %dx%d range-angle map + "
                                   "peak/SNR estimate (mimo_ofdm_radar -> range IFFT -> transpose -> angle FFT -> "
                                   "range_angle_estimator), %d frames/GPU/step resident in HBM"
                                   % (a.config, sc.T, sc.R, sc.N, sc.S, Ir, Ia, NR, NA, F),
                       "frames_per_gpu_per_step": F, "parallelism": "frame-sharded x%d, no data-path collective" % world,
                       "gather_results": bool(do_gather), "gather_maps_per_gpu": int(k_maps)},
            "roofline": {"bound": "hbm", "kernel": "range_angle_fused_kernel<%d>" % P, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": k_ms, "launches_timed": kt["launches"] * n_launch,
                         "launches_per_step": n_launch, "frames_per_launch": F / n_launch},
            "kernels_ms": {k: kt[k] for k in ("radar_chanest", "range_angle_fused", "ra_finalize")},
            "check": {"range_m": res[0].range_val, "angle_deg": res[0].angle_val, "snr_db": res[0].snr_est},
            "device": ctx.device_name(),
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        if world == 1 and not a.no_secondary:
            out["secondary"] = secondary_figures(a.config)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
