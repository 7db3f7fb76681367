#!/usr/bin/env python3
"""Secondary figures SURVEY.md §8(d) asks for next to bench.py's headline line (tools only; prints JSON lines):
  * radar chain WITH the RX OFDM demod in front (A6 cyclic-prefix removal + A7 fft fwd/shift fused, then A1..A5)
  * the device-resident simulation flowgraph: 4 target simulators (one per TX, summed into the RX streams) -> RX OFDM demod
    -> A1..A5, per simulated frame
  * equalizer path, config C: 4 RX lanes x frames of [2 L-LTF, SIG, 4 MIMO-LTF, 64 data symbols] x 256 subcarriers, LS, DATA
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import jrc_amd
from jrc_amd import synth


def timed(fn, steps=30, warm=5):
    """seconds per call: median of three timed thirds of `steps` (a one-off stall on the box then does not move the figure)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    per = []
    k = max(1, steps // 3)
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / k)
    return sorted(per)[1]


def roof(alg_bytes_per_step, seconds_per_step):
    """the three fields every secondary leg carries: algorithmic bytes of one step (inputs read once + outputs written once, SURVEY §8(d) counting),
    their rate, and that rate against the 8 TB/s HBM peak (/opt/skills/guides/MI355X_MICROARCH.md)"""
    g = alg_bytes_per_step / seconds_per_step / 1e9
    return dict(algorithmic_bytes_per_step=int(alg_bytes_per_step), GBps_algorithmic=g, frac_of_hbm_peak=g / 8000.0)


def radar_with_demod(cfg="B", F=256):
    sc = {"B": synth.config_B, "D": synth.config_D}[cfg]()
    Ir, Ia, P = 8, 16, sc.T * sc.R
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ctx = jrc_amd.Context(0)
    chain = jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 14.36, 15.0, 0.0, max_frames=F, ctx=ctx)
    bufs = chain.alloc(F, "cuda:0")
    fr = synth.make_frames(sc, 16)
    hf = torch.from_numpy(fr.view(np.float32).reshape((16,) + tuple(bufs["frames"].shape[1:])))
    for f0 in range(0, F, 16):
        bufs["frames"][f0:f0 + 16].copy_(hf[:min(16, F - f0)])
    n_items = sc.Npre + sc.S
    # time-domain RX streams with cyclic prefix: [F][R][n_items*(N+cp)] (content irrelevant for timing: random)
    td = torch.randn((F, sc.R, n_items * (sc.N + sc.cp), 2), dtype=torch.float32, device="cuda:0") * 1e-3
    rx_view = bufs["frames"][:, sc.T:]                      # the RX ports of the frame buffer, [F][R][n_items][N][2]
    assert rx_view.is_contiguous() is False
    rx_tmp = torch.empty((F, sc.R, n_items, sc.N, 2), dtype=torch.float32, device="cuda:0")
    L = ctx.lib
    torch.cuda.synchronize()

    def step():
        # A6+A7 for all F*R streams in one launch (n_symbols = F*R*n_items, symbols are contiguous per stream)
        ctx.check(L.jrc_cp_remove_fft_dev(ctx.h, sc.N, sc.cp, F * sc.R * n_items, td.data_ptr(), rx_tmp.data_ptr(), None))
        chain.run(bufs, F)
    t_unfused = timed(step)
    t_chain = timed(lambda: chain.run(bufs, F))
    d_tx = bufs["frames"][:, :sc.T].contiguous()
    t = timed(lambda: chain.run_td(bufs, d_tx, td, F, sc.cp))        # A6+A7+A1 in one kernel (jrc_chain_run_td_dev)
    # algorithmic bytes: the S radar symbols of the T reference ports (frequency domain) and of the R time-domain RX streams (with their cyclic
    # prefixes) read once, the complex map written once
    alg = F * ((sc.T * sc.S * sc.N + sc.R * sc.S * (sc.N + sc.cp)) * 8 + (sc.N * Ir) * (P * Ia) * 8)
    return dict(what="radar chain incl. RX OFDM demod (A6+A7+A1 one kernel, time-domain RX in) config %s" % cfg, frames_per_step=F, **roof(alg, t),
                ms_per_step=t * 1e3, frames_per_s=F / t, ms_per_step_separate_demod=t_unfused * 1e3,
                frames_per_s_separate_demod=F / t_unfused, ms_chain_only=t_chain * 1e3, frames_per_s_chain_only=F / t_chain)


def detect_only(cfg="B", F=None, noise_only=False):
    """detect-only chain mode (SURVEY §8(d) 'if the map is not materialised'): A1 -> fused transforms + arg-max without map stores ->
    noise-window rows -> estimator epilogue; algorithmic bytes per frame = inputs + the 48-byte result.
    noise_only: RX ports hold noise alone — the worst case of the bound-pruned angle stage (nothing stands out, little is skipped)"""
    sc = {"B": synth.config_B, "D": synth.config_D}[cfg]()
    F = F or int(os.environ.get("JRC_DETECT_FRAMES", "0")) or (512 if cfg == "B" else 256)
    Ir, Ia, P = 8, 16, sc.T * sc.R
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ctx = jrc_amd.Context(0)
    chain = jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 14.36, 15.0, 0.0, max_frames=F, ctx=ctx)
    bufs = chain.alloc(F, "cuda:0")
    fr = synth.make_frames(sc, 8)
    if noise_only:
        g = np.random.default_rng(5)
        fr[:, sc.T:] = ((g.standard_normal(fr[:, sc.T:].shape) + 1j * g.standard_normal(fr[:, sc.T:].shape)) * 1e-3).astype(np.complex64)
    hf = torch.from_numpy(fr.view(np.float32).reshape((8,) + tuple(bufs["frames"].shape[1:])))
    for f0 in range(0, F, 8):
        bufs["frames"][f0:f0 + 8].copy_(hf[:min(8, F - f0)])
    torch.cuda.synchronize()
    chain.run(bufs, F)
    ctx.sync()
    want = bytes(bufs["results"].cpu().numpy().tobytes())
    t_map = timed(lambda: chain.run(bufs, F))
    chain.set_write_map(False)
    del bufs["map"]
    bufs["map"] = None
    torch.cuda.empty_cache()
    chain.run(bufs, F)
    ctx.sync()
    same = bytes(bufs["results"].cpu().numpy().tobytes()) == want
    t = timed(lambda: chain.run(bufs, F))                  # the entry point as a caller gets it (sliced pipeline, chain.hip chain_run)
    chain.set_timing(True)                                 # per-kernel events: the kernels one after the other on one stream
    t_serial = timed(lambda: chain.run(bufs, F))
    kt = chain.get_timing()
    chain.set_timing(False)
    chain.run(bufs, F)
    ctx.sync()
    same = same and bytes(bufs["results"].cpu().numpy().tobytes()) == want
    alg = F * ((sc.T + sc.R) * sc.S * sc.N * 8 + 48)
    return dict(what="detect-only chain (no map stored; results bit-identical to map mode), config %s, %d frames per step%s" % (cfg, F, ", NOISE-ONLY frames (worst case of the pruned angle stage)" if noise_only else ""),
                frames_per_step=F, ms_per_step=t * 1e3, frames_per_s=F / t, results_equal_map_mode=bool(same),
                algorithmic_bytes_per_frame=alg // F, **roof(alg, t),
                kernels_ms={"radar_chanest": kt["radar_chanest"], "fused_detect_plus_window": kt["range_angle_fused"], "ra_finalize": kt["ra_finalize"]},
                ms_per_step_map_mode=t_map * 1e3, ms_per_step_kernels_in_series=t_serial * 1e3)


def power_map(cfg="B", F=None):
    """power-map format: the chain with the map stored as float |z|^2 (the heat-map branch's stream), half the map bytes"""
    sc = {"B": synth.config_B, "D": synth.config_D}[cfg]()
    F = F or (512 if cfg == "B" else 256)
    Ir, Ia, P = 8, 16, sc.T * sc.R
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ctx = jrc_amd.Context(0)
    chain = jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 14.36, 15.0, 0.0, max_frames=F, ctx=ctx)
    chain.set_map_format(True)
    bufs = chain.alloc(F, "cuda:0", power_map=True)
    fr = synth.make_frames(sc, 8)
    hf = torch.from_numpy(fr.view(np.float32).reshape((8,) + tuple(bufs["frames"].shape[1:])))
    for f0 in range(0, F, 8):
        bufs["frames"][f0:f0 + 8].copy_(hf[:min(8, F - f0)])
    torch.cuda.synchronize()
    chain.run(bufs, F)
    ctx.sync()
    chain.set_timing(True)
    t = timed(lambda: chain.run(bufs, F))
    kt = chain.get_timing()
    alg = F * ((sc.T + sc.R) * sc.S * sc.N * 8 + chain.NR * chain.NA * 4 + 48)
    return dict(what="chain with the map as float |z|^2 (heat-map stream), config %s, %d frames per step" % (cfg, F), frames_per_step=F,
                ms_per_step=t * 1e3, frames_per_s=F / t, algorithmic_bytes_per_frame=alg // F, **roof(alg, t),
                kernels_ms={"radar_chanest": kt["radar_chanest"], "fused_power_plus_window": kt["range_angle_fused"], "ra_finalize": kt["ra_finalize"]})


def range_doppler(cfg="D", F=8, Id=1):
    """row D (no reference counterpart): D[p][sym][sc] = rx conj(tx) -> IFFT over subcarriers (N*Ir) -> FFT over symbols (S*Id, shifted)"""
    sc = {"B": synth.config_B, "D": synth.config_D}[cfg]()
    Ir, Ia, P = 8, 16, sc.T * sc.R
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ctx = jrc_amd.Context(0)
    chain = jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 14.36, 15.0, 0.0, max_frames=F, ctx=ctx)
    bufs = chain.alloc(F, "cuda:0")
    fr = synth.make_frames(sc, min(F, 8))
    hf = torch.from_numpy(fr.view(np.float32).reshape((fr.shape[0],) + tuple(bufs["frames"].shape[1:])))
    for f0 in range(0, F, fr.shape[0]):
        bufs["frames"][f0:f0 + fr.shape[0]].copy_(hf[:min(fr.shape[0], F - f0)])
    c = chain.cfg
    work = torch.empty((F, P, c.N_sym, chain.NR, 2), dtype=torch.float32, device="cuda:0")
    out = torch.empty((F, P, chain.NR, c.N_sym * Id, 2), dtype=torch.float32, device="cuda:0")
    L = ctx.lib
    import ctypes as C

    def step():
        ctx.check(L.jrc_range_doppler_dev(ctx.h, C.byref(chain.cfg), Id, F, bufs["frames"].data_ptr(), work.data_ptr(), out.data_ptr(), None))
    t = timed(step, steps=10, warm=2)
    alg = F * ((sc.T + sc.R) * sc.S * sc.N * 8 + P * chain.NR * sc.S * Id * 8)
    return dict(what="range-Doppler map (row D, build's own definition) config %s: %d pairs x %d range bins x %d Doppler bins per frame" % (cfg, P, chain.NR, sc.S * Id),
                frames_per_step=F, ms_per_step=t * 1e3, frames_per_s=F / t, **roof(alg, t))


def simulated_chain(cfg="B", F=64):
    """T target simulators (accumulating into the RX bursts) + RX demod + radar chain per frame, everything in HBM"""
    sc = {"B": synth.config_B, "D": synth.config_D}[cfg]()
    Ir, Ia, P = 8, 16, sc.T * sc.R
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ctx = jrc_amd.Context(0)
    chain = jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 14.36, 15.0, 0.0, max_frames=F, ctx=ctx)
    bufs = chain.alloc(F, "cuda:0")
    n_items = sc.Npre + sc.S
    n = (n_items + 3) * (sc.N + sc.cp)                     # burst incl. the 3 zero-pad symbols of the flowgraph
    lam = 3e8 / sc.fc
    tg = sc.targets or [(10.0 + 7 * k, -50.0 + 14 * k, 5.0 * k, 50.0) for k in range(8)]
    rng_m, az, vel, rcs = (np.array(v, np.float32) for v in zip(*tg))
    sims = [jrc_amd.target_simulator(rng_m, vel, rcs, az, [(r * sc.T + t + 2) * lam / 2 for r in range(sc.R)], int(sc.fs), sc.fc,
                                     sum_targets=True, max_bursts=F, ctx=ctx) for t in range(sc.T)]
    tx_td = [torch.randn((F, n, 2), dtype=torch.float32, device="cuda:0") * 0.1 for _ in range(sc.T)]
    rx_td = torch.zeros((F, sc.R, n, 2), dtype=torch.float32, device="cuda:0")
    rx_f = torch.empty((F, sc.R, n_items + 3, sc.N, 2), dtype=torch.float32, device="cuda:0")
    L = ctx.lib

    def sim_only():
        for t in range(sc.T):
            sims[t].run_dev(tx_td[t], rx_td, F, n, accumulate_out=(t > 0))

    d_tx = bufs["frames"][:, :sc.T].contiguous()

    def step():
        sim_only()
        chain.run_td(bufs, d_tx, rx_td, F, sc.cp)           # A6+A7+A1 one kernel; the streams carry 3 trailing pad symbols
    t = timed(step, steps=10, warm=2)
    ts = timed(sim_only, steps=10, warm=2)
    return dict(what="simulated frame: %d target simulators (%d targets, %d-sample bursts) + RX demod + radar chain, config %s"
                % (sc.T, len(tg), n, cfg), frames_per_step=F, ms_per_step=t * 1e3,
                frames_per_s=F / t, ms_simulators_only=ts * 1e3, bursts_per_s_simulators=F * sc.T / ts)


def device_resident_flowgraph(F=64):
    """the whole radar simulation flowgraph as one device-resident leg at config B's geometry (examples/radar_sim_device_resident.py):
    data symbols in HBM -> precoder -> OFDM modulator -> zero_pad -> 4 target simulators -> A6+A7+A1 -> A2..A5 -> records in HBM"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import radar_sim_device_resident as drm
    o = drm.config_b_tables()
    sim = drm.DeviceResidentRadarSim(o, 256, 4, 60, 64, F, trgt_range=[10.0], trgt_velocity=[0.0], trgt_rcs_dbsm=[20.0], trgt_angle=[20.0])
    rng = np.random.default_rng(1)
    sim.load_symbols(np.stack([drm.qpsk_symbols(rng, 60 * sim.nd) for _ in range(F)]))
    t = timed(lambda: sim.step(F), steps=12, warm=3)
    r = sim.results(F)[0]
    # algorithmic bytes of a pass = what has to cross HBM if every block of the graph read its input once and wrote its output once, block by
    # block (the graph's edges are the reference's stream buffers): symbols in, TX frequency-domain frames, TX time-domain, padded bursts, RX bursts, map
    T, R, N = sim.T, sim.R, sim.N
    edges = dict(symbols=sim.n_data * sim.nd, tx_f=T * sim.n_total * N, tx_t=T * sim.n_in, bursts=T * sim.n_burst, rx_t=R * sim.n_burst,
                 map=(N * 8) * (T * R * 16))
    alg = F * 8 * (edges["symbols"] + 2 * edges["tx_f"] + 2 * edges["tx_t"] + 2 * edges["bursts"] + 2 * edges["rx_t"] + edges["map"])
    alg_what = "every edge of the graph written once and read once (symbols, tx_f, tx_t, bursts, rx_t) + the map written: %s cf32 per packet" % edges
    return dict(what="device-resident simulation flowgraph, config B geometry (4x4, 256 subcarriers, radar window 64 symbols, %d-sample bursts): "
                     "precoder -> OFDM mod -> zero_pad -> 4 target simulators -> RX demod + A1 -> A2..A5, %d packets per pass, no host hop" % (sim.n_burst, F),
                frames_per_step=F, ms_per_step=t * 1e3, frames_per_s=F / t, **roof(alg, t), algorithmic_bytes_what=alg_what,
                simulators_summed_on_the_spectrum=bool(sim.sum_on_spectrum), packet0=dict(range_m=r.range_val, angle_deg=r.angle_val, snr_db=r.snr_est))


def device_resident_flowgraph_grc(F=512):
    """the same device-resident leg at the reference flowgraph's OWN geometry (…radar_sim.grc: 4 TX x 2 RX, fft_len 64, 100-byte PDUs = 18 data symbols,
    radar window N_pre 5 / N_sym 4, 2400-sample bursts, map 512 x 128), constant tables from tests/golden/ofdm_config_64.npz"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    import radar_sim_device_resident as drm
    o = np.load(os.path.join(root, "tests", "golden", "ofdm_config_64.npz"))
    o = {k: o[k] for k in o.files}
    sim = drm.DeviceResidentRadarSim(o, 64, 2, 18, 4, F, trgt_range=[10.0], trgt_velocity=[0.0], trgt_rcs_dbsm=[20.0], trgt_angle=[20.0])
    rng = np.random.default_rng(1)
    sim.load_symbols(np.stack([drm.qpsk_symbols(rng, 18 * sim.nd) for _ in range(F)]))
    t = timed(lambda: sim.step(F), steps=12, warm=3)
    r = sim.results(F)[0]
    return dict(what="device-resident simulation flowgraph at the .grc's own geometry (4x2, 64 subcarriers, %d-sample bursts), %d packets per pass" % (sim.n_burst, F),
                frames_per_step=F, ms_per_step=t * 1e3, frames_per_s=F / t, packet0=dict(range_m=r.range_val, angle_deg=r.angle_val, snr_db=r.snr_est))


def sync_front_end(n_frames=512):
    """capture in HBM -> frames of symbols (detection metrics, frame_detector, frame_sync run to completion)"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sync_probe.py"), "--frames", str(n_frames)],
                         capture_output=True, text=True).stdout.strip().splitlines()
    line = [l for l in out if l.startswith("capture")][-1]
    ms = float(line.split(":")[1].split("ms")[0])
    nsamp = int(line.split()[2])
    nf = int(line.split("frames found")[0].split(",")[-1])
    return dict(what="sync front end (metrics + frame_detector + frame_sync) on a %d-sample capture with %d frames" % (nsamp, nf),
                ms_per_capture=ms, M_samples_per_s=nsamp / ms / 1e3, frames_per_s=nf / ms * 1e3,
                algorithmic_bytes_what="the capture read once (8 B per sample); the frames' symbols written are < 10 % of it", **roof(nsamp * 8, ms * 1e-3))


def comm_rx_chain(n_frames=4096):
    """capture -> sync front end -> RX FFT -> equalizer -> Viterbi decoder, device-resident (tools/comm_rx_probe.py)"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "comm_rx_probe.py"), "--frames", str(n_frames), "--json"],
                         capture_output=True, text=True).stdout.strip().splitlines()
    r = json.loads([l for l in out if l.startswith("{")][-1])
    if "M_samples_per_s" in r and "frames_per_s" in r and r.get("frames"):
        sec = r["frames"] / r["frames_per_s"]
        r.update(algorithmic_bytes_what="the capture read once (8 B per sample); decoded payloads are bytes per frame", **roof(r["M_samples_per_s"] * 1e6 * sec * 8, sec))
    return r


def precoder_config_c(n_frames=2048):
    """C2 batched: 4 TX, 256 subcarriers, 64 data symbols per packet, DATA, per-subcarrier steering + 3 radar streams"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    N, T, S = 256, 4, 64
    rng = np.random.default_rng(0)
    guard = N // 16
    act = [c for c in range(-N // 2 + guard, N // 2 - guard + 1) if c != 0]
    pilots = [c for c in act if c % 32 == 16][:8]
    data = [c for c in act if c not in pilots]
    ltf = np.zeros(N, np.complex64)
    ltf[np.array(act) + N // 2] = rng.choice([-1.0, 1.0], len(act))
    mapped = np.stack([(synth.hadamard(T) * ltf[sc]).reshape(-1) for sc in range(N)]).astype(np.complex64)
    pil = np.array([[1, 1, 1, -1, 1, 1, 1, -1], [-1, -1, -1, 1, -1, -1, -1, 1], [1, 1, 1, -1, 1, 1, 1, -1]], np.complex64)[:, :len(pilots)]
    sync = np.stack([ltf, ltf, ltf, ltf])
    nd = len(data)
    nbytes = (S * nd - 22) // 8
    ctx = jrc_amd.Context(0)
    pre = jrc_amd.mimo_precoder(N, T, 1, data, pilots, pil, sync, mapped, ctx=ctx)
    d_in = torch.randn((n_frames, S * nd, 2), dtype=torch.float32, device="cuda:0")
    d_rs = torch.randn((n_frames, T - 1, S, N, 2), dtype=torch.float32, device="cuda:0")
    d_q = torch.randn((N, T * T, 2), dtype=torch.float32, device="cuda:0")
    d_out = torch.empty((n_frames, T, S + 9, N, 2), dtype=torch.float32, device="cuda:0")
    out = {}
    for name, kw in (("dft", {}), ("per-subcarrier steering + radar streams", dict(steer_mode=2, d_Q_sc=d_q, d_radar_streams=d_rs))):
        t = timed(lambda: pre.frames_dev(d_in, 2, 2, nbytes, d_out=d_out, **kw), steps=20, warm=3)
        byts = d_in.numel() * 4 + d_out.numel() * 4 + (d_rs.numel() * 4 if kw else 0)
        out[name] = dict(ms_per_step=t * 1e3, frames_per_s=n_frames / t, GBps=byts / t / 1e9, **roof(byts, t))
    return dict(what="precoder config C: %d packets x 4 TX, 73 symbols x 256 sc, DATA" % n_frames, **{k.replace(" ", "_"): v for k, v in out.items()})


def equalizer_config_c(n_frames=2048, lanes=4, S=64, N=256):
    cp, T = N // 4, 4
    rng = np.random.default_rng(0)
    guard = N // 16
    act = [c for c in range(-N // 2 + guard, N // 2 - guard + 1) if c != 0]
    pilots = [c for c in act if c % 32 == 16][:8]
    data = [c for c in act if c not in pilots]
    ltf = np.zeros(N, np.complex64)
    ltf[np.array(act) + N // 2] = rng.choice([-1.0, 1.0], len(act))
    Pm = synth.hadamard(T)
    mapped = np.stack([(Pm * ltf[sc]).reshape(-1) for sc in range(N)]).astype(np.complex64)
    pil = np.tile(np.array([1, 1, 1, -1, 1, 1, 1, -1], np.complex64)[:len(pilots)], (8, 1))
    sw = np.stack([ltf, ltf, ltf, ltf])
    ctx = jrc_amd.Context(0)
    pre = jrc_amd.mimo_precoder(N, T, 1, data, pilots, pil, sw, mapped, ctx=ctx)
    nd = len(data)
    mcs = 2
    nbytes = (S * nd - 22) // 8
    ns = jrc_amd.n_ofdm_sym(mcs, nd, nbytes)
    assert ns == S, (ns, S)
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    s = pts[rng.integers(0, 4, ns * nd)].astype(np.complex64)
    tx = pre.work(s, mcs, jrc_amd.DATA, nbytes)
    streams = n_frames * lanes
    base = []
    for i in range(8):
        h = (rng.standard_normal(T) + 1j * rng.standard_normal(T)).astype(np.complex64)
        y = np.tensordot(h, tx, axes=(0, 0))
        y = np.concatenate([y[3:4], y[3:]], axis=0)
        y = y + 1e-3 * (rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape))
        base.append(y.astype(np.complex64))
    n_sym = base[0].shape[0]
    x = np.stack([base[i % 8] for i in range(streams)])
    eq = jrc_amd.mimo_ofdm_equalizer(jrc_amd.LS, 24e9, 125e6, N, cp, data, pilots, pil, ltf, mapped, T, n_streams=streams, ctx=ctx)
    d_in = torch.from_numpy(x.view(np.float32).reshape(streams, n_sym, N, 2)).to("cuda:0")
    d_ph = torch.zeros(streams, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    out, n_out, ev = eq.frames_dev(d_in, d_ph, n_sym, ns)
    ctx.sync()
    assert n_out.min().item() == ns and n_out.max().item() == ns
    got = out[0].cpu().numpy().view(np.complex64)[..., 0]
    err = np.abs(got - s.reshape(ns, nd)).max() / np.abs(s).max()
    t = timed(lambda: eq.frames_dev(d_in, d_ph, n_sym, ns))
    byts = streams * (n_sym * N * 8 + ns * nd * 8)
    return dict(what="equalizer config C: %d frames x %d RX lanes, %d symbols x %d sc, LS, DATA" % (n_frames, lanes, n_sym, N),
                ms_per_step=t * 1e3, frames_per_s=n_frames / t, lane_frames_per_s=streams / t, GBps=byts / t / 1e9, **roof(byts, t),
                symbol_error_vs_tx=float(err))


if __name__ == "__main__":
    # the two probes that run as child processes go first: once this process holds a GPU context they would time-slice with it
    only = os.environ.get("JRC_BENCH_EXTRA_ONLY")
    LEGS = {"detectB": lambda: detect_only("B"), "detectB_noise": lambda: detect_only("B", noise_only=True), "detectD": lambda: detect_only("D"),
            "detectD_noise": lambda: detect_only("D", noise_only=True), "powerB": lambda: power_map("B"), "powerD": lambda: power_map("D"),
            "equalizer": equalizer_config_c, "precoder": precoder_config_c, "rdD": lambda: range_doppler("D", 64), "rdB": lambda: range_doppler("B", 64),
            "demodB": lambda: radar_with_demod("B", 512), "demodD": lambda: radar_with_demod("D", 256), "comm_rx": comm_rx_chain,
            "simB": lambda: simulated_chain("B", 64), "simD": lambda: simulated_chain("D", 8), "flowgraphB": lambda: device_resident_flowgraph(64),
            "flowgraph_grc": lambda: device_resident_flowgraph_grc(512)}
    GROUPS = {"detect": ["detectB", "detectB_noise", "detectD", "detectD_noise", "powerB", "powerD"], "demod": ["demodB", "demodD"],
              "comm": ["comm_rx", "equalizer", "precoder"]}
    if only:
        for name in only.split(","):
            for leg in GROUPS.get(name, [name]):
                print(json.dumps(LEGS[leg]()))
        sys.exit(0)
    for fn in (sync_front_end, comm_rx_chain, lambda: detect_only("B"), lambda: detect_only("D"), lambda: radar_with_demod("B", 512), lambda: radar_with_demod("D", 256), precoder_config_c, lambda: range_doppler("D", 8), lambda: range_doppler("B", 64),
               lambda: simulated_chain("B", 64), lambda: simulated_chain("D", 8), equalizer_config_c):
        print(json.dumps(fn()))
