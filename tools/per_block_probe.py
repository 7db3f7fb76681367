#!/usr/bin/env python3
"""The literal per-block drop-ins, timed one work() call at a time (north_star: "the new blocks drop into the existing examples/simulation
flowgraphs unchanged"; /root/reference/examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2189-2197).

In the unchanged graph every block is called with HOST buffers of GNU Radio's and the stock fft_vxx blocks between them stay on the CPU, so
each of these calls pays  H2D of its input + kernel(s) + D2H of its output + one stream synchronise.  For each of the seven hot blocks this
prints p50 / p99 of ONE call through the C++ block classes (gr-mimo-ofdm-jrc_amd/host/jrc_blocks.cc, driven through their C harness, tags
attached as the scheduler would) and the bytes the call moves over PCIe, at two operating points:

   grc : the reference flowgraph's own point, 4 TX x 2 RX, fft_len 64, N_pre 5, N_sym 4, interp 8 x 16 (map 512 x 128), 100-byte PDUs
   B   : BASELINE config B / C, 4 x 4, fft_len 256, 64 symbols, interp 8 x 16 (map 2048 x 256)

and the five-block radar branch wired as in the unchanged .grc — GPU mimo_ofdm_radar -> CPU fft_vxx (scipy pocketfft stands in for the
stock block) -> GPU matrix_transpose -> CPU fft_vxx -> GPU range_angle_estimator — as packets/s on one thread, beside the same branch as
the one radar_chain block.  The CPU blocks' own times (the oracle restatement, same shapes) are measured by bench.py's
secondary.per_block_drop_in leg beside these (this tool never touches oracle/).    usage: tools/per_block_probe.py [--calls 300] [--json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))          # hostblocks.py: the ctypes driver of libjrc_blocks.so (no oracle in it)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def crandn(rng, *shape, scale=1.0):
    return (scale * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))).astype(np.complex64)


def pct(v):
    v = np.sort(np.asarray(v)) * 1e6
    return dict(p50_us=float(v[len(v) // 2]), p99_us=float(v[min(len(v) - 1, int(len(v) * 0.99))]), min_us=float(v[0]), calls=len(v))


def time_calls(fn, calls, warm=20):
    for _ in range(warm):
        fn()
    t = []
    for _ in range(calls):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return pct(t)


def tables(point):
    if point == "grc":
        o = np.load(os.path.join(ROOT, "tests", "golden", "ofdm_config_64.npz"))
        return {k: o[k] for k in o.files}
    import radar_sim_device_resident as drm
    return drm.config_b_tables()


def shapes(point):
    """(N, cp, T, R, N_pre, N_sym radar, n data symbols of the packet, Ir, Ia)"""
    return (64, 16, 4, 2, 5, 4, 18, 8, 16) if point == "grc" else (256, 64, 4, 4, 5, 64, 60, 8, 16)


def block_legs(point, calls):
    import hostblocks as hb
    import jrc_amd as jrc
    L = hb.lib()
    rng = np.random.default_rng(7)
    N, cp, T, R, Npre, S, n_data, Ir, Ia = shapes(point)
    o = tables(point)
    P, NR, NA = T * R, N * Ir, T * R * Ia
    n_sync = len(o["l_stf_ltf_64"])
    n_total = n_sync + 1 + T + n_data
    nd = len(o["data_subcarriers"])
    rb, ab = jrc.radar_axes(N, 125e6, Ir, P, Ia)
    out = {}

    # --- mimo_ofdm_radar: T + R ports of n_total items in, P items of N * Ir out ----------------------------------------------------------
    blk = hb.radar(N, T, R, S, Npre, interp=Ir)
    tx = [crandn(rng, n_total, N) for _ in range(T)]
    rx = [crandn(rng, n_total, N) for _ in range(R)]
    H = np.zeros((P, NR), np.complex64)
    pos = [0]

    def radar_call():
        blk.tag(0, pos[0], "packet_len", n_total)
        blk.tag(T, pos[0], "packet_len", n_total)
        assert blk.run(P, tx + rx, [H]) == P
        pos[0] += n_total
    r = time_calls(radar_call, calls)
    # (only the N_sym used symbols of each port go up, and only the P x N estimates come back: the rows are zero-padded on the host side of the link)
    r.update(h2d_bytes=(T + R) * S * N * 8, d2h_bytes=P * N * 8, what="%d ports x %d items x %d carriers in, %d x %d out (zero-padded rows)" % (T + R, n_total, N, P, NR))
    out["mimo_ofdm_radar"] = r

    # --- matrix_transpose: [P][NR] -> [NR][NA] ------------------------------------------------------------------------------------------------
    tr = hb.transpose(NR, P, Ia)
    xin = crandn(rng, P, NR)
    xout = np.zeros((NR, NA), np.complex64)
    tpos = [0]

    def tr_call():
        tr.tag(0, tpos[0], "packet_len", P)
        assert tr.run(NR, [xin], [xout]) == NR
        tpos[0] += P
    r = time_calls(tr_call, calls)
    r.update(h2d_bytes=P * NR * 8, d2h_bytes=NR * NA * 8, what="%d x %d in, %d x %d out (15/16 zeros)" % (P, NR, NR, NA))
    out["matrix_transpose"] = r

    # --- range_angle_estimator: the [NR][NA] map in, a message out ---------------------------------------------------------------------------
    est = hb.estimator(NA, rb, ab, 2.4, 28.96 if P == 8 else 14.36, 15.0, 0.0)
    m = crandn(rng, NR, NA, scale=0.05)
    m[NR // 8, NA // 2 + 9] += 3.0
    epos = [0]

    def est_call():
        est.tag(0, epos[0], "packet_len", NR)
        assert est.run(0, [m], []) == 0
        epos[0] += NR
    r = time_calls(est_call, calls)
    r.update(h2d_bytes=NR * NA * 8, d2h_bytes=48, what="%d x %d map in, one record out" % (NR, NA))
    out["range_angle_estimator"] = r

    # --- ofdm_cyclic_prefix_remover: one RX stream of a packet ---------------------------------------------------------------------------------
    cpr = hb.cp_remover(N, cp)
    s = crandn(rng, n_total * (N + cp))
    so = np.zeros((n_total, N), np.complex64)
    cpos = [0]

    def cp_call():
        cpr.tag(0, cpos[0], "packet_len", s.size)
        assert cpr.run(n_total, [s], [so]) == n_total
        cpos[0] += s.size
    r = time_calls(cp_call, calls)
    r.update(h2d_bytes=s.size * 8, d2h_bytes=so.size * 8, what="%d symbols of %d + %d samples" % (n_total, N, cp))
    out["ofdm_cyclic_prefix_remover"] = r

    # --- fft_peak_detect: the USRP alignment flowgraph's 40000-bin spectrum (examples/usrp/mimo_usrp_alignment_4tx2rx.grc:720ff) -------------
    n = 40000
    spec = crandn(rng, n, scale=0.001)
    spec[3100] = 2 * np.exp(-0.7j)
    pd = hb.peak_detect(125000000, 8.0, -20.0, 10)
    f3 = [np.zeros(1, np.float32) for _ in range(3)]
    ppos = [0]

    def pd_call():
        pd.tag(0, ppos[0], "packet_len", n)
        assert pd.run(1, [spec], f3) == 1
        ppos[0] += n
    r = time_calls(pd_call, calls)
    r.update(h2d_bytes=n * 8, d2h_bytes=16, what="%d-bin spectrum in, 3 floats out" % n)
    out["fft_peak_detect"] = r

    # --- mimo_precoder: one PDU's symbols in, T ports of n_total items out ---------------------------------------------------------------------
    dc = np.ascontiguousarray(o["data_subcarriers"], np.int32)
    pc = np.ascontiguousarray(o["pilot_subcarriers"], np.int32)
    cf = lambda a: np.ascontiguousarray(a, np.complex64)
    ps, sw, ml, ltf = cf(o["pilot_symbols"]), cf(o["l_stf_ltf_64"]), cf(o["ltf_mapped_sc__ss_sym"]), cf(o["ltf_64"])
    fp, ip = hb._fp, hb._ip
    pre = hb.Block(L.jrcb_make_precoder(N, T, dc.ctypes.data_as(ip), len(dc), pc.ctypes.data_as(ip), len(pc), ps.view(np.float32).ctypes.data_as(fp), ps.shape[0],
                                        sw.view(np.float32).ctypes.data_as(fp), sw.shape[0], ml.view(np.float32).ctypes.data_as(fp), b"", 0, b"", 0, 0, 0))
    mcs = 2
    pdu_len = (n_data * nd - 22) // 8                        # the PSDU that fills n_data QPSK 1/2 symbols (16 service + 6 tail bits, lib/utils.cc:31)
    assert jrc.n_ofdm_sym(mcs, nd, pdu_len) == n_data, (jrc.n_ofdm_sym(mcs, nd, pdu_len), n_data)
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    sym = pts[rng.integers(0, 4, n_data * nd)].astype(np.complex64)
    outs = [np.zeros((n_total, N), np.complex64) for _ in range(T)]
    qpos = [0]

    def pre_call():
        for key, val in (("packet_len", sym.size), ("mcs", mcs), ("packet_type", 2), ("pdu_len", pdu_len)):
            pre.tag(0, qpos[0], key, val)
        assert pre.run(n_total, [sym], outs) == n_total
        qpos[0] += sym.size
    r = time_calls(pre_call, calls)
    r.update(h2d_bytes=sym.size * 8, d2h_bytes=T * n_total * N * 8, what="%d data symbols x %d carriers in, %d ports x %d items out" % (n_data, nd, T, n_total))
    out["mimo_precoder"] = r

    # --- mimo_ofdm_equalizer: the frame frame_sync hands over ([LTF, LTF, SIG, MIMO-LTFs, data]) through a flat channel -----------------------------
    txf = np.stack(outs)                                      # [T][n_total][N]
    h = crandn(rng, T)
    y = np.tensordot(h, txf, axes=(0, 0))
    y = np.ascontiguousarray(np.concatenate([y[n_sync - 1:n_sync], y[n_sync - 1:]]), np.complex64)     # frame_sync drops the STFs: [LTF, LTF, SIG, MIMO-LTFs, data]
    eq = hb.Block(L.jrcb_make_equalizer(0, 24e9, 125e6, N, cp, dc.ctypes.data_as(ip), len(dc), pc.ctypes.data_as(ip), len(pc), ps.view(np.float32).ctypes.data_as(fp),
                                        ps.shape[0], ltf.view(np.float32).ctypes.data_as(fp), ml.view(np.float32).ctypes.data_as(fp), ml.shape[1], T, b""))
    eo = np.zeros((len(y), nd), np.complex64)
    ypos = [0]
    got = []

    def eq_call():
        eq.tag(0, ypos[0], "frame_start", 0.0)
        got.append(eq.run(len(y), [y], [eo]))
        ypos[0] += eq.consumed(0)
    r = time_calls(eq_call, calls)
    assert got[-1] == n_data, (got[-1], n_data)
    err = float(np.abs(eo[:n_data].reshape(-1) - sym).max())
    assert err < 1e-3, err                                     # the block equalised the frame it was timed on
    r.update(h2d_bytes=y.size * 8, d2h_bytes=n_data * nd * 8, what="%d symbols x %d carriers in, %d x %d equalised cells out (LS, DATA, QPSK 1/2)" % (len(y), N, n_data, nd))
    out["mimo_ofdm_equalizer"] = r
    return out


def radar_branch(point, seconds):
    """the radar branch of the unchanged .grc on one thread: GPU radar -> CPU FFT -> GPU transpose -> CPU FFT -> GPU estimator; and the radar_chain block"""
    import hostblocks as hb
    import jrc_amd as jrc
    import scipy.fft as sfft
    rng = np.random.default_rng(9)
    N, cp, T, R, Npre, S, n_data, Ir, Ia = shapes(point)
    P, NR, NA = T * R, N * Ir, T * R * Ia
    n_total = 4 + 1 + T + n_data
    rb, ab = jrc.radar_axes(N, 125e6, Ir, P, Ia)
    nda = 28.96 if P == 8 else 14.36
    from jrc_amd import synth
    sc = synth.Scenario(N, T, R, S, targets=[(10.0, 20.0, 0.0, 100.0)])
    fr = synth.make_frames(sc, 8)                               # [F][T+R][Npre+S][N]
    pad = np.zeros((n_total - Npre - S, N), np.complex64)
    ports = [[np.ascontiguousarray(np.concatenate([fr[f, p], pad])) for p in range(T + R)] for f in range(8)]
    radar, tr, est = hb.radar(N, T, R, S, Npre, interp=Ir), hb.transpose(NR, P, Ia), hb.estimator(NA, rb, ab, 2.4, nda, 15.0, 0.0)
    H, m = np.zeros((P, NR), np.complex64), np.zeros((NR, NA), np.complex64)
    stage = dict(radar=0.0, fft_range_cpu=0.0, transpose=0.0, fft_angle_cpu=0.0, estimator=0.0)
    k, t_start = 0, time.perf_counter()
    pr, pt, pe = 0, 0, 0
    while True:
        t0 = time.perf_counter()
        radar.tag(0, pr, "packet_len", n_total)
        radar.tag(T, pr, "packet_len", n_total)
        assert radar.run(P, ports[k % 8], [H]) == P
        pr += n_total
        t1 = time.perf_counter()
        rp = (sfft.ifft(H, axis=1) * np.float32(NR)).astype(np.complex64)                  # fft_vxx reverse, unnormalised (…radar_sim.grc:940-962)
        t2 = time.perf_counter()
        tr.tag(0, pt, "packet_len", P)
        assert tr.run(NR, [rp], [m]) == NR
        pt += P
        t3 = time.perf_counter()
        mp = np.ascontiguousarray(sfft.fftshift(sfft.fft(m, axis=1), axes=1), np.complex64)   # fft_vxx forward + shift (:963-985)
        t4 = time.perf_counter()
        est.tag(0, pe, "packet_len", NR)
        assert est.run(0, [mp], []) == 0
        pe += NR
        t5 = time.perf_counter()
        for name, d in zip(stage, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            stage[name] += d
        k += 1
        if k >= 20 and time.perf_counter() - t_start > seconds:
            break
    el = time.perf_counter() - t_start
    pub = est.state()["published"]
    res = dict(packets_per_s_one_thread=k / el, us_per_packet_by_stage={n: 1e6 * v / k for n, v in stage.items()},
               packets_per_s_pipeline_ideal=k / max(stage.values()), packets=k, messages_published=len(pub),
               what="GPU mimo_ofdm_radar -> CPU fft (scipy pocketfft) -> GPU matrix_transpose -> CPU fft + shift -> GPU range_angle_estimator, "
                    "one work() call per block per packet, host buffers between all of them; pipeline-ideal = 1 / slowest stage (a thread per block)")
    if pub:
        got = {kk: vv[0] for kk, vv in pub[-1]["msg"]}
        res["last_message"] = got
    # the same branch as the ONE radar_chain block, one packet per turn (the latency regime), then 64 per turn
    chain = hb.radar_chain(N, T, R, S, Npre, Ir, Ia, rb, ab, 2.4, nda, 15.0, 0.0, frames_per_batch=16, batches_in_flight=3)
    pc_, t0, k1 = 0, time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds / 2:
        chain.tag(0, pc_, "packet_len", n_total)
        chain.tag(T, pc_, "packet_len", n_total)
        chain.run(0, ports[k1 % 8], [])
        chain.set("flush", 0)
        pc_ += n_total
        k1 += 1
    res["radar_chain_block_one_packet_per_turn_flushed_packets_per_s"] = k1 / (time.perf_counter() - t0)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--points", default="grc,B")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    out = {}
    for point in a.points.split(","):
        out[point] = dict(shape=dict(zip(("fft_len", "cp", "N_tx", "N_rx", "N_pre", "N_sym", "n_data_symbols", "interp_range", "interp_angle"), shapes(point))),
                          blocks=block_legs(point, a.calls), radar_branch_unchanged_grc=radar_branch(point, a.seconds))
    if a.json:
        print(json.dumps(out))
        return
    for point, r in out.items():
        print("== operating point %s: %s" % (point, r["shape"]))
        for name, b in r["blocks"].items():
            print("  %-28s p50 %8.1f us  p99 %8.1f us   H2D %9d B  D2H %9d B   %s" % (name, b["p50_us"], b["p99_us"], b["h2d_bytes"], b["d2h_bytes"], b["what"]))
        rb = r["radar_branch_unchanged_grc"]
        print("  radar branch (unchanged .grc wiring): %.0f packets/s on one thread (pipeline-ideal %.0f); per stage us: %s; radar_chain block, 1 packet per turn: %.0f packets/s"
              % (rb["packets_per_s_one_thread"], rb["packets_per_s_pipeline_ideal"], {k: round(v, 1) for k, v in rb["us_per_packet_by_stage"].items()},
                 rb["radar_chain_block_one_packet_per_turn_flushed_packets_per_s"]))


if __name__ == "__main__":
    main()
