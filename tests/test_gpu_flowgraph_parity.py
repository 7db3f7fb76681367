"""GPU tier, the acceptance scenario of BASELINE.json's north_star: the reference's simulation flowgraphs
(examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2165-2232, examples/simulation/communication/mimo_ofdm_jrc_comm_sim.grc)
wired twice by the SAME wiring code (examples/radar_sim_flowgraph.py, examples/comm_sim_flowgraph.py) — once over the HIP blocks,
once over the CPU oracle's blocks (tests/oracle_blocks.py) — fed identical symbols and identical draws of the random sources
(zero_pad noise, noise sources), and compared edge by edge:

  * chained: each graph runs on its own upstream results, so an edge's error is everything accumulated from the PDU to it;
    complex-float edges  ||a-b||_inf / ||b||_inf <= 1e-4 (north_star's tolerance), integer edges bit-exact (lengths announced by the
    tagged-stream blocks, items consumed, SIG fields, tag offsets, estimator indices, decoded bytes, CRC flag);
  * block by block: the oracle graph re-run with every block reading the HIP graph's tensor on its input edge (`force=`), which
    isolates each block's own error (the number docs/history.md §5.2 tabulates).

Shapes: the reference's operating point (4x2, N=64, N_pre=5, N_sym=4; tests/golden/radar_flowgraph_point.npz), BASELINE config A
(1x1, 64 subcarriers, 16 symbols) and a config-B scale-up (4x4, 256 subcarriers, 64 symbols).  The per-edge errors of the run are
written to gpurun_out/flowgraph_parity.json."""
import json
import os
import sys

import numpy as np
import pytest

import oracle
import oracle_blocks
from conftest import GOLDEN, ROOT, rel_err
from metric_truth import assert_metric_parity

sys.path.insert(0, os.path.join(ROOT, "examples"))
pytestmark = pytest.mark.gpu

TOL = 1e-4                                     # north_star: 1e-4 relative complex-float tolerance
REPORT = {}


def teardown_module(module):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "flowgraph_parity.json"), "w") as fh:
            json.dump(REPORT, fh, indent=1, sort_keys=True)
    except OSError:
        pass


def tables_64(ofdm64, T):
    """the reference's 64-carrier tables; for T != 4 the MIMO-LTF mapping P_ltf shrinks to its leading T x T block"""
    o = {k: ofdm64[k] for k in ofdm64.files}
    if T != 4:
        P = np.asarray(ofdm64["P_ltf"])[:T, :T]
        o["ltf_mapped_sc__ss_sym"] = np.stack([(P * ofdm64["ltf_64"][sc]).reshape(-1) for sc in range(64)]).astype(np.complex64)
        o["N_tx"] = np.int32(T)
    return o


def tables_n(N, T=4):
    """config_c_tables for another carrier count, radar graph only (no STF needed)"""
    from test_gpu_comm import config_c_tables
    data, pilots, pil, ltf, mapped, sync = config_c_tables(N, T)
    return dict(N_tx=np.int32(T), data_subcarriers=np.array(data, np.int32), pilot_subcarriers=np.array(pilots, np.int32), pilot_symbols=pil,
                l_stf_ltf_64=sync.astype(np.complex64), ltf_64=ltf, ltf_mapped_sc__ss_sym=mapped)


def tables_256(T=4):
    """config_c_tables (test_gpu_comm.py) plus the 802.11-style sync words the sync front end needs, scaled like the 64-carrier
    ones: STF on every 4th carrier (period N/4 in time, what the detector's delay-N/4 autocorrelation looks for), then the LTF
    advanced by N/4 samples ((-j)^i phase ramp, as l_stf_ltf_64[2] is of l_stf_ltf_64[3]) and the LTF itself"""
    from test_gpu_comm import config_c_tables
    N = 256
    data, pilots, pil, ltf, mapped, _ = config_c_tables(N, T)
    rng = np.random.default_rng(77)
    act = np.flatnonzero(ltf)
    stf = np.zeros(N, np.complex64)
    idx = np.array([i for i in act if i % 4 == 0])
    stf[idx] = rng.choice([-1.0, 1.0], idx.size) * (1 + 1j) * np.sqrt(act.size / (2.0 * idx.size))
    ltf_rot = (ltf * (-1j) ** np.arange(N)).astype(np.complex64)
    sync = np.stack([stf, stf, ltf_rot, ltf]).astype(np.complex64)
    l_ltf_time = N * np.fft.ifft(np.fft.fftshift(ltf)) / np.sqrt(np.count_nonzero(ltf))      # the ofdm_config module's l_ltf_fir rule
    return dict(N_tx=np.int32(T), data_subcarriers=np.array(data, np.int32), pilot_subcarriers=np.array(pilots, np.int32),
                pilot_symbols=pil, l_stf_ltf_64=sync, ltf_64=ltf, ltf_mapped_sc__ss_sym=mapped,
                l_ltf_fir=np.conj(l_ltf_time)[::-1].astype(np.complex64))


def qpsk(rng, n):
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    return pts[rng.integers(0, 4, n)].astype(np.complex64)


RADAR_CF32_EDGES = ("tx_f", "tx_t", "bursts", "sims", "rx_t", "rx_f", "H", "range_profile", "transposed", "map")
RESULT_INTS = ("peak_range_idx", "peak_angle_idx", "angle_null_idx", "discard_range_idx", "discard_angle_idx", "n_noise_samples", "published")
RESULT_FLOATS = ("peak_power", "noise_power", "snr_est", "range_val", "angle_val")


def compare_results(g, o, exact_floats):
    for k in RESULT_INTS:
        assert getattr(g, k) == getattr(o, k), k
    for k in RESULT_FLOATS:
        a, b = getattr(g, k), getattr(o, k)
        if exact_floats or k in ("range_val", "angle_val"):      # bin values are table look-ups: exact once the indices agree
            assert a == b, (k, a, b)
        else:
            assert abs(a - b) <= TOL * abs(b), (k, a, b)


RADAR_SHAPES = {
    # name: (tables, fft_len, N_rx, n data symbols, radar N_sym, targets (range, velocity, rcs dBsm, angle))
    "operating_point_4x2_N64": (lambda o: tables_64(o, 4), 64, 2, None, None, ([10.0], [0.0], [20.0], [0.0])),
    "operating_point_moving_target": (lambda o: tables_64(o, 4), 64, 2, None, None, ([17.0], [12.0], [20.0], [-30.0])),
    "config_A_1x1_N64_S16": (lambda o: tables_64(o, 1), 64, 1, 15, 16, ([10.0], [0.0], [20.0], [0.0])),
    "config_B_4x4_N256_S64": (lambda o: tables_256(4), 256, 4, 60, 64, ([10.0], [0.0], [20.0], [20.0])),
    "two_targets_2x2_N128": (None, 128, 2, 10, 8, ([12.0, 30.0], [0.0, -8.0], [20.0, 23.0], [25.0, -40.0])),
    # BASELINE config D: 4x4, 1024 subcarriers, radar window 128 symbols (4 MIMO-LTFs + 124 data symbols = the 133 symbols of config D's packets), 8 targets
    "config_D_4x4_N1024_S128_8_targets": (lambda o: tables_n(1024, 4), 1024, 4, 124, 128,
                                          ([9.0, 21.0, 37.0, 55.0, 80.0, 140.0, 260.0, 410.0], [0.0, 12.0, -30.0, 5.0, -8.0, 40.0, -22.0, 3.0],
                                           [20.0, 14.0, 17.0, 11.0, 19.0, 13.0, 16.0, 12.0], [20.0, -35.0, 5.0, 48.0, -12.0, -55.0, 30.0, -3.0])),
}
HEAVY = {"config_D_4x4_N1024_S128_8_targets"}        # one variant (fused demod, DATA) of the shapes whose oracle graph takes a minute


def tables_128(T=2):
    from test_gpu_comm import config_c_tables
    from jrc_amd import synth
    N = 128
    data, pilots, pil, ltf, mapped, sync = config_c_tables(N, 4, seed=3)
    mapped = np.stack([(synth.hadamard(T) * ltf[sc]).reshape(-1) for sc in range(N)]).astype(np.complex64)
    return dict(N_tx=np.int32(T), data_subcarriers=np.array(data, np.int32), pilot_subcarriers=np.array(pilots, np.int32),
                pilot_symbols=pil[:, :len(pilots)], l_stf_ltf_64=sync.astype(np.complex64), ltf_64=ltf, ltf_mapped_sc__ss_sym=mapped)


@pytest.mark.parametrize("fused_demod", [True, False], ids=["fused_demod", "block_demod"])
@pytest.mark.parametrize("shape", list(RADAR_SHAPES))
def test_radar_flowgraph_edge_by_edge(jrc, ctx, ofdm64, shape, fused_demod):
    import radar_sim_flowgraph as fgm
    if shape in HEAVY and not fused_demod:
        pytest.skip("heavy shape: the fused-demod variant only")
    mk, N, R, n_data, S_radar, (rng_m, vel, rcs, az) = RADAR_SHAPES[shape]
    o = tables_128(2) if mk is None else mk(ofdm64)
    T = int(o["N_tx"])
    kw = dict(trgt_range=rng_m, trgt_velocity=vel, trgt_rcs_dbsm=rcs, trgt_angle=az, N_rx=R, fft_len=N, N_sym_radar=S_radar, seed=5)
    if shape.startswith("operating_point"):                      # the .grc's values, as minted from it (tests/golden)
        fgp = np.load(os.path.join(GOLDEN, "radar_flowgraph_point.npz"))
        assert [N, T, R] == [int(fgp["radar_ints"][0]), int(fgp["radar_ints"][1]), int(fgp["radar_ints"][2])]
    hip = fgm.RadarSimFlowgraph(o, ctx=ctx, fused_demod=fused_demod, **kw)
    orc = fgm.RadarSimFlowgraph(o, blocks=oracle_blocks, fused_demod=fused_demod, **kw)
    if shape.startswith("operating_point"):
        assert (hip.N_pre, hip.N_sym_radar, hip.Ir, hip.Ia) == (int(fgp["radar_ints"][4]), int(fgp["radar_ints"][3]), 8, 16)
        assert np.array_equal(hip.range_bins, fgp["estimator_range_bins_f32"]) and np.array_equal(hip.angle_bins, fgp["estimator_angle_bins_f32"])
        assert hip.pad_tail == int(fgp["zero_pad_tail"]) and abs(hip.noise_var - float(fgp["var_noise_var"])) < 1e-18
        assert np.allclose(np.array(hip.TX_RXs), np.stack([fgp["var_TX%d_RXs" % (t + 1)] for t in range(4)]), rtol=1e-12)
    rng = np.random.default_rng(11)
    nd = len(o["data_subcarriers"])
    mcs = 2
    if n_data is None:
        nbytes = 100
        n_data = jrc.n_ofdm_sym(mcs, nd, nbytes)
    else:
        nbytes = (n_data * nd - 22) // 8
        assert jrc.n_ofdm_sym(mcs, nd, nbytes) == n_data
    rep = REPORT.setdefault("radar/%s/%s" % (shape, "fused" if fused_demod else "blocks"), {})
    for ptype in ((fgm.DATA,) if shape in HEAVY else (fgm.DATA, fgm.NDP)):
        sym = qpsk(rng, n_data * nd)
        gres, ge = hip.run_packet(sym, mcs, ptype, nbytes)
        src = dict(pads=ge["pads"], noise=ge["noise"])
        for t in range(T):                                       # zero_pad passes the burst through untouched
            assert np.array_equal(ge["bursts"][t][:ge["tx_t"].shape[1]], ge["tx_t"][t])
            assert 0.007 < ge["pads"][t].real.std() < 0.014 and 0.007 < ge["pads"][t].imag.std() < 0.014   # normal_distribution(0, 1e-2): sigma 0.01
        # (1) chained: the oracle graph on its own upstream results
        ores, oe = orc.run_packet(sym, mcs, ptype, nbytes, sources=src)
        assert ge["lengths"] == oe["lengths"]                     # every announced packet length / consumed count, bit-exact
        P = T * R
        assert ge["lengths"]["radar_out"] == P and ge["lengths"]["transpose_out"] == N * 8 and ge["lengths"]["map_rows"] == N * 8
        assert ge["map"].shape == (N * 8, P * 16)
        n_total = hip.n_sync + 1 + T + n_data
        assert ge["lengths"]["precoder_out"] == n_total and ge["lengths"]["cp_remover_out"] == n_total + 3
        for k in RADAR_CF32_EDGES:
            assert ge[k].shape == oe[k].shape and ge[k].dtype == np.complex64, k
            err = rel_err(ge[k], oe[k])
            rep["chained:" + k] = max(rep.get("chained:" + k, 0.0), err)
            assert err <= TOL, (k, err)
        compare_results(gres, ores, exact_floats=False)
        # (2) block by block: every oracle block on the HIP graph's input edge
        bres, be = orc.run_packet(sym, mcs, ptype, nbytes, sources=src, force={k: ge[k] for k in RADAR_CF32_EDGES})
        for k in RADAR_CF32_EDGES:
            err = rel_err(ge[k], be[k])
            rep["block:" + k] = max(rep.get("block:" + k, 0.0), err)
            assert err <= 2e-5, (k, err)
        assert np.array_equal(ge["bursts"], be["bursts"]) and np.array_equal(ge["transposed"], be["transposed"])   # copies: bit-equal
        assert np.array_equal(ge["H"], be["H"])                   # A1: the reference's own evaluation order, bit-exact
        compare_results(gres, bres, exact_floats=True)            # A5 on the same map: every field exact
        if ptype == fgm.DATA and "moving" not in shape and len(rng_m) == 1 and T > 1:
            assert gres.published == 1 and abs(gres.range_val - rng_m[0]) < 0.8 and abs(gres.angle_val - az[0]) < 2.5


COMM_CF32_EDGES = ("symbols", "tx_f", "tx_t", "padded", "rx", "detector_out", "sync_out", "y", "eq_out")


def same_events(ge, oe, tol, freq_floor=1.0):
    assert len(ge) == len(oe)
    for g, o in zip(ge, oe):
        assert g["kind"] == o["kind"] and g["offset"] == o["offset"]
        if g["kind"] == 1:
            assert (g["data_bytes"], g["mcs"], g["packet_type"]) == (o["data_bytes"], o["mcs"], o["packet_type"])
            assert abs(g["snr"] - o["snr"]) <= max(tol * abs(o["snr"]), 1e-3) and abs(g["freq_offset"] - o["freq_offset"]) <= tol * max(freq_floor, abs(o["freq_offset"]))
        else:
            assert abs(g["snr_data"] - o["snr_data"]) <= max(tol * abs(o["snr_data"]), 1e-3)
            assert g["chan_mean"].shape == o["chan_mean"].shape
            if g["chan_mean"].size:
                assert rel_err(g["chan_mean"], o["chan_mean"]) <= tol


def compare_comm_edges(ge, oe, rep, tag, tol, fft_len=64, freq_floor=1.0, payload_equal=True):
    assert ge["encoder_tags"] == oe["encoder_tags"]               # packet_len, packet_type, mcs, pdu_len
    for k in COMM_CF32_EDGES:
        assert k in ge and k in oe, k
        assert ge[k].shape == oe[k].shape, (k, ge[k].shape, oe[k].shape)
        err = rel_err(ge[k], oe[k]) if ge[k].size else 0.0
        rep[tag + k] = max(rep.get(tag + k, 0.0), err)
        assert err <= tol, (k, err)
    assert np.array_equal(ge["symbols"], oe["symbols"])           # constellation points: exact
    for i, name in enumerate(("metric_delayed", "metric_corr", "metric_norm")):
        a, b = ge["metrics"][i], oe["metrics"][i]
        if name == "metric_norm":
            # float edge |corr| / power.  The stock blocks_moving_average_ff keeps a RUNNING sum whose round-off depends on everything that went
            # through it since the scheduler call began (worst behind the 40 dB step from burst to pad noise); the oracle restates the worst
            # case (one sum over the capture), the device adds each window afresh (sync.hip).  So the edge is held against its definition in
            # float64 (tests/metric_truth.py): device within north_star's 1e-4 of it and no further from it than the oracle's running sum;
            # device against oracle within 1e-4 plus the running sum's own measured distance from the definition.  DESIGN.md §4.
            window = fft_len // 2                                  # corr_window_size of the flowgraph (examples/comm_sim_flowgraph.py)
            pw = int(1.5 * window)
            x_ora = ge["rx"] if tag == "block:" else oe["rx"]      # block by block the oracle's metric blocks read the HIP graph's samples
            e = assert_metric_parity(a, b, ge["rx"], x_ora, fft_len // 4, window, pw, 1 / 1.5, live=pw, tol=TOL)
            rep[tag + name + "_vs_float64_definition"] = max(rep.get(tag + name + "_vs_float64_definition", 0.0), e["dev"])
            rep[tag + name + "_oracle_running_sum_vs_definition"] = max(rep.get(tag + name + "_oracle_running_sum_vs_definition", 0.0), e["ora"])
            rep[tag + name] = max(rep.get(tag + name, 0.0), e["dev_vs_ora"])
            continue
        err = rel_err(a, b)
        rep[tag + name] = max(rep.get(tag + name, 0.0), err)
        assert err <= tol, (name, err)
    for key in ("detector_tags", "sync_tags"):                    # tag offsets bit-exact, the CFO values they carry to tolerance
        assert [t[0] for t in ge[key]] == [t[0] for t in oe[key]], key
        for a, b in zip(ge[key], oe[key]):
            assert abs(a[1] - b[1]) <= tol * max(1.0, abs(b[1])), key
    assert ge["eq_consumed"] == oe["eq_consumed"]
    same_events(ge["eq_events"], oe["eq_events"], tol, freq_floor)
    if oe["chan_est"] is None:
        assert ge["chan_est"] is None
    else:
        err = rel_err(ge["chan_est"], oe["chan_est"])
        rep[tag + "chan_est"] = max(rep.get(tag + "chan_est", 0.0), err)
        assert err <= tol
    assert ge.get("crc_ok") == oe.get("crc_ok") and (not payload_equal or ge.get("payload") == oe.get("payload"))       # decoded PDU bytes + CRC flag


@pytest.mark.parametrize("est", [0, 1], ids=["LS", "STA"])
@pytest.mark.parametrize("mcs", [0, 2, 3, 5])
def test_comm_flowgraph_edge_by_edge_at_the_grc_operating_point(jrc, ctx, ofdm64, mcs, est):
    """4 TX, N=64, line-of-sight channel of the .grc (distance 20 m, theta 20 deg, tx_multiplier 0.5, NF 10 dB): NDP (channel
    sounding) then DATA packets without and with the steering derived from the sounding (per subcarrier = chan_est_smoothing
    False as in the .grc, and from the mean channel)."""
    import comm_sim_flowgraph as cfm
    fgp = np.load(os.path.join(GOLDEN, "radar_flowgraph_point.npz"))
    rep = REPORT.setdefault("comm/N64/mcs%d/%s" % (mcs, "STA" if est else "LS"), {})
    for smoothing in (False, True):
        hip = cfm.CommSimFlowgraph(ofdm64, mcs=mcs, estimator=est, seed=2, ctx=ctx, channel="los", smoothing=smoothing)
        orc = cfm.CommSimFlowgraph(ofdm64, mcs=mcs, estimator=est, seed=2, blocks=oracle_blocks, channel="los", smoothing=smoothing)
        orcb = cfm.CommSimFlowgraph(ofdm64, mcs=mcs, estimator=est, seed=2, blocks=oracle_blocks, channel="los", smoothing=smoothing)   # block by block
        assert [hip.pad_front, hip.pad_tail] == list(fgp["comm_zero_pad"]) and hip.sync_length == int(fgp["comm_frame_sync_ints"][2])
        assert hip.ignore_gap == int(fgp["comm_frame_detector"][4]) and hip.corr_window_size == int(fgp["comm_moving_avg"][0])
        assert abs(1 / abs(hip.h[0]) - float(fgp["comm_noise_var_path_loss"][1])) < 1e-5 * float(fgp["comm_noise_var_path_loss"][1])
        assert abs(hip.noise_var - float(fgp["comm_noise_var_path_loss"][0])) < 1e-18
        rng = np.random.default_rng(100 + mcs)
        pdus = [(bytes([1]) + b"sounding packet", False)]
        pdus += [(bytes([2]) + rng.integers(0, 256, 80 + 57 * i, dtype=np.uint8).tobytes(), i >= 1) for i in range(3)]
        for pdu, steer in pdus:
            gok, gpay, ginfo = hip.send(pdu, steer=steer)
            ge = ginfo["edges"]
            src = dict(pads=ge["pads"], noise=ge["noise"])
            T = hip.T
            for t in range(T):                                    # zero_pad: the burst untouched between the pads
                assert np.array_equal(ge["padded"][t][hip.pad_front:hip.pad_front + ge["tx_t"].shape[1]], ge["tx_t"][t])
            ook, opay, oinfo = orc.send(pdu, steer=steer, sources=src)
            oe = oinfo["edges"]
            assert (gok, gpay) == (ook, opay)
            if steer:
                err = rel_err(ge["steering"], oe["steering"])
                rep["chained:steering"] = max(rep.get("chained:steering", 0.0), err)
                assert err <= TOL
            compare_comm_edges(ge, oe, rep, "chained:", TOL)
            if pdu[0] == 2 and mcs < 5:
                assert gok and gpay == pdu                        # and the link works: the PDU comes back (16-QAM 3/4 may fail at this SNR)
            # block by block: a second oracle graph in lockstep (the encoder's scrambler seed and the decoder's PER window are state),
            # every block reading the HIP graph's input edge; the sounding it steers with is the HIP graph's as well
            keys = [k for k in COMM_CF32_EDGES + ("metrics", "detector_tags", "sync_tags", "eq_events", "chan_est") if ge.get(k) is not None]
            if steer:
                orcb.chan_est = hip.chan_est
            bok, bpay, binfo = orcb.send(pdu, steer=steer, sources=src, force={k: ge[k] for k in keys})
            compare_comm_edges(ge, binfo["edges"], rep, "block:", 2e-5)
            assert (bok, bpay) == (gok, gpay)


def test_comm_flowgraph_edge_by_edge_256_subcarriers(jrc, ctx):
    """the same graph scaled to BASELINE config C's carrier count (4 TX, N=256, cp 64) on the flat drawn channel"""
    import comm_sim_flowgraph as cfm
    o = tables_256(4)
    rep = REPORT.setdefault("comm/N256/mcs2/LS", {})
    hip = cfm.CommSimFlowgraph(o, mcs=2, estimator=0, seed=4, ctx=ctx, fft_len=256, channel="flat")
    orc = cfm.CommSimFlowgraph(o, mcs=2, estimator=0, seed=4, blocks=oracle_blocks, fft_len=256, channel="flat")
    assert np.array_equal(hip.h, orc.h)
    rng = np.random.default_rng(9)
    for pdu, steer in [(bytes([1]) + b"sounding", False), (bytes([2]) + rng.integers(0, 256, 900, dtype=np.uint8).tobytes(), False),
                       (bytes([2]) + rng.integers(0, 256, 1400, dtype=np.uint8).tobytes(), True)]:
        gok, gpay, ginfo = hip.send(pdu, snr_db=30.0, steer=steer, lead=2560)
        ge = ginfo["edges"]
        ook, opay, oinfo = orc.send(pdu, snr_db=30.0, steer=steer, lead=2560, sources=dict(pads=None, noise=ge["noise"]))
        assert "y" in ge, ginfo.keys()
        # (behind the burst, 30 dB over the noise that follows, the oracle's running power sum keeps the burst's round-off — 2e-3 of the metric
        # there; the device's metric is within 1e-4 of the float64 definition throughout: compare_comm_edges)
        compare_comm_edges(ge, oinfo["edges"], rep, "chained:", TOL, fft_len=256)
        assert (gok, gpay) == (ook, opay)
        if pdu[0] == 2:
            assert gok and gpay == pdu


def test_radar_receive_graph_through_the_host_blocks_tags_and_consumption(jrc, ctx, ofdm64):
    """the receive side of the radar graph once more through the C++ block classes (gr-mimo-ofdm-jrc_amd/host: the reference's make() /
    work() interface), one scheduler turn per block with the stream tags handed from block to block: every length tag, its offset, the
    items each block consumes and the estimator's message are the integers the reference's rules give (cited per assertion), and every
    buffer equals the edge of the Python-wired graph bit for bit."""
    import hostblocks as hb
    import radar_sim_flowgraph as fgm
    o = tables_64(ofdm64, 4)
    N, cp, T, R, Ir, Ia = 64, 16, 4, 2, 8, 16
    P, L = T * R, N * Ir
    fg = fgm.RadarSimFlowgraph(o, [10.0], [0.0], [20.0], [20.0], ctx=ctx, fused_demod=False, seed=8)
    rng = np.random.default_rng(21)
    nbytes = 100
    n_data = jrc.n_ofdm_sym(2, 48, nbytes)
    res, e = fg.run_packet(qpsk(rng, n_data * 48), 2, fgm.DATA, nbytes)
    n_total = e["tx_f"].shape[1]
    n_burst = e["rx_t"].shape[1]
    rx_fft = jrc.fft_vcc(N, True, None, True, ctx=ctx)
    rx_f = []
    for r in range(R):
        blk = hb.cp_remover(N, cp)
        blk.tag(0, 0, "packet_len", n_burst)
        out = np.zeros((n_burst // (N + cp), N), np.complex64)
        assert blk.run(len(out), [e["rx_t"][r]], [out]) == n_total + 3             # lib/ofdm_cyclic_prefix_remover_impl.cc:86
        assert blk.consumed(0) == n_burst                                          # the TSB base consumes the whole packet
        assert {"offset": 0, "key": "packet_len", "value": n_total + 3} in blk.state()["out_tags"][0]
        rx_f.append(rx_fft.work(out))
    assert np.array_equal(np.stack(rx_f), e["rx_f"])
    radar = hb.radar(N, T, R, T, fg.N_pre, interp=Ir)
    radar.tag(0, 0, "packet_len", n_total)                                          # the precoder's length tag on the TX reference ports
    radar.tag(T, 0, "packet_len", n_total + 3)                                      # the cp remover's on the RX ports
    H = np.zeros((P, L), np.complex64)
    assert radar.run(P, [e["tx_f"][t] for t in range(T)] + rx_f, [H]) == P          # lib/mimo_ofdm_radar_impl.cc:303-339
    assert [radar.consumed(p) for p in range(T + R)] == [n_total] * T + [n_total + 3] * R   # :326-334: whole packets
    assert radar.state()["out_tags"][0] == [{"offset": 0, "key": "packet_len", "value": P}]  # :306-309
    assert np.array_equal(H, e["H"])
    prof = jrc.fft_vcc(L, False, None, False, ctx=ctx).work(H)
    tr = hb.transpose(L, P, Ia)
    tr.tag(0, 0, "packet_len", P)
    out = np.zeros((L, P * Ia), np.complex64)
    assert tr.run(L, [prof], [out]) == L and tr.consumed(0) == P                    # lib/matrix_transpose_impl.cc:62-110
    assert tr.state()["out_tags"][0] == [{"offset": 0, "key": "packet_len", "value": L}]
    assert np.array_equal(out, e["transposed"])
    m = jrc.fft_vcc(P * Ia, True, None, True, ctx=ctx).work(out)
    assert np.array_equal(m, e["map"])
    est = hb.estimator(P * Ia, fg.range_bins, fg.angle_bins, 2 * 3e8 / (2 * fg.samp_rate), 2 * float(np.rad2deg(np.arcsin(2 / P))), 15.0, 0.0)
    est.tag(0, 0, "packet_len", L)
    assert est.run(0, [m], []) == 0 and est.consumed(0) == L                        # lib/range_angle_estimator_impl.cc:114-119, :283
    msg = est.state()["published"]
    assert res.published == 1 and len(msg) == 1 and msg[0]["port"] == "params"      # :234-253
    assert {k: v[0] for k, v in msg[0]["msg"]} == {"range": res.range_val, "angle": res.angle_val, "power": res.peak_power, "snr": res.snr_est}


@pytest.mark.parametrize("summed", [True, False], ids=["simulators_summed_on_the_spectrum", "simulators_one_by_one"])
def test_device_resident_flowgraph_equals_the_block_by_block_graph(jrc, ctx, monkeypatch, summed):
    """precoder -> OFDM modulator -> zero_pad -> target simulators -> A6+A7+A1 -> A2..A5 as one frame-batched leg that
    never leaves HBM (examples/radar_sim_device_resident.py: jrc_precoder_frames_dev, jrc_ofdm_mod_dev, jrc_zero_pad_strided_dev,
    jrc_tsim_run_dev, jrc_chain_run_td_dev) at config B's geometry, against the block-by-block graph of examples/radar_sim_flowgraph.py
    fed the same symbols and the pads this leg drew (no noise sources on either side), packet by packet."""
    import radar_sim_device_resident as drm
    import radar_sim_flowgraph as fgm
    o = tables_256(4)
    N, R, n_data, S, F = 256, 4, 60, 64, 5
    tg = dict(trgt_range=[10.0, 31.0], trgt_velocity=[0.0, 6.0], trgt_rcs_dbsm=[20.0, 24.0], trgt_angle=[20.0, -35.0])
    monkeypatch.setenv("JRC_DRF_SUM_ON_SPECTRUM", "1" if summed else "0")        # jrc_tsim_run_sum_dev, or jrc_tsim_run_dev per TX port with accumulate_out
    sim = drm.DeviceResidentRadarSim(o, N, R, n_data, S, F, seed=40, ctx=ctx, **tg)
    assert sim.sum_on_spectrum == summed
    rng = np.random.default_rng(77)
    nd = len(o["data_subcarriers"])
    syms = np.stack([qpsk(rng, n_data * nd) for _ in range(F)])
    assert sim.load_symbols(syms) == F
    sim.step(F)
    res = sim.results(F)
    e = sim.edges(F)
    blk = fgm.RadarSimFlowgraph(o, ctx=ctx, N_rx=R, fft_len=N, N_sym_radar=S, fused_demod=True, **tg)
    assert blk.pad_tail == sim.pad_tail and blk.N_pre == sim.N_pre
    assert sim.sum_on_spectrum == summed                                         # (the summed pass was taken, not its fallback)
    rep = REPORT.setdefault("radar/device_resident_vs_blocks/config_B_4x4_N256_S64" + ("/summed" if summed else "/one_by_one"), {})
    for f in range(F):
        pads = [e["bursts"][f, t, sim.n_in:] for t in range(sim.T)]
        assert all(0.007 < p.real.std() < 0.014 for p in pads)
        bres, be = blk.run_packet(syms[f], 2, fgm.DATA, sim.pdu_len, sources=dict(pads=pads, noise=np.zeros((R, sim.n_burst), np.complex64)))
        assert np.array_equal(e["tx_f"][f], be["tx_f"])                          # batched precoder == per-packet precoder, bit for bit
        for k, a, b in (("tx_t", e["tx_t"][f], be["tx_t"]), ("bursts", e["bursts"][f], be["bursts"]), ("rx_t", e["rx_t"][f], be["rx_t"]),
                        ("H", e["H"][f], be["H"][:, :N]), ("map", e["map"][f], be["map"])):
            err = rel_err(a, b)
            rep[k] = max(rep.get(k, 0.0), err)
            assert err <= 1e-5, (f, k, err)
        compare_results(res[f], bres, exact_floats=False)
    assert any(r.published for r in res)


def test_radar_flowgraph_with_background_removal_across_packets(jrc, ctx, ofdm64):
    """the radar graph with mimo_ofdm_radar's background state switched on, as in the USRP flowgraph (examples/usrp/mimo_ofdm_jrc_TRX.grc:1216;
    lib/mimo_ofdm_radar_impl.cc:276-300): eight packets in sequence, record_len 3, recording switched off after the fifth — the channel
    estimate of packet n has the mean of the recorded history subtracted, so every edge from `H` on depends on the packets before it.  HIP
    graph and oracle graph packet by packet, chained: `H` bit-equal given equal inputs is not asked here (the inputs differ by 1e-7), the
    north-star tolerance is; records as in the other shapes.  A static clutter target (in every packet) next to one that appears in packet 4:
    after the history has filled, the estimator reports the new target."""
    import radar_sim_flowgraph as fgm
    o = tables_64(ofdm64, 4)
    kw = dict(N_rx=2, fft_len=64, seed=9, background_removal=True, background_recording=True, record_len=3, snr_threshold=10.0)
    clutter = dict(trgt_range=[8.0], trgt_velocity=[0.0], trgt_rcs_dbsm=[22.0], trgt_angle=[-25.0])
    both = dict(trgt_range=[8.0, 30.0], trgt_velocity=[0.0, 0.0], trgt_rcs_dbsm=[22.0, 46.0], trgt_angle=[-25.0, 30.0])
    hip = fgm.RadarSimFlowgraph(o, ctx=ctx, **kw, **clutter)
    orc = fgm.RadarSimFlowgraph(o, blocks=oracle_blocks, **kw, **clutter)
    hip2 = fgm.RadarSimFlowgraph(o, ctx=ctx, **kw, **both)                       # only their target simulators are used from packet 4 on
    orc2 = fgm.RadarSimFlowgraph(o, blocks=oracle_blocks, **kw, **both)
    rng = np.random.default_rng(31)
    rep = REPORT.setdefault("radar/background_removal/operating_point_4x2_N64", {})
    ns = jrc.n_ofdm_sym(2, 48, 100)
    seen = []
    for n in range(8):
        if n == 4:
            hip.target_sims, orc.target_sims = hip2.target_sims, orc2.target_sims
        if n == 5:
            hip.radar.set_background_record(False)
            orc.radar.set_background_record(False)
        sym = qpsk(rng, ns * 48)
        gres, ge = hip.run_packet(sym, 2, fgm.DATA, 100)
        ores, oe = orc.run_packet(sym, 2, fgm.DATA, 100, sources=dict(pads=ge["pads"], noise=ge["noise"]))
        for k in RADAR_CF32_EDGES:
            err = rel_err(ge[k], oe[k])
            rep["chained:" + k] = max(rep.get("chained:" + k, 0.0), err)
            assert err <= TOL, (n, k, err)
        compare_results(gres, ores, exact_floats=False)
        seen.append((gres.published, gres.range_val, gres.angle_val))
    assert hip.radar.ring_size() == 3
    assert abs(seen[4][1] - 30.0) < 1.0 and abs(seen[4][2] - 30.0) < 3.0         # clutter at 8 m removed: the new target is the peak
    assert np.abs(ge["H"]).max() > 0


def test_radar_flowgraph_with_tx_interleave(jrc, ctx, ofdm64):
    """enable_tx_interleave (lib/mimo_ofdm_radar_impl.cc:262-269): the pairs leave the radar block transmitter-major (p = t R + r), which re-orders the
    rows of `H` and with them the angle axis of everything downstream; the operating point once more with it switched on, chained"""
    import radar_sim_flowgraph as fgm
    o = tables_64(ofdm64, 4)
    kw = dict(trgt_range=[14.0], trgt_velocity=[0.0], trgt_rcs_dbsm=[20.0], trgt_angle=[-20.0], N_rx=2, fft_len=64, seed=6, enable_tx_interleave=True)
    hip = fgm.RadarSimFlowgraph(o, ctx=ctx, **kw)
    orc = fgm.RadarSimFlowgraph(o, blocks=oracle_blocks, **kw)
    plain = fgm.RadarSimFlowgraph(o, ctx=ctx, **dict(kw, enable_tx_interleave=False))
    rng = np.random.default_rng(41)
    ns = jrc.n_ofdm_sym(2, 48, 100)
    sym = qpsk(rng, ns * 48)
    rep = REPORT.setdefault("radar/tx_interleave/operating_point_4x2_N64", {})
    gres, ge = hip.run_packet(sym, 2, fgm.DATA, 100)
    src = dict(pads=ge["pads"], noise=ge["noise"])
    ores, oe = orc.run_packet(sym, 2, fgm.DATA, 100, sources=src)
    for k in RADAR_CF32_EDGES:
        err = rel_err(ge[k], oe[k])
        rep["chained:" + k] = err
        assert err <= TOL, (k, err)
    compare_results(gres, ores, exact_floats=False)
    _, pe = plain.run_packet(sym, 2, fgm.DATA, 100, sources=src)
    T, R = 4, 2
    perm = [r * T + t for t in range(T) for r in range(R)]                    # row p = t R + r of the interleaved estimate is row r T + t of the plain one
    assert np.array_equal(ge["H"], pe["H"][perm]) and not np.array_equal(ge["H"], pe["H"])


@pytest.mark.parametrize("smoothing", [False, True], ids=["per_subcarrier", "mean_channel"])
def test_comm_flowgraph_with_phased_steering(jrc, ctx, ofdm64, smoothing):
    """phased_steering (lib/mimo_precoder_impl.cc:850-853): column 0 of the steering matrix is sqrt(T) conj(h)/|h|, the rest zero — the comm graph at the
    .grc's point with the DATA packets steered that way after the sounding, chained edge by edge"""
    import comm_sim_flowgraph as cfm
    rep = REPORT.setdefault("comm/N64/phased/%s" % ("mean" if smoothing else "per_sc"), {})
    kw = dict(mcs=2, estimator=0, seed=12, channel="los", smoothing=smoothing, phased_steering=True)
    hip = cfm.CommSimFlowgraph(ofdm64, ctx=ctx, **kw)
    orc = cfm.CommSimFlowgraph(ofdm64, blocks=oracle_blocks, **kw)
    rng = np.random.default_rng(77)
    for pdu, steer in [(bytes([1]) + b"sounding", False), (bytes([2]) + rng.integers(0, 256, 120, dtype=np.uint8).tobytes(), True),
                       (bytes([2]) + rng.integers(0, 256, 333, dtype=np.uint8).tobytes(), True)]:
        gok, gpay, ginfo = hip.send(pdu, steer=steer)
        ge = ginfo["edges"]
        ook, opay, oinfo = orc.send(pdu, steer=steer, sources=dict(pads=ge["pads"], noise=ge["noise"]))
        compare_comm_edges(ge, oinfo["edges"], rep, "chained:", TOL)
        assert (gok, gpay) == (ook, opay)
        if steer:
            qg, qo = ge["steering"].reshape(-1, 4, 4), oinfo["edges"]["steering"].reshape(-1, 4, 4)
            dead = np.isnan(qo).all(axis=(1, 2))                  # carriers without a channel estimate: Q * sqrt(T) / Q.norm() = 0 / 0 everywhere (:851)
            assert np.array_equal(np.isnan(qg), np.isnan(qo)) and not np.isnan(qo[~dead]).any()
            assert dead.sum() == (0 if smoothing else int((np.asarray(ofdm64["ltf_64"]) == 0).sum())) and rel_err(qg[~dead], qo[~dead]) <= TOL
            assert np.abs(qg[~dead][..., 1:]).max() == 0 and np.abs(qg[~dead][..., 0]).min() > 0       # phased: only column 0
            assert gok and gpay == pdu


def test_radar_aided_precoding_across_the_two_flowgraphs(jrc, ctx, ofdm64):
    """the paper's loop, radar -> comm, over both graphs: the radar graph sees the user as its target (10 m, 20 deg), range_angle_estimator's angle goes to the
    precoder (in the reference through the last line of radar_log.csv: lib/range_angle_estimator_impl.cc:266-269 -> lib/mimo_precoder_impl.cc:901-983), which
    steers the DATA packets of the comm graph towards it with no sounding at all; the .grc's line-of-sight channel points the same way (theta 20 deg).
    HIP graphs against oracle graphs, each pair chained on its own radar estimate; and the beam really forms: the equalizer's precoded channel mean grows by
    about ||h|| / |sum h_t / sqrt(T)|."""
    import comm_sim_flowgraph as cfm
    import radar_sim_flowgraph as fgm
    o = tables_64(ofdm64, 4)
    rkw = dict(trgt_range=[10.0], trgt_velocity=[0.0], trgt_rcs_dbsm=[20.0], trgt_angle=[20.0], N_rx=2, fft_len=64, seed=14)
    hr, orr = fgm.RadarSimFlowgraph(o, ctx=ctx, **rkw), fgm.RadarSimFlowgraph(o, blocks=oracle_blocks, **rkw)
    rng = np.random.default_rng(51)
    ns = jrc.n_ofdm_sym(2, 48, 100)
    sym = qpsk(rng, ns * 48)
    gres, ge = hr.run_packet(sym, 2, fgm.DATA, 100)
    ores, _ = orr.run_packet(sym, 2, fgm.DATA, 100, sources=dict(pads=ge["pads"], noise=ge["noise"]))
    compare_results(gres, ores, exact_floats=False)
    assert gres.published == 1 and gres.angle_val == ores.angle_val and abs(gres.angle_val - 20.0) < 2.0
    ckw = dict(mcs=3, estimator=0, seed=15, channel="los", theta=20.0)
    hc, oc = cfm.CommSimFlowgraph(ofdm64, ctx=ctx, **ckw), cfm.CommSimFlowgraph(ofdm64, blocks=oracle_blocks, **ckw)
    rep = REPORT.setdefault("comm/N64/radar_aided", {})
    gains = {}
    for steer in (False, True):
        pdu = bytes([2]) + rng.integers(0, 256, 200, dtype=np.uint8).tobytes()
        gok, gpay, ginfo = hc.send(pdu, steer=steer, radar_angle=gres.angle_val)
        e = ginfo["edges"]
        ook, opay, oinfo = oc.send(pdu, steer=steer, radar_angle=ores.angle_val, sources=dict(pads=e["pads"], noise=e["noise"]))
        compare_comm_edges(e, oinfo["edges"], rep, "chained:", TOL)
        assert (gok, gpay) == (ook, opay) == (True, pdu)
        if steer:
            assert rel_err(e["steering"], oinfo["edges"]["steering"]) <= TOL
        end = [ev for ev in e["eq_events"] if ev["kind"] == 2][0]
        gains[steer] = float(np.abs(end["chan_mean"][0]))
    assert gains[True] > 2.0 * gains[False], gains                 # ||h|| = 2 g against |sum_t h_t| / 2 = 0.82 g at 20 deg


def test_comm_flowgraph_with_radar_streams_on_the_null_space(jrc, ctx, ofdm64):
    """use_radar_streams (lib/mimo_precoder_impl.cc:560-631): the other N_tx - 1 columns of the steering matrix carry radar symbols; steered from the sounding
    they lie in the null space of the user's channel, so the PDU still decodes while three more streams are on the air.  Edge by edge, chained."""
    import comm_sim_flowgraph as cfm
    rep = REPORT.setdefault("comm/N64/radar_streams", {})
    kw = dict(mcs=2, estimator=0, seed=21, channel="los", smoothing=True)
    hip = cfm.CommSimFlowgraph(ofdm64, ctx=ctx, **kw)
    orc = cfm.CommSimFlowgraph(ofdm64, blocks=oracle_blocks, **kw)
    rng = np.random.default_rng(88)
    pdus = [(bytes([1]) + b"sounding", False, False), (bytes([2]) + rng.integers(0, 256, 150, dtype=np.uint8).tobytes(), True, True)]
    for pdu, steer, streams in pdus:
        rs = None
        if streams:
            ns = jrc.n_ofdm_sym(2, 48, len(pdu) + 4)
            rs = qpsk(rng, 3 * ns * 64).reshape(3, ns, 64)
        gok, gpay, ginfo = hip.send(pdu, steer=steer, radar_streams=rs)
        ge = ginfo["edges"]
        ook, opay, oinfo = orc.send(pdu, steer=steer, radar_streams=rs, sources=dict(pads=ge["pads"], noise=ge["noise"]))
        compare_comm_edges(ge, oinfo["edges"], rep, "chained:", TOL)
        assert (gok, gpay) == (ook, opay)
        if streams:
            assert gok and gpay == pdu
            plain = hip.precoder.work(ge["symbols"], 2, 2, len(pdu) + 4, **hip.steering())
            assert rel_err(ge["tx_f"], plain) > 0.1                      # the radar streams really are on the air


@pytest.mark.parametrize("i", range(max(2, int(os.environ.get("JRC_FUZZ_N", "24")) // 6)))
def test_radar_flowgraph_on_drawn_scenes(jrc, ctx, ofdm64, i):
    """the composed radar simulation flowgraph on drawn scenes: 1-3 targets at drawn range / velocity / cross-section / angle, 1, 2 or 4 receive
    antennas, a drawn PDU length, MCS and packet type, the demodulator fused or block by block, both random sources drawn once and replayed:
    HIP blocks against the oracle's blocks chained (1e-4 on every complex edge, lengths equal) and block by block on the HIP graph's edges
    (2e-5; copies, A1 and the estimator's record bit for bit)"""
    import radar_sim_flowgraph as fgm
    rng = np.random.default_rng(int(os.environ.get("JRC_FUZZ_SEED", "20261002")) + 23000 + i)
    R = int(rng.choice([1, 2, 4]))
    K = int(rng.integers(1, 4))
    kw = dict(trgt_range=list(rng.uniform(4, 45, K)), trgt_velocity=list(rng.uniform(-30, 30, K)), trgt_rcs_dbsm=list(rng.uniform(10, 35, K)),
              trgt_angle=list(rng.uniform(-60, 60, K)), N_rx=R, fft_len=64, seed=int(rng.integers(0, 1 << 30)))
    fused = bool(rng.integers(0, 2))
    hip = fgm.RadarSimFlowgraph(ofdm64, ctx=ctx, fused_demod=fused, **kw)
    orc = fgm.RadarSimFlowgraph(ofdm64, blocks=oracle_blocks, fused_demod=fused, **kw)
    mcs, ptype, nbytes = int(rng.choice([0, 2, 3])), int(rng.choice([fgm.DATA, fgm.NDP])), int(rng.integers(20, 300))
    nd = len(ofdm64["data_subcarriers"])
    n_data = jrc.n_ofdm_sym(mcs, nd, nbytes)
    sym = qpsk(rng, n_data * nd)
    draw = dict(i=i, R=R, fused=fused, mcs=mcs, ptype=ptype, nbytes=nbytes, **{k: [round(float(v), 2) for v in kw[k]] for k in ("trgt_range", "trgt_velocity", "trgt_rcs_dbsm", "trgt_angle")})
    gres, ge = hip.run_packet(sym, mcs, ptype, nbytes)
    src = dict(pads=ge["pads"], noise=ge["noise"])
    ores, oe = orc.run_packet(sym, mcs, ptype, nbytes, sources=src)
    assert ge["lengths"] == oe["lengths"], draw
    for k in RADAR_CF32_EDGES:
        assert ge[k].shape == oe[k].shape and rel_err(ge[k], oe[k]) <= TOL, (draw, k, rel_err(ge[k], oe[k]))
    bres, be = orc.run_packet(sym, mcs, ptype, nbytes, sources=src, force={k: ge[k] for k in RADAR_CF32_EDGES})
    for k in RADAR_CF32_EDGES:
        assert rel_err(ge[k], be[k]) <= 2e-5, (draw, k, rel_err(ge[k], be[k]))
    assert np.array_equal(ge["bursts"], be["bursts"]) and np.array_equal(ge["transposed"], be["transposed"]) and np.array_equal(ge["H"], be["H"]), draw
    try:
        compare_results(gres, bres, exact_floats=True)
    except AssertionError as e:
        raise AssertionError((draw, e))


@pytest.mark.parametrize("i", range(max(2, int(os.environ.get("JRC_FUZZ_N", "24")) // 6)))
def test_comm_flowgraph_on_drawn_links(jrc, ctx, ofdm64, i):
    """the composed comm simulation flowgraph on drawn links: MCS, estimator, line-of-sight geometry (distance, angle) or a drawn flat channel,
    carrier offset, lead, a sounding packet and then PDUs of drawn length without and with the steering derived from it; both random sources
    replayed: HIP blocks against the oracle's blocks chained — every complex edge to 1e-4, tags / events / lengths / decoded bytes and the CRC
    verdict equal"""
    import comm_sim_flowgraph as cfm
    rng = np.random.default_rng(int(os.environ.get("JRC_FUZZ_SEED", "20261002")) + 25000 + i)
    mcs, est = int(rng.integers(0, 5)), int(rng.integers(0, 2))
    channel = str(rng.choice(["los", "flat"]))
    kw = dict(mcs=mcs, estimator=est, seed=int(rng.integers(0, 1 << 30)), channel=channel, smoothing=bool(rng.integers(0, 2)))
    if channel == "los":
        kw.update(distance=float(rng.uniform(5, 40)), theta=float(rng.uniform(-50, 50)))
    hip = cfm.CommSimFlowgraph(ofdm64, ctx=ctx, **kw)
    orc = cfm.CommSimFlowgraph(ofdm64, blocks=oracle_blocks, **kw)
    cfo, lead = float(rng.uniform(-0.01, 0.01)), int(rng.integers(400, 1500))
    draw = dict(i=i, cfo=cfo, lead=lead, **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in kw.items()})
    rep = {}
    pdus = [(bytes([1]) + b"sounding", False)] + [(bytes([2]) + rng.integers(0, 256, int(rng.integers(10, 260)), dtype=np.uint8).tobytes(), bool(k)) for k in range(2)]
    for pdu, steer in pdus:
        gok, gpay, ginfo = hip.send(pdu, steer=steer, cfo=cfo, lead=lead)
        ge = ginfo["edges"]
        ook, opay, oinfo = orc.send(pdu, steer=steer, cfo=cfo, lead=lead, sources=dict(pads=ge["pads"], noise=ge["noise"]))
        try:
            # (a decode that fails on a marginal link fails in both graphs; what it leaves in the payload hangs on hard decisions of symbols
            # that sit on a boundary, so the bytes are only compared when the CRC holds)
            sta = est == 1                                        # (with decision-directed tracking a boundary symbol may cost one graph the CRC: checked below)
            assert sta or (gok == ook and (not gok or gpay == opay))
            oe_ = oinfo["edges"]
            if not ge["detector_tags"]:                          # a drawn channel in a deep fade: nothing detected — by both graphs, and nothing behind it exists
                assert not oe_["detector_tags"] and ge["detector_out"].size == oe_["detector_out"].size == 0 and not gok
                assert all((ge.get(k) is None) == (oe_.get(k) is None) for k in ("sync_out", "y", "eq_out", "steering", "chan_est"))
                continue
            if steer and (ge.get("steering") is not None or oe_.get("steering") is not None):   # (no sounding received: nothing to steer with)
                assert rel_err(ge["steering"], oe_["steering"]) <= TOL
            # (the normalised metric: the reference's running sums drift by an amount that depends on the power steps that went through them, the device sums windows — docs/history.md §5.1, §5.2; what the detector decides from it is compared exactly)
            # the coarse CFO is the angle of the metric's correlation sum: a last-digit difference of it (same drift) is a phase ramp over the samples the
            # detector de-rotates, and everything behind it inherits the ramp until the equalizer's own tracking takes it out
            d_cfo = max([abs(a[1] - b[1]) for a, b in zip(ge["detector_tags"], oe_["detector_tags"])] + [0.0])
            # decision-directed tracking (STA): a symbol that lands ON a decision boundary (1e-6 of it) is decided one way by one graph and the other way
            # by the other, and that carrier's channel estimate — so everything behind it on that carrier — parts ways.  Such a carrier is compared
            # up to the symbol behind which the two decisions differ, and only if they do differ there; everything else as always.
            ge_cmp, on_boundary = ge, False
            if est == 1 and ge.get("eq_out") is not None and oe_.get("eq_out") is not None and ge["eq_out"].shape == oe_["eq_out"].shape and ge["eq_out"].size:
                a, b = ge["eq_out"], oe_["eq_out"]
                off = np.abs(a - b) > 1e-3 * max(1e-9, float(np.abs(b).max()))
                if off.any():
                    bpsc = 1 if mcs <= 1 else (2 if mcs <= 3 else 4)
                    patched = a.copy()
                    for c in sorted(set(np.argwhere(off)[:, 1])):
                        s0 = int(np.argwhere(off[:, c])[0, 0])
                        assert s0 >= 1 and oracle.constellation_decide(bpsc, a[s0 - 1, c]) != oracle.constellation_decide(bpsc, b[s0 - 1, c]), (c, s0)
                        patched[s0:, c] = b[s0:, c]
                    ge_cmp, on_boundary = dict(ge, eq_out=patched, crc_ok=oe_.get("crc_ok"), payload=oe_.get("payload")), True
            # (the offset the stream_start event announces is coarse - fine in Hz: near zero it is the difference of two 1e-2 rad / sample numbers
            # known to 1e-9, so it is held to 1e-4 of a 1e-3 rad / sample offset — 2 Hz at 125 MS/s — rather than of itself)
            compare_comm_edges(ge_cmp, oe_, rep, "chained:", TOL + 1.5 * d_cfo * max(1, ge["detector_out"].size), freq_floor=2e4, payload_equal=bool(gok))
            if sta and not on_boundary:
                assert gok == ook and (not gok or gpay == opay)
        except AssertionError as e:
            raise AssertionError((draw, len(pdu), steer, e))
