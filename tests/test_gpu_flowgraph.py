"""GPU tier: the reference's radar simulation flowgraph (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2165-2232)
wired over this package's blocks (examples/radar_sim_flowgraph.py): precoder -> OFDM mod -> target_simulator per TX ->
sum + noise -> CP removal + FFT -> mimo_ofdm_radar -> range IFFT -> transpose -> angle FFT -> estimator.  The simulated
target comes back at its range / azimuth, and the frequency-domain RX symbols agree with the analytic point-target model."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err

sys.path.insert(0, os.path.join(ROOT, "examples"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rng_m,az,ptype", [(10.0, 0.0, 2), (10.0, 20.0, 2), (25.0, -30.0, 1), (17.0, 45.0, 2)])
def test_target_is_recovered(jrc, ctx, ofdm64, rng_m, az, ptype):
    import radar_sim_flowgraph as fgm
    fg = fgm.RadarSimFlowgraph(ofdm64, [rng_m], [0.0], [20.0], [az], ctx=ctx)
    rng = np.random.default_rng(3)
    nbytes, mcs = 100, 2
    ns = jrc.n_ofdm_sym(mcs, 48, nbytes)
    res, edges = fg.run_packet(fgm.qpsk_symbols(rng, ns * 48), mcs, ptype, nbytes)
    # noise sources as GNU Radio's GR_GAUSSIAN: sqrt(noise_var) per component; the far NDP case sits at the 15 dB publishing threshold
    assert res.snr_est > 12.0 and res.published == int(res.snr_est >= 15.0) and (res.published == 1 or rng_m > 20)
    assert abs(res.range_val - rng_m) < 1.2 / 2 + 0.15          # half a range cell (R_res 1.2 m) + one interpolated bin
    assert abs(res.angle_val - az) < 2.0
    assert edges["map"].shape == (64 * 8, 8 * 16)


def test_rx_symbols_match_the_point_target_model(jrc, ctx, ofdm64):
    """noise-free, static target: RX_r[sym][sc] = a * mult * sum_t exp(-j 2 pi tau_{r,t} (f_sc + fc)) TX_t[sym][sc]"""
    import radar_sim_flowgraph as fgm
    rng_m, az = 12.0, 25.0
    fg = fgm.RadarSimFlowgraph(ofdm64, [rng_m], [0.0], [20.0], [az], ctx=ctx)
    rng = np.random.default_rng(4)
    ns = jrc.n_ofdm_sym(2, 48, 60)
    _, e = fg.run_packet(fgm.qpsk_symbols(rng, ns * 48), 2, 2, 60, noise=False)
    N, fs, fc = 64, fg.samp_rate, fg.rf_freq
    f_sc = (np.arange(N) - N // 2) * fs / N
    amp = 3e8 * np.sqrt(100.0) / (4 * np.pi) ** 1.5 / rng_m ** 2 / fc * fg.tx_multiplier * np.sqrt(N)
    n_total = e["tx_f"].shape[1]
    rx_f = e["rx_f"][:, :n_total]                                 # the three zero_pad symbols behind the burst carry no signal
    want = np.zeros_like(rx_f, dtype=np.complex128)
    for r in range(fg.N_rx):
        for t in range(fg.N_tx):
            tau = float(np.float32((2 * rng_m - np.float32(fg.TX_RXs[t][r]) * np.sin(np.deg2rad(az))) / 3e8))   # float32 like :177
            want[r] += amp * np.exp(-2j * np.pi * tau * (f_sc + fc))[None, :] * e["tx_f"][t]
    # the simulator evaluates f + fc in float32 (2048 Hz steps at 24 GHz) on the burst-length frequency grid, so its
    # delay filter is a pure delay only to ~1e-3 of phase per TX
    assert rel_err(rx_f, want) < 1e-2


@pytest.mark.parametrize("mcs", [0, 2, 3])
def test_comm_flowgraph_pdu_round_trip(jrc, ctx, ofdm64, mcs):
    """examples/comm_sim_flowgraph.py: PDU -> stream_encoder -> precoder -> OFDM mod -> 4x1 channel + noise -> CP removal + FFT
    -> equalizer -> stream_decoder -> the same PDU; then the NDP channel estimate steers the precoder (beam-forming gain)"""
    import comm_sim_flowgraph as cfm
    fg = cfm.CommSimFlowgraph(ofdm64, mcs=mcs, ctx=ctx)
    rng = np.random.default_rng(mcs)
    ok, payload, info = fg.send(bytes([1]) + b"sounding packet", snr_db=30.0)
    assert info["start"]["packet_type"] == 1 and info["start"]["data_bytes"] == 20     # NDP payload rides on TX 0/1 only: its CRC
    if ok:                                                                             # depends on |h0 + h1|, the sounding does not
        assert payload == bytes([1]) + b"sounding packet"
    assert fg.chan_est is not None and fg.chan_est.shape == (64, 4)
    used = np.abs(fg.chan_est).sum(axis=1) > 0
    est = fg.chan_est[used].mean(axis=0)
    assert np.abs(est / est[0] - fg.h / fg.h[0]).max() < 0.1                     # LS estimate of the flat channel (up to a common factor)
    snrs = {}
    for steer in (False, True):
        for i in range(3):
            pdu = bytes([2]) + rng.integers(0, 256, 150 + 40 * i, dtype=np.uint8).tobytes()
            ok, payload, info = fg.send(pdu, snr_db=28.0, steer=steer)
            assert ok and payload == pdu
            assert info["start"]["mcs"] == mcs and info["start"]["data_bytes"] == len(pdu) + 4
            snrs[steer] = info["start"]["snr"]
    assert fg.decoder.per <= 100.0 / 7 + 1e-6                               # at most the NDP payload counted as lost
    ok, payload, _ = fg.send(bytes([2]) + bytes(300), snr_db=-3.0)                # hopeless SNR: no false "ok"
    assert not ok
