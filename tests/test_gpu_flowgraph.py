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
    assert res.published == 1 and res.snr_est > 15.0
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
    want = np.zeros_like(e["rx_f"], dtype=np.complex128)
    for r in range(fg.N_rx):
        for t in range(fg.N_tx):
            tau = float(np.float32((2 * rng_m - np.float32(fg.TX_RXs[t][r]) * np.sin(np.deg2rad(az))) / 3e8))   # float32 like :177
            want[r] += amp * np.exp(-2j * np.pi * tau * (f_sc + fc))[None, :] * e["tx_f"][t]
    # the simulator evaluates f + fc in float32 (2048 Hz steps at 24 GHz) on the burst-length frequency grid, so its
    # delay filter is a pure delay only to ~1e-3 of phase per TX
    assert rel_err(e["rx_f"], want) < 1e-2
