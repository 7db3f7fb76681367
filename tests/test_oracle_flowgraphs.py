"""CPU tier: the wiring of examples/radar_sim_flowgraph.py and examples/comm_sim_flowgraph.py run over the oracle's blocks alone
(tests/oracle_blocks.py) — the graphs the GPU tier compares the HIP blocks with are themselves sane: the radar graph reports its target, the
comm graph returns the PDU, the integer bookkeeping follows the reference's rules.  (The examples never import the oracle; the block set is handed in.)"""
import os
import sys

import numpy as np

import oracle_blocks
from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "examples"))


def qpsk(rng, n):
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    return pts[rng.integers(0, 4, n)].astype(np.complex64)


def test_oracle_radar_graph_reports_its_target(ofdm64):
    import radar_sim_flowgraph as fgm
    fg = fgm.RadarSimFlowgraph(ofdm64, [14.0], [0.0], [20.0], [-25.0], blocks=oracle_blocks, seed=2)
    assert fg.add_order == [[0, 1, 3, 2], [0, 1, 2, 3]]                  # the .grc's adder ports: RX1 takes TX4 before TX3
    rng = np.random.default_rng(1)
    ns = oracle_blocks.n_ofdm_sym(2, 48, 100)
    n_total = 4 + 1 + 4 + ns
    n_burst = n_total * 80 + 240
    src = dict(pads=[(0.01 * (rng.standard_normal(240) + 1j * rng.standard_normal(240))).astype(np.complex64) for _ in range(4)],
               noise=fg.draw_noise(n_burst))
    res, e = fg.run_packet(qpsk(rng, ns * 48), 2, fgm.DATA, 100, sources=src)
    assert res.published == 1 and abs(res.range_val - 14.0) < 0.8 and abs(res.angle_val + 25.0) < 2.5
    assert e["lengths"] == dict(precoder_out=n_total, mod_out=n_total * 80, zero_pad_out=n_burst, tsim_out=n_burst, cp_remover_out=n_total + 3,
                                radar_out=8, transpose_out=512, map_rows=512, radar_consumed_tx=n_total, radar_consumed_rx=n_total + 3)
    assert e["map"].shape == (512, 128) and e["H"].shape == (8, 512) and not e["H"][:, 64:].any()      # zero padding of the radar block (:312-315)
    # other symbols go in, but every block reads the first run's tensor on its input edge: the map is the first run's again
    res2, e2 = fg.run_packet(qpsk(np.random.default_rng(5), ns * 48), 2, fgm.DATA, 100, sources=src,
                             force={k: e[k] for k in ("tx_f", "tx_t", "bursts", "sims", "rx_t", "rx_f", "H", "range_profile", "transposed")})
    assert not np.array_equal(e2["tx_f"], e["tx_f"])
    assert np.array_equal(e2["map"], e["map"]) and res2.peak_range_idx == res.peak_range_idx


def test_oracle_comm_graph_returns_the_pdu_and_sounds_the_channel(ofdm64):
    import comm_sim_flowgraph as cfm
    fg = cfm.CommSimFlowgraph(ofdm64, mcs=3, estimator=0, seed=3, blocks=oracle_blocks, channel="los", smoothing=False)
    assert (fg.pad_front, fg.pad_tail, fg.sync_length, fg.ignore_gap, fg.corr_window_size) == (5, 490, 320, 640, 32)
    rng = np.random.default_rng(4)

    def sources(pdu):
        sym, _ = oracle_blocks.stream_encoder(3, 48).work(pdu)
        n = 640 + 5 + (4 + 1 + 4 + len(sym) // 48) * 80 + fg.pad_tail
        pads = [((0.01 * (rng.standard_normal(5) + 1j * rng.standard_normal(5))).astype(np.complex64),
                 (0.01 * (rng.standard_normal(fg.pad_tail) + 1j * rng.standard_normal(fg.pad_tail))).astype(np.complex64)) for _ in range(4)]
        return dict(pads=pads, noise=(np.sqrt(fg.noise_var) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64))

    ndp = bytes([1]) + b"sounding"
    ok, pay, info = fg.send(ndp, sources=sources(ndp))
    assert info["start"]["packet_type"] == 1 and fg.chan_est is not None and fg.chan_est.shape == (64, 4)
    used = np.abs(fg.chan_est).sum(axis=1) > 0
    est = fg.chan_est[used].mean(axis=0)
    assert np.abs(est / est[0] - fg.h / fg.h[0]).max() < 0.1             # the line-of-sight array response, up to a common factor
    for steer in (False, True):
        pdu = bytes([2]) + rng.integers(0, 256, 180, dtype=np.uint8).tobytes()
        ok, pay, info = fg.send(pdu, steer=steer, sources=sources(pdu))
        assert ok and pay == pdu and info["start"]["mcs"] == 3 and info["start"]["data_bytes"] == len(pdu) + 4
        assert [t[0] for t in info["edges"]["detector_tags"]][:1] and info["edges"]["eq_consumed"] == len(info["edges"]["y"])
