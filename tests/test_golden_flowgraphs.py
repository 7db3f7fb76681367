"""tests/golden/flowgraph_fixtures.npz (minted by tests/golden/make_flowgraph_fixtures.py): the two simulation flowgraphs at the .grc
operating point with committed inputs, committed draws of the random sources and the oracle graphs' outputs on the key edges.
CPU tier: the oracle graphs must keep reproducing the committed edges exactly (a change in the restatement or in the wiring shows up
here, not only as a shift both sides of a live comparison share).  GPU tier: the HIP graphs against the committed data — no oracle call."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

sys.path.insert(0, GOLDEN)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from make_flowgraph_fixtures import COMM_KW, RADAR_KW, record  # noqa: E402

TOL = 1e-4
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "flowgraph_fixtures.npz"))


def run_radar(fx, ofdm64, **blocks):
    import radar_sim_flowgraph as fgm
    fg = fgm.RadarSimFlowgraph(ofdm64, **RADAR_KW, **blocks)
    return fg.run_packet(fx["radar_symbols"], 2, fgm.DATA, 100, sources=dict(pads=list(fx["radar_pads"]), noise=fx["radar_noise"]))


def run_comm(fx, ofdm64, i, **blocks):
    import comm_sim_flowgraph as cfm
    fg = cfm.CommSimFlowgraph(ofdm64, **COMM_KW, **blocks)
    outs = []
    for j in range(i + 1):                    # the graph carries state from PDU to PDU (scrambler seed, sounded channel): replay in order
        k = "comm%d_" % j
        pads = [(fx[k + "pads_front"][t], fx[k + "pads_tail"][t]) for t in range(4)]
        outs.append(fg.send(fx[k + "pdu"].tobytes(), steer=bool(fx[k + "steer"]), sources=dict(pads=pads, noise=fx[k + "noise"])))
    return outs


def check_comm(fx, outs, exact):
    for j, (ok, pay, info) in enumerate(outs):
        k, e = "comm%d_" % j, info["edges"]
        assert ok and pay == fx[k + "pdu"].tobytes() and int(e["crc_ok"]) == int(fx[k + "crc_ok"])
        assert [t[0] for t in e["detector_tags"]] == list(fx[k + "detector_tag_offsets"])                 # integer edges: exact on both tiers
        assert e["sync_tags"][0][0] == int(fx[k + "sync_tag"][0])
        assert [e["eq_events"][0][f] for f in ("offset", "data_bytes", "mcs", "packet_type")] == list(fx[k + "start"])
        for name in ("tx_f", "rx", "y", "eq_out"):
            if exact:
                assert np.array_equal(e[name], fx[k + name]), (j, name)
            else:
                assert rel_err(e[name], fx[k + name]) < TOL, (j, name, rel_err(e[name], fx[k + name]))
        assert abs(e["sync_tags"][0][1] - float(fx[k + "sync_tag_value"])) <= (0 if exact else 1e-3 * abs(float(fx[k + "sync_tag_value"])))
        if k + "chan_est" in fx.files:
            assert e["chan_est"] is not None
            assert np.array_equal(e["chan_est"], fx[k + "chan_est"]) if exact else rel_err(e["chan_est"], fx[k + "chan_est"]) < TOL


# ---------------------------------------------------------------- CPU tier
def test_oracle_radar_graph_reproduces_the_committed_edges(fx, ofdm64):
    import oracle_blocks
    res, e = run_radar(fx, ofdm64, blocks=oracle_blocks)
    ri, rf = record(res)
    assert np.array_equal(ri, fx["radar_result_ints"]) and np.array_equal(rf, fx["radar_result_floats"])
    assert np.array_equal(e["tx_f"], fx["radar_tx_f"]) and np.array_equal(e["rx_f"], fx["radar_rx_f"])
    assert np.array_equal(e["H"][:, :64], fx["radar_H"]) and np.array_equal(e["map"][::37], fx["radar_map_rows"])
    assert np.abs(e["map"].astype(np.complex128)).sum() == float(fx["radar_map_abs_sum"])
    assert ri[4] == 1 and abs(rf[3] - 17.0) < 1.0 and abs(rf[4] + 35.0) < 3.0            # and the committed record is the scene's target


def test_oracle_comm_graph_reproduces_the_committed_edges(fx, ofdm64):
    import oracle_blocks
    check_comm(fx, run_comm(fx, ofdm64, 2, blocks=oracle_blocks), exact=True)


# ---------------------------------------------------------------- GPU tier
@gpu
def test_hip_radar_graph_matches_the_committed_edges(jrc, ctx, fx, ofdm64):
    for fused in (True, False):
        res, e = run_radar(fx, ofdm64, ctx=ctx, fused_demod=fused)
        ri, rf = record(res)
        assert np.array_equal(ri, fx["radar_result_ints"])
        assert np.allclose(rf, fx["radar_result_floats"], rtol=1e-3)
        assert rel_err(e["tx_f"], fx["radar_tx_f"]) < TOL and rel_err(e["H"][:, :64], fx["radar_H"]) < TOL
        if not fused:
            assert rel_err(e["rx_f"], fx["radar_rx_f"]) < TOL
        scale = np.abs(fx["radar_map_rows"]).max()
        assert np.abs(e["map"][::37] - fx["radar_map_rows"]).max() < TOL * scale
        assert abs(np.abs(e["map"].astype(np.complex128)).sum() / float(fx["radar_map_abs_sum"]) - 1) < TOL


@gpu
def test_hip_comm_graph_matches_the_committed_edges(jrc, ctx, fx, ofdm64):
    check_comm(fx, run_comm(fx, ofdm64, 2, ctx=ctx), exact=False)
