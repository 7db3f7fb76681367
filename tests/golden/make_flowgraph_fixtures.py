"""Mints tests/golden/flowgraph_fixtures.npz: the reference's two simulation flowgraphs at their .grc operating point, wired by
examples/radar_sim_flowgraph.py / examples/comm_sim_flowgraph.py over the CPU oracle's blocks (tests/oracle_blocks.py), with seeded inputs and
seeded draws of the random sources: the inputs, the sources and the outputs on the key edges.  Regression fixtures of the composed oracle
graphs (they pin the restatement AND the wiring against silent change) and committed data the GPU tier compares the HIP graphs with; not
reference outputs (the reference cannot be built here).

    python tests/golden/make_flowgraph_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "examples"))
import oracle_blocks  # noqa: E402


def qpsk(rng, n):
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    return pts[rng.integers(0, 4, n)].astype(np.complex64)


def record(r):
    return np.array([r.peak_range_idx, r.peak_angle_idx, r.angle_null_idx, r.n_noise_samples, r.published], np.int64), \
        np.array([r.peak_power, r.noise_power, r.snr_est, r.range_val, r.angle_val], np.float32)


RADAR_KW = dict(trgt_range=[17.0], trgt_velocity=[6.0], trgt_rcs_dbsm=[20.0], trgt_angle=[-35.0], N_rx=2, fft_len=64, seed=0)
COMM_KW = dict(mcs=3, estimator=0, seed=0, channel="los", smoothing=False)


def radar_case(o, rng):
    import radar_sim_flowgraph as fgm
    fg = fgm.RadarSimFlowgraph(o, blocks=oracle_blocks, **RADAR_KW)
    ns = oracle_blocks.n_ofdm_sym(2, 48, 100)
    sym = qpsk(rng, ns * 48)
    n_burst = (4 + 1 + 4 + ns) * 80 + fg.pad_tail
    pads = np.stack([(0.01 * (rng.standard_normal(fg.pad_tail) + 1j * rng.standard_normal(fg.pad_tail))).astype(np.complex64) for _ in range(4)])
    noise = fg.draw_noise(n_burst)
    res, e = fg.run_packet(sym, 2, fgm.DATA, 100, sources=dict(pads=list(pads), noise=noise))
    ri, rf = record(res)
    return dict(radar_symbols=sym, radar_pads=pads, radar_noise=noise, radar_tx_f=e["tx_f"], radar_rx_f=e["rx_f"], radar_H=e["H"][:, :64],
                radar_map_rows=e["map"][::37], radar_map_abs_sum=np.float64(np.abs(e["map"].astype(np.complex128)).sum()),
                radar_result_ints=ri, radar_result_floats=rf)


def comm_case(o, rng):
    import comm_sim_flowgraph as cfm
    fg = cfm.CommSimFlowgraph(o, blocks=oracle_blocks, **COMM_KW)
    out = {}
    pdus = [bytes([1]) + b"sounding packet", bytes([2]) + rng.integers(0, 256, 150, dtype=np.uint8).tobytes(),
            bytes([2]) + rng.integers(0, 256, 211, dtype=np.uint8).tobytes()]
    for i, (pdu, steer) in enumerate(zip(pdus, (False, False, True))):
        sym, _ = oracle_blocks.stream_encoder(3, 48).work(pdu)
        n = 640 + 5 + (4 + 1 + 4 + len(sym) // 48) * 80 + fg.pad_tail
        pads_f = np.stack([(0.01 * (rng.standard_normal(5) + 1j * rng.standard_normal(5))).astype(np.complex64) for _ in range(4)])
        pads_t = np.stack([(0.01 * (rng.standard_normal(fg.pad_tail) + 1j * rng.standard_normal(fg.pad_tail))).astype(np.complex64) for _ in range(4)])
        noise = (np.sqrt(fg.noise_var) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        ok, pay, info = fg.send(pdu, steer=steer, sources=dict(pads=[(pads_f[t], pads_t[t]) for t in range(4)], noise=noise))
        e = info["edges"]
        assert ok and pay == pdu, i
        k = "comm%d_" % i
        out.update({k + "pdu": np.frombuffer(pdu, np.uint8), k + "steer": np.int64(steer), k + "pads_front": pads_f, k + "pads_tail": pads_t, k + "noise": noise,
                    k + "tx_f": e["tx_f"], k + "rx": e["rx"], k + "y": e["y"], k + "eq_out": e["eq_out"],
                    k + "detector_tag_offsets": np.array([t[0] for t in e["detector_tags"]], np.int64),
                    k + "sync_tag": np.array([e["sync_tags"][0][0], 0], np.int64), k + "sync_tag_value": np.float64(e["sync_tags"][0][1]),
                    k + "start": np.array([e["eq_events"][0][f] for f in ("offset", "data_bytes", "mcs", "packet_type")], np.int64),
                    k + "crc_ok": np.int64(e["crc_ok"])})
        if e["chan_est"] is not None:
            out[k + "chan_est"] = e["chan_est"]
    return out


def main():
    o = np.load(os.path.join(HERE, "ofdm_config_64.npz"))
    rng = np.random.default_rng(20261003)
    fx = {}
    fx.update(radar_case(o, rng))
    fx.update(comm_case(o, rng))
    dst = os.path.join(HERE, "flowgraph_fixtures.npz")
    np.savez_compressed(dst, **fx)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB,", len(fx), "arrays")


if __name__ == "__main__":
    sys.exit(main())
