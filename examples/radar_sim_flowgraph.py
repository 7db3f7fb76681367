#!/usr/bin/env python3
"""The radar simulation flowgraph of the reference (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2165-2232)
wired block by block over this package's MI355X blocks — every block between the stream encoder and the GUI sinks:

  symbols -> mimo_precoder -> per TX: fft_vxx(reverse, shift, window 1/sqrt(N)) + cyclic prefixer -> x tx_multiplier
          -> zero_pad(3 symbols) -> target_simulator per TX (R outputs each) -> blocks_add_xx per RX (+ noise source)
          -> per RX: ofdm_cyclic_prefix_remover -> fft_vxx(forward, shift)
          -> mimo_ofdm_radar(N_pre = 5, N_sym = N_tx) -> fft_vxx(reverse, N*Ir) -> matrix_transpose
          -> fft_vxx(forward, shift, P*Ia) -> range_angle_estimator

Variable names and values follow the .grc (fft_len 64, cp 16, N_rx 2, samp_rate 125 MHz, rf_freq 24 GHz, interp 8 / 16,
tx_multiplier 0.1, noise figure 10 dB, snr_threshold 15).  Needs a GPU: there is no CPU path behind these blocks.

The wiring is written against a *block set* (`blocks=`, default: this package): any namespace offering the same block
classes can be wired identically — tests/test_gpu_flowgraph_parity.py hands in the CPU oracle's blocks to compare the two
graphs edge by edge.  `run_packet` returns the tensor on every block edge plus the integer bookkeeping of the graph
(packet lengths the tagged-stream blocks announce, items the radar block consumes).

  python examples/radar_sim_flowgraph.py [--range 10 --angle 20 --velocity 0 --rcs-dbsm 20]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NDP, DATA = 1, 2


def adder_port_order(N_tx, N_rx):
    """which TX's target_simulator sits on which input port of each RX's blocks_add_xx (the float sum runs in port order).
    In the reference's 4x2 graph RX1's adder takes TX4 on port 2 and TX3 on port 3 (…radar_sim.grc connections
    `target_simulator_0_0_0 -> blocks_add_xx_0:3`, `target_simulator_0_0_0_0 -> blocks_add_xx_0:2`); RX2's is in TX order."""
    order = [list(range(N_tx)) for _ in range(N_rx)]
    if N_tx == 4 and N_rx == 2:
        order[0] = [0, 1, 3, 2]
    return order


class RadarSimFlowgraph:
    def __init__(self, ofdm_config, trgt_range=(10.0,), trgt_velocity=(0.0,), trgt_rcs_dbsm=(20.0,), trgt_angle=(0.0,),
                 N_rx=2, samp_rate=125_000_000, freq=4e9, noise_figure_dB=10.0, tx_multiplier=0.1, interp_factor_range=8,
                 interp_factor_angle=16, sum_targets=True, seed=0, ctx=None, fft_len=64, cp_len=None, N_sym_radar=None,
                 blocks=None, fused_demod=True, snr_threshold=15.0, background_removal=False, background_recording=False, record_len=8,
                 enable_tx_interleave=False):
        if blocks is None:
            import jrc_amd as blocks
            ctx = ctx or blocks.Context(0)
        B = self.B = blocks
        o = ofdm_config
        self.ctx = ctx
        self.fft_len = int(fft_len)
        self.cp_len = self.fft_len // 4 if cp_len is None else int(cp_len)
        self.N_tx, self.N_rx = int(o["N_tx"]), N_rx
        self.samp_rate, self.rf_freq = samp_rate, freq + 20e9
        self.tx_multiplier = tx_multiplier
        self.noise_var = 4.003886160000000e-21 * samp_rate * 10 ** (noise_figure_dB / 10.0)
        self.Ir, self.Ia = interp_factor_range, interp_factor_angle
        self.fused_demod = fused_demod
        self.rng = np.random.default_rng(seed)
        wavelength = 3e8 / self.rf_freq
        T, R, N = self.N_tx, N_rx, self.fft_len
        P = T * R
        # TX1_RXs .. TX4_RXs of the .grc, extended to N_rx receivers 2 wavelengths apart
        self.TX_RXs = [[(1 + t / 2 + 2 * r) * wavelength for r in range(R)] for t in range(T)]
        self.add_order = adder_port_order(T, R)
        self.n_sync = len(o["l_stf_ltf_64"])
        self.N_pre = self.n_sync + 1                                                  # sync words + SIG (…radar_sim.grc:1292)
        self.N_sym_radar = T if N_sym_radar is None else int(N_sym_radar)             # the MIMO-LTFs (:1293-1295)
        self.pad_tail = 3 * (N + self.cp_len)
        self.precoder = B.mimo_precoder(N, T, 1, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                                        o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], ctx=ctx)
        rcs = [10 ** (d / 10.0) for d in trgt_rcs_dbsm]
        self.target_sims = [B.target_simulator(trgt_range, trgt_velocity, rcs, trgt_angle, self.TX_RXs[t], samp_rate,
                                               self.rf_freq, -40.0, False, False, sum_targets=sum_targets, ctx=ctx)
                            for t in range(T)]
        self.zero_pads = [B.zero_pad(False, 0, self.pad_tail, seed=seed + 100 * t, ctx=ctx) for t in range(T)]
        self.cp_remover = B.ofdm_cyclic_prefix_remover(N, self.cp_len, ctx=ctx)
        self.rx_fft = B.fft_vcc(N, True, None, True, ctx=ctx)
        self.radar = B.mimo_ofdm_radar(N, T, R, self.N_sym_radar, self.N_pre, background_removal, background_recording, record_len, self.Ir, enable_tx_interleave, "",
                                       ctx=ctx)       # the simulation flowgraph has both off (…radar_sim.grc:1298-1299), the USRP one on
        self.range_ifft = B.fft_vcc(N * self.Ir, False, None, False, ctx=ctx)
        self.transpose = B.matrix_transpose(N * self.Ir, P, self.Ia, ctx=ctx)
        self.angle_fft = B.fft_vcc(P * self.Ia, True, None, True, ctx=ctx)
        self.range_bins = np.linspace(0, 3e8 * N / (2 * samp_rate), N * self.Ir).astype(np.float32)
        self.angle_bins = (np.arcsin(2 / (P * self.Ia) * (np.arange(0, P * self.Ia) - np.floor(P * self.Ia / 2) + 0.5))
                           * 180 / np.pi).astype(np.float32)
        R_res = 3e8 / (2 * samp_rate)
        angle_res = float(np.rad2deg(np.arcsin(2 / P))) if P > 2 else 15.0            # the .grc expression needs P > 2
        self.estimator = B.range_angle_estimator(P * self.Ia, self.range_bins, self.angle_bins, R_res * 2, angle_res * 2,
                                                 snr_threshold, 0.0, "", False, ctx=ctx)

    def draw_noise(self, n):
        """the two analog_noise_source_x (GR_GAUSSIAN, amp sqrt(noise_var)): amp x (N(0,1) + j N(0,1)) per sample"""
        a = np.float32(np.sqrt(self.noise_var))
        return (a * (self.rng.standard_normal((self.N_rx, n)) + 1j * self.rng.standard_normal((self.N_rx, n)))).astype(np.complex64)

    def run_packet(self, symbols, mcs=2, packet_type=DATA, pdu_len=None, noise=True, sources=None, force=None):
        """one PDU through the whole graph; returns the estimator result and the tensors on the block edges.
        `sources` = {"pads": [T][pad_tail], "noise": [R][n]} replays the random sources of an earlier run (the zero_pad
        noise and the noise sources are draws, not computations); with noise=False both are zero.
        `force` = edges of another run: every block then reads that run's tensor on its input edge instead of the one this
        graph computed (the computed one is still what is returned) — block-by-block comparison of two block sets."""
        B, N, cp, T, R = self.B, self.fft_len, self.cp_len, self.N_tx, self.N_rx
        force = force or {}

        def use(name, val):
            return force[name] if name in force else val

        e = {}
        e["tx_f"] = self.precoder.work(symbols, mcs, packet_type, pdu_len)            # [T][n_total][N]
        tx_f = use("tx_f", e["tx_f"])
        n_total = tx_f.shape[1]
        window = np.full(N, 1 / N ** 0.5, np.float32)
        n_burst = n_total * (N + cp) + self.pad_tail
        e["tx_t"] = np.stack([B.ofdm_mod(tx_f[t], N, cp, window=window, ctx=self.ctx).ravel() * np.float32(self.tx_multiplier)
                              for t in range(T)])
        tx_t = use("tx_t", e["tx_t"])
        bursts = []
        for t in range(T):
            if sources is not None:
                bursts.append(np.concatenate([tx_t[t], sources["pads"][t]]).astype(np.complex64))   # zero_pad's deterministic part is a copy
            elif noise:
                bursts.append(self.zero_pads[t].work(tx_t[t]))                        # zero_pad(0, 3 symbols)
            else:
                bursts.append(np.concatenate([tx_t[t], np.zeros(self.pad_tail, np.complex64)]))
        e["bursts"] = np.stack(bursts)
        e["pads"] = [b[n_burst - self.pad_tail:] for b in bursts]
        bursts = use("bursts", e["bursts"])
        e["sims"] = np.stack([self.target_sims[t].work(bursts[t]) for t in range(T)])  # [T][R][n_burst]
        sims = use("sims", e["sims"])
        nz = sources["noise"] if sources is not None else (self.draw_noise(n_burst) if noise else None)
        e["noise"] = nz
        rx_t = np.zeros((R, n_burst), np.complex64)
        for r in range(R):                                                            # blocks_add_xx: ports summed in order
            acc = sims[self.add_order[r][0]][r].copy()
            for t in self.add_order[r][1:]:
                acc += sims[t][r]
            if nz is not None:
                acc += nz[r]
            rx_t[r] = acc
        e["rx_t"] = rx_t
        rx_t = use("rx_t", rx_t)
        if self.fused_demod:
            e["rx_f"] = np.stack([self.cp_remover.work(rx_t[r], fused_fft=True) for r in range(R)])   # A6 + A7 in one launch
        else:
            e["rx_f"] = np.stack([self.rx_fft.work(self.cp_remover.work(rx_t[r])) for r in range(R)])
        rx_f = use("rx_f", e["rx_f"])
        e["H"] = self.radar.general_work([tx_f[t] for t in range(T)], [rx_f[r] for r in range(R)])     # [P][N*Ir]
        e["range_profile"] = self.range_ifft.work(use("H", e["H"]))
        e["transposed"] = self.transpose.work(use("range_profile", e["range_profile"]))                 # [N*Ir][P*Ia]
        e["map"] = self.angle_fft.work(use("transposed", e["transposed"]))
        res = self.estimator.work(use("map", e["map"]))
        e["lengths"] = dict(precoder_out=self.precoder.calculate_output_stream_length(len(symbols)), mod_out=int(tx_t.shape[1]),
                            zero_pad_out=int(bursts.shape[1]), tsim_out=int(sims.shape[2]),
                            cp_remover_out=self.cp_remover.calculate_output_stream_length(n_burst), radar_out=int(e["H"].shape[0]),
                            transpose_out=self.transpose.calculate_output_stream_length(e["H"].shape[0]),
                            map_rows=int(e["map"].shape[0]), radar_consumed_tx=int(n_total), radar_consumed_rx=int(rx_f.shape[1]))
        return res, e


def load_ofdm_config():
    return np.load(os.path.join(ROOT, "tests", "golden", "ofdm_config_64.npz"))


def qpsk_symbols(rng, n):
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    return pts[rng.integers(0, 4, n)].astype(np.complex64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--range", type=float, default=10.0)
    ap.add_argument("--angle", type=float, default=20.0)
    ap.add_argument("--velocity", type=float, default=0.0)
    ap.add_argument("--rcs-dbsm", type=float, default=20.0)
    ap.add_argument("--packets", type=int, default=3)
    a = ap.parse_args()
    import jrc_amd
    fg = RadarSimFlowgraph(load_ofdm_config(), [a.range], [a.velocity], [a.rcs_dbsm], [a.angle])
    rng = np.random.default_rng(1)
    nbytes, mcs = 100, 2
    ns = jrc_amd.n_ofdm_sym(mcs, 48, nbytes)
    for i in range(a.packets):
        res, _ = fg.run_packet(qpsk_symbols(rng, ns * 48), mcs, DATA, nbytes)
        print("packet %d: range %.2f m, angle %.2f deg, snr %.1f dB, published %d"
              % (i, res.range_val, res.angle_val, res.snr_est, res.published))


if __name__ == "__main__":
    main()
