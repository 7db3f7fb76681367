#!/usr/bin/env python3
"""The radar simulation flowgraph of the reference (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2165-2232)
wired block by block over this package's MI355X blocks — every block between the stream encoder and the GUI sinks:

  symbols -> mimo_precoder -> per TX: fft_vxx(reverse, shift, window 1/sqrt(N)) + cyclic prefixer -> x tx_multiplier
          -> zero_pad(3 symbols) -> target_simulator per TX (R outputs each) -> sum over TX + noise
          -> per RX: ofdm_cyclic_prefix_remover -> fft_vxx(forward, shift)
          -> mimo_ofdm_radar(N_pre = 5, N_sym = N_tx) -> fft_vxx(reverse, N*Ir) -> matrix_transpose
          -> fft_vxx(forward, shift, P*Ia) -> range_angle_estimator

Variable names and values follow the .grc (fft_len 64, cp 16, N_rx 2, samp_rate 125 MHz, rf_freq 24 GHz, interp 8 / 16,
tx_multiplier 0.1, noise figure 10 dB, snr_threshold 15).  Needs a GPU: there is no CPU path behind these blocks.

  python examples/radar_sim_flowgraph.py [--range 10 --angle 20 --velocity 0 --rcs-dbsm 20]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NDP, DATA = 1, 2


class RadarSimFlowgraph:
    def __init__(self, ofdm_config, trgt_range=(10.0,), trgt_velocity=(0.0,), trgt_rcs_dbsm=(20.0,), trgt_angle=(0.0,),
                 N_rx=2, samp_rate=125_000_000, freq=4e9, noise_figure_dB=10.0, tx_multiplier=0.1, interp_factor_range=8,
                 interp_factor_angle=16, sum_targets=True, seed=0, ctx=None):
        import jrc_amd as jrc
        self.jrc, o = jrc, ofdm_config
        self.ctx = ctx or jrc.Context(0)
        self.fft_len, self.cp_len = 64, 16
        self.N_tx, self.N_rx = int(o["N_tx"]), N_rx
        self.samp_rate, self.rf_freq = samp_rate, freq + 20e9
        self.tx_multiplier = tx_multiplier
        self.noise_var = 4.003886160000000e-21 * samp_rate * 10 ** (noise_figure_dB / 10.0)
        self.Ir, self.Ia = interp_factor_range, interp_factor_angle
        self.rng = np.random.default_rng(seed)
        wavelength = 3e8 / self.rf_freq
        T, R, N = self.N_tx, N_rx, self.fft_len
        # TX1_RXs .. TX4_RXs of the .grc, extended to N_rx receivers 2 wavelengths apart
        self.TX_RXs = [[(1 + t / 2 + 2 * r) * wavelength for r in range(R)] for t in range(T)]
        self.n_sync = len(o["l_stf_ltf_64"])
        self.precoder = jrc.mimo_precoder(N, T, 1, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                                          o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], ctx=self.ctx)
        rcs = [10 ** (d / 10.0) for d in trgt_rcs_dbsm]
        self.target_sims = [jrc.target_simulator(trgt_range, trgt_velocity, rcs, trgt_angle, self.TX_RXs[t], samp_rate,
                                                 self.rf_freq, -40.0, False, False, sum_targets=sum_targets, ctx=self.ctx)
                            for t in range(T)]
        self.zero_pads = [jrc.zero_pad(False, 0, 3 * (N + self.cp_len), seed=seed + 100 * t, ctx=self.ctx) for t in range(T)]
        self.cp_remover = jrc.ofdm_cyclic_prefix_remover(N, self.cp_len, ctx=self.ctx)
        self.radar = jrc.mimo_ofdm_radar(N, T, R, T, self.n_sync + 1, False, False, 8, self.Ir, False, "", ctx=self.ctx)
        self.range_ifft = jrc.fft_vcc(N * self.Ir, False, None, False, ctx=self.ctx)
        self.transpose = jrc.matrix_transpose(N * self.Ir, T * R, self.Ia, ctx=self.ctx)
        self.angle_fft = jrc.fft_vcc(T * R * self.Ia, True, None, True, ctx=self.ctx)
        P = T * R
        self.range_bins = np.linspace(0, 3e8 * N / (2 * samp_rate), N * self.Ir).astype(np.float32)
        self.angle_bins = (np.arcsin(2 / (P * self.Ia) * (np.arange(0, P * self.Ia) - np.floor(P * self.Ia / 2) + 0.5))
                           * 180 / np.pi).astype(np.float32)
        R_res = 3e8 / (2 * samp_rate)
        angle_res = float(np.rad2deg(np.arcsin(2 / P)))
        self.estimator = jrc.range_angle_estimator(P * self.Ia, self.range_bins, self.angle_bins, R_res * 2, angle_res * 2,
                                                   15.0, 0.0, "", False, ctx=self.ctx)

    def run_packet(self, symbols, mcs=2, packet_type=DATA, pdu_len=None, noise=True):
        """one PDU through the whole graph; returns the estimator result and the tensors on the block edges"""
        jrc, N, cp, T, R = self.jrc, self.fft_len, self.cp_len, self.N_tx, self.N_rx
        tx_f = self.precoder.work(symbols, mcs, packet_type, pdu_len)                 # [T][n_total][N]
        n_total = tx_f.shape[1]
        window = np.full(N, 1 / N ** 0.5, np.float32)
        rx_t = np.zeros((R, (n_total + 3) * (N + cp)), np.complex64)
        for t in range(T):
            td = jrc.ofdm_mod(tx_f[t], N, cp, window=window, ctx=self.ctx).ravel() * np.float32(self.tx_multiplier)
            burst = self.zero_pads[t].work(td) if noise else np.concatenate([td, np.zeros(3 * (N + cp), np.complex64)])   # zero_pad(0, 3 symbols)
            rx_t += self.target_sims[t].work(burst)                                   # blocks_add_xx
        if noise:
            s = np.sqrt(self.noise_var / 2)
            rx_t = rx_t + (s * (self.rng.standard_normal(rx_t.shape) + 1j * self.rng.standard_normal(rx_t.shape))).astype(np.complex64)
        rx_f = [self.cp_remover.work(rx_t[r], fused_fft=True)[:n_total] for r in range(R)]
        H = self.radar.general_work([tx_f[t] for t in range(T)], rx_f)                # [P][N*Ir]
        rng_prof = self.range_ifft.work(H)
        m = self.angle_fft.work(self.transpose.work(rng_prof))                        # [N*Ir][P*Ia]
        res = self.estimator.work(m)
        return res, dict(tx_f=tx_f, rx_t=rx_t, rx_f=np.stack(rx_f), H=H, map=m)


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
