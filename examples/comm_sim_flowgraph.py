#!/usr/bin/env python3
"""The communication simulation flowgraph of the reference (examples/simulation/communication/mimo_ofdm_jrc_comm_sim.grc)
wired over this package's MI355X blocks, PDU bytes in -> PDU bytes out:

  PDU -> stream_encoder -> mimo_precoder -> per TX: fft_vxx(reverse, shift, window 1/sqrt(N)) + cyclic prefixer
      -> x tx_multiplier -> zero_pad(5, 6 symbols + 10) -> x line-of-sight gain of antenna k, (1/path_loss) e^{j k pi sin(theta)}
      -> blocks_add_xx -> channels_channel_model (carrier offset 0.02/N cycles per sample, noise sqrt(noise_var), one tap)
      -> delay / conjugate-multiply / moving averages / divide (detection metrics) -> frame_detector -> frame_sync
      -> fft_vxx(forward, shift) -> mimo_ofdm_equalizer -> stream_decoder -> PDU

Values follow the .grc (fft_len 64, cp 16, tx_multiplier 0.5, distance 20 m, theta 20 deg, noise figure 10 dB, rf 24 GHz,
threshold 0.6, min_n_peaks 10).  An NDP packet first makes the equalizer write the channel estimate; with steer=True the
precoder then beam-forms the DATA packets with the steering matrices derived from it (the chan_est.csv loop of the reference,
here passed in memory).  `channel="flat"` swaps the line-of-sight gains for a drawn flat 4x1 channel at a chosen SNR.

The wiring is written against a *block set* (`blocks=`, default: this package) like examples/radar_sim_flowgraph.py;
`send()` returns every block edge so that tests/test_gpu_flowgraph_parity.py can compare two block sets edge by edge.

  python examples/comm_sim_flowgraph.py [--mcs 3] [--packets 5] [--steer] [--flat --snr-db 25]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NDP, DATA, LS, STA = 1, 2, 0, 1


class CommSimFlowgraph:
    def __init__(self, ofdm_config, mcs=2, estimator=LS, seed=0, ctx=None, blocks=None, fft_len=64, cp_len=None, channel="flat",
                 samp_rate=125_000_000, freq=4e9, noise_figure_dB=10.0, tx_multiplier=0.5, distance=20.0, theta=20.0,
                 smoothing=True, phased_steering=False):
        if blocks is None:
            import jrc_amd as blocks
            ctx = ctx or blocks.Context(0)
        B = self.B = blocks
        o = ofdm_config
        self.ctx = ctx
        self.N = int(fft_len)
        self.cp = self.N // 4 if cp_len is None else int(cp_len)
        self.T = int(o["N_tx"])
        N, cp, T = self.N, self.cp, self.T
        self.n_dc = len(o["data_subcarriers"])
        self.rf_freq = freq + 20e9
        self.encoder = B.stream_encoder(mcs, self.n_dc, ctx=ctx)
        self.precoder = B.mimo_precoder(N, T, 1, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                                        o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], ctx=ctx)
        self.rx_fft = B.fft_vcc(N, True, None, True, ctx=ctx)
        self.ltf_fir = o["l_ltf_fir"]
        self.n_sync = len(o["l_stf_ltf_64"])
        self.sync_length = 4 * (N + cp)
        self.corr_window_size = N // 2
        self.ignore_gap = (self.n_sync + T) * (N + cp)
        self.pad_front, self.pad_tail = 5, 6 * (N + cp) + 10
        self.smoothing, self.phased = smoothing, phased_steering
        self.equalizer = B.mimo_ofdm_equalizer(estimator, self.rf_freq, samp_rate, N, cp, o["data_subcarriers"],
                                               o["pilot_subcarriers"], o["pilot_symbols"], o["l_stf_ltf_64"][3],
                                               o["ltf_mapped_sc__ss_sym"], T, ctx=ctx)
        self.decoder = B.stream_decoder(self.n_dc, ctx=ctx)
        self.zero_pads = [B.zero_pad(False, self.pad_front, self.pad_tail, seed=seed + 100 * t, ctx=ctx) for t in range(T)]
        self.rng = np.random.default_rng(seed)
        self.channel = channel
        self.tx_multiplier = tx_multiplier if channel == "los" else 1.0
        self.noise_var = 4e-21 * samp_rate * 10 ** (noise_figure_dB / 10.0)
        if channel == "los":                                                      # blocks_multiply_const_vxx_1*: antenna k of a lambda/2 array
            path_loss = 4 * np.pi * distance / (3e8 / self.rf_freq)
            self.h = np.array([(1 / path_loss) * np.exp(1j * k * np.pi * np.sin(np.deg2rad(theta))) for k in range(T)]).astype(np.complex64)
            self.cfo = 2 * np.pi * 0.02 / N                                        # channel_model freq_offset, cycles -> rad per sample
        else:
            self.h = ((self.rng.standard_normal(T) + 1j * self.rng.standard_normal(T)) / np.sqrt(2)).astype(np.complex64)
            self.cfo = 0.01
        self.chan_est = None

    def steering(self):
        """compute_steering_matrix() (lib/mimo_precoder_impl.cc:775-898): from the mean channel with smoothing, per subcarrier without"""
        if self.chan_est is None:
            return {}
        if self.smoothing:
            return dict(steer_mode=1, Q_mean=self.B.steering_from_channel(self.chan_est.mean(axis=0), self.phased, ctx=self.ctx))
        return dict(steer_mode=2, Q_sc=self.B.steering_from_channel(self.chan_est, self.phased, ctx=self.ctx))

    def radar_aided_steering(self, angle_estimate):
        """compute_radar_aided_steering() (lib/mimo_precoder_impl.cc:901-983): the user is where the radar saw its target — the last line of
        radar_log.csv carries the estimator's angle — so the channel is taken to be the array response h[t] = exp(j pi sin(angle) t)"""
        a = np.float32(angle_estimate)
        h = np.array([np.exp(1j * np.float32(np.pi * np.sin(float(a) / 180.0 * np.pi) * t)) for t in range(self.T)]).astype(np.complex64)
        return dict(steer_mode=1, Q_mean=self.B.steering_from_channel(h, self.phased, ctx=self.ctx))

    def send(self, pdu, snr_db=30.0, steer=False, cfo=None, lead=640, sources=None, force=None, radar_angle=None, radar_streams=None):
        """one PDU through the graph; returns (crc_ok, payload, info); info["edges"] holds every block edge.
        `sources` = {"pads": [T][2] (front, tail), "noise": [n]} replays the random sources of an earlier run;
        `force` = edges of another run that every block reads instead of what this graph computed upstream."""
        B, N, cp, T = self.B, self.N, self.cp, self.T
        cfo = self.cfo if cfo is None else cfo
        force = force or {}

        def use(name, val):
            return force[name] if name in force else val

        e = {}
        info = dict(edges=e)
        sym, tags = self.encoder.work(pdu)
        if sym is None:
            return None, None, info
        e["symbols"], e["encoder_tags"] = sym, dict(tags)
        sym = use("symbols", sym)
        kw = (self.radar_aided_steering(radar_angle) if radar_angle is not None else self.steering()) if steer else {}
        e["steering"] = kw.get("Q_mean", kw.get("Q_sc"))
        if radar_streams is not None:                                            # use_radar_streams: N_tx - 1 streams of radar symbols on the other columns of Q (:560-631)
            kw = dict(kw, radar_streams=radar_streams)
        e["tx_f"] = self.precoder.work(sym, tags["mcs"], tags["packet_type"], tags["pdu_len"], **kw)       # [T][n_total][N]
        tx_f = use("tx_f", e["tx_f"])
        window = np.full(N, 1 / N ** 0.5, np.float32)
        e["tx_t"] = np.stack([B.ofdm_mod(tx_f[t], N, cp, window=window, ctx=self.ctx).ravel() * np.float32(self.tx_multiplier)
                              for t in range(T)])
        tx_t = use("tx_t", e["tx_t"])
        padded = []
        for t in range(T):
            if self.channel != "los":
                padded.append(np.concatenate([np.zeros(lead, np.complex64), tx_t[t], np.zeros(2 * lead, np.complex64)]))
            elif sources is not None:
                padded.append(np.concatenate([sources["pads"][t][0], tx_t[t], sources["pads"][t][1]]).astype(np.complex64))
            else:
                padded.append(self.zero_pads[t].work(tx_t[t]))
        e["padded"] = np.stack(padded)
        e["pads"] = [(p[:self.pad_front], p[self.pad_front + tx_t.shape[1]:]) for p in padded] if self.channel == "los" else None
        padded = use("padded", e["padded"])
        rx = np.zeros(padded.shape[1], np.complex64)
        for t in range(T):                                                       # multiply_const per antenna, then blocks_add_xx in port order
            rx = rx + (padded[t] * self.h[t]).astype(np.complex64)
        if self.channel == "los":
            sigma = np.sqrt(self.noise_var)                                        # noise_voltage: per component
            lead_n = lead
        else:
            p_sig = float(np.mean(np.abs(rx[lead:lead + tx_t.shape[1]]) ** 2))
            sigma = np.sqrt(p_sig / 10 ** (snr_db / 10.0) / 2)
            lead_n = 0
        n = rx.size + lead_n
        nz = sources["noise"] if sources is not None else \
            (sigma * (self.rng.standard_normal(n) + 1j * self.rng.standard_normal(n))).astype(np.complex64)
        x = np.concatenate([np.zeros(lead_n, np.complex64), rx])                  # receiver running before the burst arrives
        x = (x * np.exp(1j * cfo * np.arange(x.size)).astype(np.complex64) + nz).astype(np.complex64)   # channel_model
        e["noise"], e["rx"] = nz, x
        x = use("rx", x)
        # sync front-end: the stock metric blocks, then frame_detector and frame_sync (fresh per capture here)
        e["metrics"] = B.sync_metrics(x, N // 4, self.corr_window_size, int(1.5 * self.corr_window_size), 1 / 1.5, ctx=self.ctx)
        xd, in_abs, in_cor = use("metrics", e["metrics"])
        seg, dtags = B.frame_detector(N, cp, 0.6, 10, self.ignore_gap, ctx=self.ctx).run(xd, in_abs, in_cor)
        e["detector_out"], e["detector_tags"] = seg, dtags
        seg, dtags = use("detector_out", seg), use("detector_tags", dtags)
        if not dtags:
            info["detected"] = False
            return False, b"", info
        delayed = np.concatenate([np.zeros(self.sync_length, np.complex64), seg])[:seg.size]     # blocks_delay(sync_length)
        sym_t, stags = B.frame_sync(N, cp, self.sync_length, self.ltf_fir, ctx=self.ctx).run(seg, delayed, dtags)
        e["sync_out"], e["sync_tags"] = sym_t, stags
        sym_t, stags = use("sync_out", sym_t), use("sync_tags", stags)
        if not stags:
            info.update(detected=True, synced=False)
            return False, b"", info
        sym_t = sym_t[:(sym_t.size // N) * N].reshape(-1, N)
        e["y"] = self.rx_fft.work(sym_t) / np.float32(N ** 0.5)                    # frame_sync hands over [LTF, LTF, SIG, MIMO-LTFs, data]
        y = use("y", e["y"])
        eq = self.equalizer.general_work(y, [(stags[0][0] // N, stags[0][1])])
        e["eq_out"], e["eq_events"], e["eq_consumed"], e["chan_est"] = eq["out"], eq["events"], eq["consumed"], eq["chan_est"]
        if eq["chan_est"] is not None:
            self.chan_est = use("chan_est", eq["chan_est"])                       # what an NDP writes to chan_est.csv
        eq_out, events = use("eq_out", eq["out"]), use("eq_events", eq["events"])
        starts = [ev for ev in events if ev["kind"] == 1]
        info["events"] = events
        if not starts:
            return False, b"", info
        info["start"] = starts[0]
        need = B.stream_n_ofdm_sym(starts[0]["mcs"], self.n_dc, starts[0]["data_bytes"])
        if need < 0 or len(eq_out) < need:                                        # a mis-decoded SIG announces more symbols than the frame
            return False, b"", info                                              # has: the block would wait for them forever
        ok, payload = self.decoder.work(eq_out, starts[0])
        e["crc_ok"], e["payload"] = ok, payload
        info.update(n_symbols=len(eq_out), per=self.decoder.per, coarse_cfo=dtags[0][1], cfo_tag=stags[0][1])
        return ok, payload, info


def load_ofdm_config():
    return np.load(os.path.join(ROOT, "tests", "golden", "ofdm_config_64.npz"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mcs", type=int, default=3)
    ap.add_argument("--snr-db", type=float, default=25.0)
    ap.add_argument("--packets", type=int, default=5)
    ap.add_argument("--steer", action="store_true")
    ap.add_argument("--flat", action="store_true", help="drawn flat channel at --snr-db instead of the .grc's line-of-sight point")
    a = ap.parse_args()
    fg = CommSimFlowgraph(load_ofdm_config(), mcs=a.mcs, channel="flat" if a.flat else "los")
    rng = np.random.default_rng(1)
    ok, _, info = fg.send(bytes([NDP]) + b"sounding", a.snr_db)
    print("NDP: crc ok %s, channel estimate %s" % (ok, "written" if fg.chan_est is not None else "missing"))
    for i in range(a.packets):
        pdu = bytes([DATA]) + rng.integers(0, 256, 200, dtype=np.uint8).tobytes()
        ok, payload, info = fg.send(pdu, a.snr_db, steer=a.steer)
        print("packet %d: crc ok %s, payload intact %s, SIG snr %.1f dB, CFO estimate %.4f rad/sample, PER %.0f %%"
              % (i, ok, payload == pdu, info["start"]["snr"] if "start" in info else float("nan"), info.get("cfo_tag", float("nan")),
                 info.get("per", 0.0)))


if __name__ == "__main__":
    main()
