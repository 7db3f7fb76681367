#!/usr/bin/env python3
"""The communication simulation flowgraph of the reference (examples/simulation/communication/mimo_ofdm_jrc_comm_sim.grc)
wired over this package's MI355X blocks, PDU bytes in -> PDU bytes out:

  PDU -> stream_encoder -> mimo_precoder -> per TX: fft_vxx(reverse, shift, window 1/sqrt(N)) + cyclic prefixer
      -> flat 4x1 MISO channel, carrier offset, noise (time domain, the burst somewhere inside a longer capture)
      -> delay / conjugate-multiply / moving averages / divide (detection metrics) -> frame_detector -> frame_sync
      -> fft_vxx(forward, shift) -> mimo_ofdm_equalizer -> stream_decoder -> PDU

An NDP packet first makes the equalizer write the channel estimate; with --steer the precoder then beam-forms the DATA
packets with the steering matrix derived from it (the chan_est.csv loop of the reference, here passed in memory).

  python examples/comm_sim_flowgraph.py [--mcs 3] [--snr-db 25] [--packets 5] [--steer]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NDP, DATA, LS, STA = 1, 2, 0, 1


class CommSimFlowgraph:
    def __init__(self, ofdm_config, mcs=2, estimator=LS, seed=0, ctx=None):
        import jrc_amd as jrc
        self.jrc, o = jrc, ofdm_config
        self.ctx = ctx or jrc.Context(0)
        self.N, self.cp, self.T = 64, 16, int(o["N_tx"])
        self.n_dc = len(o["data_subcarriers"])
        self.encoder = jrc.stream_encoder(mcs, self.n_dc, ctx=self.ctx)
        self.precoder = jrc.mimo_precoder(self.N, self.T, 1, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                                          o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], ctx=self.ctx)
        self.rx_fft = jrc.fft_vcc(self.N, True, None, True, ctx=self.ctx)
        self.ltf_fir = o["l_ltf_fir"]
        self.sync_length = 4 * (self.N + self.cp)
        self.equalizer = jrc.mimo_ofdm_equalizer(estimator, 24e9, 125e6, self.N, self.cp, o["data_subcarriers"],
                                                 o["pilot_subcarriers"], o["pilot_symbols"], o["ltf_64"],
                                                 o["ltf_mapped_sc__ss_sym"], self.T, ctx=self.ctx)
        self.decoder = jrc.stream_decoder(self.n_dc, ctx=self.ctx)
        self.rng = np.random.default_rng(seed)
        self.h = (self.rng.standard_normal(self.T) + 1j * self.rng.standard_normal(self.T)).astype(np.complex64) / np.sqrt(2)
        self.chan_est = None

    def send(self, pdu, snr_db=30.0, steer=False, cfo=0.01, lead=640):
        """one PDU through the graph; returns (crc_ok, payload, info)"""
        jrc, N, cp, T = self.jrc, self.N, self.cp, self.T
        sym, tags = self.encoder.work(pdu)
        if sym is None:
            return None, None, {}
        kw = {}
        if steer and self.chan_est is not None:                                   # compute_steering_matrix(): mean channel -> Q
            kw = dict(steer_mode=1, Q_mean=jrc.steering_from_channel(self.chan_est.mean(axis=0), ctx=self.ctx))
        tx_f = self.precoder.work(sym, tags["mcs"], tags["packet_type"], tags["pdu_len"], **kw)       # [T][n_total][N]
        window = np.full(N, 1 / N ** 0.5, np.float32)
        tx_t = np.stack([jrc.ofdm_mod(tx_f[t], N, cp, window=window, ctx=self.ctx).ravel() for t in range(T)])
        rx_t = np.tensordot(self.h, tx_t, axes=(0, 0))
        p_sig = float(np.mean(np.abs(rx_t) ** 2))
        sigma = np.sqrt(p_sig / 10 ** (snr_db / 10.0) / 2)
        x = np.concatenate([np.zeros(lead, np.complex64), rx_t, np.zeros(2 * lead, np.complex64)])
        x = x * np.exp(1j * cfo * np.arange(x.size))                               # carrier frequency offset, rad/sample
        x = (x + sigma * (self.rng.standard_normal(x.size) + 1j * self.rng.standard_normal(x.size))).astype(np.complex64)
        # sync front-end: the stock metric blocks, then frame_detector and frame_sync (fresh per capture here)
        xd, in_abs, in_cor = jrc.sync_metrics(x, N // 4, N // 2, int(1.5 * (N // 2)), 1 / 1.5, ctx=self.ctx)
        ignore_gap = (4 + T) * (N + cp)
        seg, dtags = jrc.frame_detector(N, cp, 0.6, 10, ignore_gap, ctx=self.ctx).run(xd, in_abs, in_cor)
        if not dtags:
            return False, b"", dict(detected=False)
        delayed = np.concatenate([np.zeros(self.sync_length, np.complex64), seg])[:seg.size]     # blocks_delay(sync_length)
        sym_t, stags = jrc.frame_sync(N, cp, self.sync_length, self.ltf_fir, ctx=self.ctx).run(seg, delayed, dtags)
        if not stags:
            return False, b"", dict(detected=True, synced=False)
        sym_t = sym_t[:(sym_t.size // N) * N].reshape(-1, N)
        y = self.rx_fft.work(sym_t) / np.float32(N ** 0.5)                         # frame_sync hands over [LTF, LTF, SIG, MIMO-LTFs, data]
        eq = self.equalizer.general_work(y, [(stags[0][0] // N, stags[0][1])])
        if eq["chan_est"] is not None:
            self.chan_est = eq["chan_est"]                                        # what an NDP writes to chan_est.csv
        starts = [e for e in eq["events"] if e["kind"] == 1]
        if not starts:
            return False, b"", dict(events=eq["events"])
        need = self.ctx.lib.jrc_stream_n_ofdm_sym(starts[0]["mcs"], self.n_dc, starts[0]["data_bytes"])
        if need < 0 or len(eq["out"]) < need:                                     # a mis-decoded SIG announces more symbols than the frame
            return False, b"", dict(start=starts[0], events=eq["events"])        # has: the block would wait for them forever
        ok, payload = self.decoder.work(eq["out"], starts[0])
        return ok, payload, dict(start=starts[0], events=eq["events"], n_symbols=len(eq["out"]), per=self.decoder.per,
                                 coarse_cfo=dtags[0][1], cfo_tag=stags[0][1])


def load_ofdm_config():
    return np.load(os.path.join(ROOT, "tests", "golden", "ofdm_config_64.npz"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mcs", type=int, default=3)
    ap.add_argument("--snr-db", type=float, default=25.0)
    ap.add_argument("--packets", type=int, default=5)
    ap.add_argument("--steer", action="store_true")
    a = ap.parse_args()
    fg = CommSimFlowgraph(load_ofdm_config(), mcs=a.mcs)
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
