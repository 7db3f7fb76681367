"""Synthetic comm captures for the probes, made with the package's own blocks (stream_encoder -> mimo_precoder -> OFDM modulator),
then a flat 4x1 channel, carrier offset and noise in numpy.  No test-infrastructure imports."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, CP = 64, 16


def ofdm_config():
    return np.load(os.path.join(ROOT, "tests", "golden", "ofdm_config_64.npz"))


class BurstMaker:
    def __init__(self, ctx, mcs=2):
        import jrc_amd
        o = ofdm_config()
        self.jrc, self.ctx, self.mcs = jrc_amd, ctx, mcs
        self.enc = jrc_amd.stream_encoder(mcs, len(o["data_subcarriers"]), ctx=ctx)
        self.pre = jrc_amd.mimo_precoder(N, int(o["N_tx"]), 1, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                                         o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], ctx=ctx)
        self.h = np.array([1.0, 0.5j, -0.3, 0.2 + 0.1j], np.complex64)
        self.window = np.full(N, 1 / N ** 0.5, np.float32)

    def burst(self, payload, rng, lead=700, tail=3000, cfo=0.01, noise=0.02):
        sym, tags = self.enc.work(payload)
        tx_f = self.pre.work(sym, tags["mcs"], tags["packet_type"], tags["pdu_len"])
        tx_t = np.stack([self.jrc.ofdm_mod(tx_f[t], N, CP, window=self.window, ctx=self.ctx).ravel() for t in range(tx_f.shape[0])])
        frame = np.tensordot(self.h, tx_t, axes=(0, 0))
        x = np.concatenate([np.zeros(lead, np.complex64), frame, np.zeros(tail, np.complex64)])
        x = x * np.exp(1j * cfo * np.arange(x.size))
        x = x + noise * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))
        return x.astype(np.complex64)
