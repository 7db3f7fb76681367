#!/usr/bin/env python3
"""The radar simulation flowgraph of the reference (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2165-2232) as ONE device-resident,
frame-batched leg — what SURVEY.md §8(f) ranks 1-2 were built for: data symbols in HBM -> records in HBM, no host hop in between:

  d_symbols [F][n_sym * N_data]
    -> jrc_precoder_frames_dev        mimo_precoder, all F packets in one launch          -> [F][T][n_total][N]      (also the radar's TX reference)
    -> jrc_ofdm_mod_dev               fft_vxx(reverse, shift, window) + cyclic prefixer;  the window carries the blocks_multiply_const (tx_multiplier)
    -> jrc_zero_pad_strided_dev       zero_pad(0, 3 symbols), one launch per TX port      -> [T][F][n_burst]
    -> jrc_tsim_run_sum_dev           the T target_simulators and the blocks_add_xx per RX behind them, summed on the spectrum -> [F][R][n_burst]
                                      (jrc_tsim_run_dev per TX with accumulate_out where the burst length has no direct route)
    -> jrc_chain_run_td_dev           A6 + A7 + A1 as one kernel, then A2..A5             -> channel estimate, map, records

The analog noise sources of the .grc are not part of this leg (a stock block without a device counterpart here); examples/radar_sim_flowgraph.py
run with the pads of this leg and no noise sources is the block-by-block path it is compared with (tests/test_gpu_flowgraph_parity.py).
torch only allocates; every step is a C-ABI call on the context's stream.

  python examples/radar_sim_device_resident.py [--frames 64]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class DeviceResidentRadarSim:
    def __init__(self, tables, fft_len, N_rx, n_data_symbols, N_sym_radar, max_frames, trgt_range=(10.0,), trgt_velocity=(0.0,),
                 trgt_rcs_dbsm=(20.0,), trgt_angle=(20.0,), samp_rate=125_000_000, rf_freq=24e9, tx_multiplier=0.1, Ir=8, Ia=16,
                 mcs=2, packet_type=2, seed=0, ctx=None):
        import torch
        import jrc_amd as jrc
        self.jrc, self.torch = jrc, torch
        o = tables
        self.ctx = ctx or jrc.Context(0)
        N, cp = int(fft_len), int(fft_len) // 4
        T, R, F = int(o["N_tx"]), int(N_rx), int(max_frames)
        self.N, self.cp, self.T, self.R, self.F = N, cp, T, R, F
        self.nd = len(o["data_subcarriers"])
        self.n_data, self.mcs, self.ptype = int(n_data_symbols), mcs, packet_type
        self.pdu_len = (self.n_data * self.nd - 22) // 8 if mcs == 2 else None
        self.n_sync = len(o["l_stf_ltf_64"])
        self.n_total = self.n_sync + 1 + T + self.n_data
        self.N_pre, self.S = self.n_sync + 1, int(N_sym_radar)
        assert self.N_pre + self.S <= self.n_total
        self.pad_tail = 3 * (N + cp)
        self.n_in = self.n_total * (N + cp)
        self.n_burst = self.n_in + self.pad_tail
        self.seed = seed
        self.sum_on_spectrum = os.environ.get("JRC_DRF_SUM_ON_SPECTRUM", "1") != "0"
        # the OFDM modulator and the T zero_pads as one kernel (jrc_ofdm_mod_pad_dev, round 6): bit-identical bursts, the unpadded time-domain packet
        # is never written.  Opt-in until it has been timed on a device (round 6 had no GPU); off = the 1 + T launches of rounds 4-5.
        self.fused_mod = os.environ.get("JRC_DRF_FUSED_MOD", "0") != "0"
        P = T * R
        self.precoder = jrc.mimo_precoder(N, T, 1, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"], o["l_stf_ltf_64"],
                                          o["ltf_mapped_sc__ss_sym"], ctx=self.ctx)
        wavelength = 3e8 / rf_freq
        self.TX_RXs = [[(1 + t / 2 + 2 * r) * wavelength for r in range(R)] for t in range(T)]
        rcs = [10 ** (d / 10.0) for d in trgt_rcs_dbsm]
        self.sims = [jrc.target_simulator(trgt_range, trgt_velocity, rcs, trgt_angle, self.TX_RXs[t], samp_rate, rf_freq, -40.0, False, False,
                                          sum_targets=True, max_bursts=F, ctx=self.ctx) for t in range(T)]
        self.range_bins = np.linspace(0, 3e8 * N / (2 * samp_rate), N * Ir).astype(np.float32)
        self.angle_bins = (np.arcsin(2 / (P * Ia) * (np.arange(0, P * Ia) - np.floor(P * Ia / 2) + 0.5)) * 180 / np.pi).astype(np.float32)
        angle_res = float(np.rad2deg(np.arcsin(2 / P))) if P > 2 else 15.0
        self.chain = jrc.RadarChain(N, T, R, self.S, self.N_pre, Ir, Ia, self.range_bins, self.angle_bins, 2 * 3e8 / (2 * samp_rate), 2 * angle_res,
                                    15.0, 0.0, n_items=self.n_total, max_frames=F, ctx=self.ctx)
        dev = "cuda:%d" % self.ctx.device
        self.bufs = self.chain.alloc(F, dev)
        f32 = torch.float32
        self.d_sym = torch.zeros((F, self.n_data * self.nd, 2), dtype=f32, device=dev)
        self.d_txf = torch.zeros((F, T, self.n_total, N, 2), dtype=f32, device=dev)
        self.d_txt = torch.zeros((F, T, self.n_in, 2), dtype=f32, device=dev)
        self.d_pad = torch.zeros((T, F, self.n_burst, 2), dtype=f32, device=dev)
        self.d_rx = torch.zeros((F, R, self.n_burst, 2), dtype=f32, device=dev)
        w = np.full(N, np.float32(1 / N ** 0.5) * np.float32(tx_multiplier), np.float32)      # fft window x blocks_multiply_const
        self.d_window = torch.from_numpy(w).to(dev)
        L = self.ctx.lib
        L.jrc_zero_pad_strided_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
        L.jrc_ofdm_mod_pad_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p]
        torch.cuda.synchronize()

    def load_symbols(self, symbols):
        """symbols: complex64 [F][n_data * N_data] (host) -> HBM"""
        s = np.ascontiguousarray(symbols, np.complex64).reshape(-1, self.n_data * self.nd)
        self.d_sym[:len(s)].copy_(self.torch.from_numpy(s.view(np.float32).reshape(len(s), -1, 2)))
        self.torch.cuda.synchronize()
        return len(s)

    def step(self, n_frames, pads=True):
        """one pass of the whole graph over n_frames packets; asynchronous on the context's stream"""
        c, L, T = self.ctx, self.ctx.lib, self.T
        self.precoder.frames_dev(self.d_sym[:n_frames], self.mcs, self.ptype, self.pdu_len, d_out=self.d_txf)
        if self.fused_mod:
            c.check(L.jrc_ofdm_mod_pad_dev(c.h, self.N, self.cp, self.d_window.data_ptr(), n_frames, T, self.n_total, 0, self.pad_tail if pads else 0,
                                           self.seed, 100, self.d_txf.data_ptr(), self.d_pad.data_ptr(), self.F * self.n_burst, self.n_burst, None))
        else:
            c.check(L.jrc_ofdm_mod_dev(c.h, self.N, self.cp, self.d_window.data_ptr(), n_frames * T * self.n_total, self.d_txf.data_ptr(),
                                       self.d_txt.data_ptr(), None))
        for t in range(T):
            if not self.fused_mod:
                src = self.d_txt.data_ptr() + 8 * t * self.n_in
                c.check(L.jrc_zero_pad_strided_dev(c.h, n_frames, self.n_in, 0, self.pad_tail if pads else 0, self.seed + 100 * t, src, T * self.n_in,
                                                   self.d_pad[t].data_ptr(), self.n_burst, None))
            if not self.sum_on_spectrum:
                self.sims[t].run_dev(self.d_pad[t], self.d_rx, n_frames, self.n_burst, accumulate_out=(t > 0))
        if self.sum_on_spectrum:
            # the T simulators and the blocks_add_xx behind them in one pass (jrc_tsim_run_sum_dev): one inverse transform per RX antenna
            try:
                self.jrc.target_simulator.run_sum_dev(self.sims, [self.d_pad[t] for t in range(T)], self.d_rx, n_frames, self.n_burst)
            except self.jrc.JrcError as e:
                if e.status != self.jrc.JRC_ERR_UNSUPPORTED:
                    raise
                self.sum_on_spectrum = False                                         # a burst length outside the direct route: one by one
                for t in range(T):
                    self.sims[t].run_dev(self.d_pad[t], self.d_rx, n_frames, self.n_burst, accumulate_out=(t > 0))
        self.chain.run_td(self.bufs, self.d_txf, self.d_rx, n_frames, self.cp)

    def results(self, n_frames):
        return self.chain.results(self.bufs, n_frames)

    def edges(self, n_frames):
        """the tensors on the block edges, on the host (for the comparison with the block-by-block graph)"""
        self.ctx.sync()
        cx = lambda t: t.cpu().numpy().view(np.complex64)[..., 0]
        e = dict(tx_f=cx(self.d_txf[:n_frames]), tx_t=cx(self.d_txt[:n_frames]), bursts=np.swapaxes(cx(self.d_pad[:, :n_frames]), 0, 1),
                 rx_t=cx(self.d_rx[:n_frames]), H=cx(self.bufs["chanest"][:n_frames]), map=cx(self.bufs["map"][:n_frames]))
        if self.fused_mod:
            del e["tx_t"]                       # the unpadded time-domain packet does not exist on this path
        return e


def config_b_tables(T=4, N=256):
    """the documented 256-carrier generalisation of the reference's 64-carrier tables (tests/test_gpu_comm.py config_c_tables)"""
    from jrc_amd import synth
    rng = np.random.default_rng(0)
    guard = N // 16
    act = [c for c in range(-N // 2 + guard, N // 2 - guard + 1) if c != 0]
    pilots = [c for c in act if c % 32 == 16][:8]
    data = [c for c in act if c not in pilots]
    ltf = np.zeros(N, np.complex64)
    ltf[np.array(act) + N // 2] = rng.choice([-1.0, 1.0], len(act))
    mapped = np.stack([(synth.hadamard(T) * ltf[sc]).reshape(-1) for sc in range(N)]).astype(np.complex64)
    pil = np.array([[1, 1, 1, -1, 1, 1, 1, -1], [-1, -1, -1, 1, -1, -1, -1, 1], [1, 1, 1, -1, 1, 1, 1, -1]], np.complex64)[:, :len(pilots)]
    return dict(N_tx=np.int32(T), data_subcarriers=np.array(data, np.int32), pilot_subcarriers=np.array(pilots, np.int32), pilot_symbols=pil,
                l_stf_ltf_64=np.stack([ltf, ltf, ltf, ltf]), ltf_64=ltf, ltf_mapped_sc__ss_sym=mapped)


def qpsk_symbols(rng, n):
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    return pts[rng.integers(0, 4, n)].astype(np.complex64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    a = ap.parse_args()
    o = config_b_tables()
    sim = DeviceResidentRadarSim(o, 256, 4, 60, 64, a.frames)
    rng = np.random.default_rng(1)
    sim.load_symbols(np.stack([qpsk_symbols(rng, 60 * sim.nd) for _ in range(a.frames)]))
    sim.step(a.frames)
    sim.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        sim.step(a.frames)
    sim.ctx.sync()
    el = (time.perf_counter() - t0) / 10
    r = sim.results(a.frames)[0]
    print("%d packets per pass: %.3f ms, %.1f k packets/s; packet 0: range %.2f m, angle %.2f deg, snr %.1f dB"
          % (a.frames, el * 1e3, a.frames / el / 1e3, r.range_val, r.angle_val, r.snr_est))


if __name__ == "__main__":
    main()
