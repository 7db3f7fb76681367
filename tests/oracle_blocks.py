"""TEST INFRASTRUCTURE: the CPU oracle's blocks behind the same names / signatures as the jrc_amd block classes, so that
examples/radar_sim_flowgraph.py and examples/comm_sim_flowgraph.py can wire the reference's flowgraphs over them exactly
as they wire them over the HIP blocks (`blocks=oracle_blocks`).  Only tests import this module (the oracle rule).

Lengths the tagged-stream blocks announce are the reference's calculate_output_stream_length() rules:
matrix_transpose lib/matrix_transpose_impl.cc:62-67, ofdm_cyclic_prefix_remover lib/ofdm_cyclic_prefix_remover_impl.cc:62-67,
mimo_precoder lib/mimo_precoder_impl.cc:265-272, zero_pad lib/zero_pad_impl.cc:55-60, target_simulator
lib/target_simulator_impl.cc:195-200.
"""
import numpy as np

import oracle

MAX_PAYLOAD_SIZE = 3100        # lib/utils.h:33 (stream_encoder_impl.cc:139)


def n_ofdm_sym(mcs, n_data_carriers, nbytes):
    return oracle.n_ofdm_sym(mcs, n_data_carriers, nbytes)


class Context:
    def __init__(self, device=0):
        pass


class mimo_precoder(oracle.Precoder):
    def __init__(self, fft_len, N_tx, N_ss, data_carriers, pilot_carriers, pilot_symbols, sync_words, mapped_ltf_symbols, ctx=None, **kw):
        super().__init__(fft_len, N_tx, N_ss, data_carriers, pilot_carriers, pilot_symbols, sync_words, mapped_ltf_symbols)


def steering_from_channel(h, phased=False, ctx=None):
    h = np.asarray(h)
    if h.ndim == 1:
        return oracle.steering_from_channel(h, phased)
    return np.stack([oracle.steering_from_channel(r, phased) for r in h])


def ofdm_mod(x, fft_len, cp_len, window=None, ctx=None):
    """fft_vxx(reverse, shift, window) then digital_ofdm_cyclic_prefixer(rolloff 0): the last cp_len samples in front"""
    t = oracle.fft_vcc(np.asarray(x, np.complex64).reshape(-1, fft_len), forward=False, shift=True, window=window)
    return np.concatenate([t[:, fft_len - cp_len:], t], axis=1)


class zero_pad:
    """the pad is a random draw (lib/zero_pad_impl.cc:76-90): the flowgraphs replay the HIP run's pads through `sources`,
    so this block is only constructed, never asked to draw"""

    def __init__(self, debug=False, pad_front=0, pad_tail=0, seed=0, ctx=None):
        self.pad_front, self.pad_tail = int(pad_front), int(pad_tail)

    def calculate_output_stream_length(self, ninput_items):
        return ninput_items + self.pad_front + self.pad_tail

    def work(self, x):
        raise RuntimeError("the oracle graph replays the pads of the run it is compared with")


class target_simulator:
    def __init__(self, range, velocity, rcs, azimuth, position_rx, samp_rate, center_freq, self_coupling_db=-40.0,
                 rndm_phaseshift=False, self_coupling=False, len_key="packet_len", debug=False, sum_targets=False, ctx=None, **kw):
        self._o = oracle.TargetSimulator(range, velocity, rcs, azimuth, position_rx, samp_rate, center_freq, self_coupling_db,
                                         rndm_phaseshift, self_coupling)
        self._sum = sum_targets

    def calculate_output_stream_length(self, ninput_items):
        return ninput_items

    def work(self, x, target_phase=None):
        return self._o.work(x, target_phase, self._sum)


class ofdm_cyclic_prefix_remover:
    def __init__(self, fft_len, cp_len, len_key="packet_len", ctx=None):
        self.fft_len, self.cp_len = fft_len, cp_len

    def calculate_output_stream_length(self, ninput_items):
        return ninput_items // (self.fft_len + self.cp_len)

    def work(self, x, fused_fft=False):
        y = oracle.cp_remove(x, self.fft_len, self.cp_len)
        return oracle.fft_vcc(y, forward=True, shift=True) if fused_fft else y     # the stock fft_vxx that follows the block


class fft_vcc:
    def __init__(self, fft_size, forward, window=None, shift=False, ctx=None):
        self.n, self.forward, self.shift = fft_size, bool(forward), bool(shift)
        self.window = None if window is None or len(window) == 0 else np.ascontiguousarray(window, np.float32)

    def work(self, x):
        return oracle.fft_vcc(x, self.forward, self.shift, self.window)


class mimo_ofdm_radar:
    def __init__(self, fft_len, N_tx, N_rx, N_sym, N_pre, background_removal=False, background_recording=False, record_len=8,
                 interp_factor=1, enable_tx_interleave=False, radar_chan_file="", len_tag_key="packet_len", debug=False, ctx=None):
        self._o = oracle.Radar(fft_len, N_tx, N_rx, N_sym, N_pre, background_removal, background_recording, record_len,
                               interp_factor, enable_tx_interleave)

    def general_work(self, tx, rx, tx_discard=0):
        return self._o.work(tx, rx, tx_discard)

    def set_background_record(self, background_record):
        self._o.set_background_record(background_record)


class matrix_transpose:
    def __init__(self, input_len, output_len, interp_factor, debug=False, len_key="packet_len", ctx=None):
        self.a = (input_len, output_len, interp_factor)

    def calculate_output_stream_length(self, ninput_items):
        return self.a[0]

    def work(self, x):
        return oracle.matrix_transpose(np.asarray(x).reshape(-1, self.a[0]), *self.a)


class range_angle_estimator:
    def __init__(self, vlen, range_bins, angle_bins, noise_discard_range_m, noise_discard_angle_deg, snr_threshold, power_threshold,
                 stats_path="", stats_record=False, len_key="packet_len", debug=False, ctx=None):
        self.vlen = vlen
        self.a = (np.asarray(range_bins, np.float32), np.asarray(angle_bins, np.float32), float(noise_discard_range_m),
                  float(noise_discard_angle_deg), float(snr_threshold), float(power_threshold))

    def work(self, m):
        return oracle.ra_estimate(np.asarray(m).reshape(-1, self.vlen), *self.a)


# ---- comm side ----------------------------------------------------------------------------------------------------------
class stream_encoder:
    def __init__(self, mod_encode, data_len, N_ss_radar=0, debug=False, ctx=None):
        self.mcs, self.data_len, self.d_scrambler = int(mod_encode), int(data_len), 1

    def work(self, pdu):
        p = np.frombuffer(bytes(pdu), np.uint8)
        if p.size + 4 > MAX_PAYLOAD_SIZE:
            return None, None
        out, tags = oracle.stream_encode(self.mcs, self.data_len, p, self.d_scrambler)
        self.d_scrambler = 1 if self.d_scrambler + 1 > 127 else self.d_scrambler + 1       # lib/stream_encoder_impl.cc:171-175
        return out, tags


class stream_decoder:
    def __init__(self, n_data_carriers, comm_log_file="", stats_record=False, debug=False, ctx=None):
        self.n_data_carriers = int(n_data_carriers)
        self._per = []

    @property
    def per(self):
        w = self._per[-25:]
        return 100.0 * (sum(w) / len(w)) if w else 0.0

    def work(self, symbols, stream_start):
        ok, payload = oracle.stream_decode(int(stream_start["mcs"]), self.n_data_carriers, int(stream_start["data_bytes"]), symbols)
        if ok is None:
            return None, None
        self._per.append(0 if ok else 1)
        return ok, payload


def stream_n_ofdm_sym(mcs, n_dc, data_bytes):
    if not 0 <= mcs <= 5 or data_bytes < 0:
        return -1
    return oracle.packet_params(mcs, n_dc, data_bytes)["n_ofdm_sym"]


def sync_metrics(x, delay, window, pwindow, pscale, ctx=None):
    return oracle.sync_metrics(x, delay, window, pwindow, pscale)


class frame_detector(oracle.FrameDetector):
    def __init__(self, fft_len, cp_len, threshold, min_n_peaks, ignore_gap, debug=False, ctx=None):
        super().__init__(fft_len, cp_len, threshold, min_n_peaks, ignore_gap)


class frame_sync(oracle.FrameSync):
    def __init__(self, fft_len, cp_len, sync_length, ltf_seq_time, debug=False, ctx=None):
        super().__init__(fft_len, cp_len, sync_length, ltf_seq_time)


class mimo_ofdm_equalizer(oracle.Equalizer):
    def __init__(self, estimator_algo, freq, bw, fft_len, cp_len, data_carriers, pilot_carriers, pilot_symbols, ltf_seq,
                 mapped_ltf_symbols, n_mimo_ltf, chan_est_file="", comm_log_file="", stats_record=False, debug=False, ctx=None):
        super().__init__(estimator_algo, freq, bw, fft_len, cp_len, data_carriers, pilot_carriers, pilot_symbols, ltf_seq,
                         mapped_ltf_symbols, n_mimo_ltf)
