"""ctypes driver for gr-mimo-ofdm-jrc_amd/lib/libjrc_blocks.so (the C++ host-side blocks + their C test harness)."""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("JRC_BLOCKS_LIB_PATH") or os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd", "lib", "libjrc_blocks.so")   # (the override: tests/hipcpu emulation mode)
_vp, _fp, _ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
_lib = None


def lib():
    global _lib
    if _lib is None:
        import jrc_amd
        jrc_amd.load(build_if_missing=True)           # torch first, then libjrc_hip.so (one HIP runtime)
        if not os.path.exists(LIB):
            jrc_amd._build.build_host()
        L = C.CDLL(LIB)
        L.jrcb_last_error.restype = C.c_char_p
        for name in ("jrcb_make_radar", "jrcb_make_radar2", "jrcb_make_radar_chain", "jrcb_make_radar_chain_bg", "jrcb_make_transpose", "jrcb_make_estimator", "jrcb_make_cp_remover",
                     "jrcb_make_peak_detect", "jrcb_make_equalizer", "jrcb_make_precoder", "jrcb_make_target_simulator", "jrcb_make_stream_encoder", "jrcb_make_stream_decoder",
                     "jrcb_make_moving_avg", "jrcb_make_frame_detector", "jrcb_make_frame_sync", "jrcb_make_zero_pad", "jrcb_make_frame_generator"):
            getattr(L, name).restype = _vp
        L.jrcb_make_radar.argtypes = [C.c_int] * 10
        L.jrcb_make_radar2.argtypes = [C.c_int] * 10 + [C.c_char_p]
        L.jrcb_make_transpose.argtypes = [C.c_int] * 3
        L.jrcb_make_radar_chain.argtypes = [C.c_int] * 8 + [_fp, C.c_int, _fp, C.c_int] + [C.c_float] * 4 + [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.jrcb_make_radar_chain_bg.argtypes = L.jrcb_make_radar_chain.argtypes + [C.c_int] * 3
        L.jrcb_make_estimator.argtypes = [C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_char_p, C.c_int]
        L.jrcb_make_cp_remover.argtypes = [C.c_int, C.c_int]
        L.jrcb_make_peak_detect.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int]
        L.jrcb_make_equalizer.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _ip, C.c_int, _ip, C.c_int, _fp, C.c_int,
                                          _fp, _fp, C.c_int, C.c_int, C.c_char_p]
        L.jrcb_make_precoder.argtypes = [C.c_int, C.c_int, _ip, C.c_int, _ip, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_char_p,
                                         C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.jrcb_make_target_simulator.argtypes = [_fp, _fp, _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int]
        L.jrcb_make_stream_encoder.argtypes = [C.c_int, C.c_int]
        L.jrcb_make_moving_avg.argtypes = [C.c_int, C.c_float, C.c_int]
        L.jrcb_make_zero_pad.argtypes = [C.c_uint, C.c_uint]
        L.jrcb_make_frame_generator.argtypes = [C.c_int, C.c_int, _ip, _ip, C.c_int, _ip, _ip, C.c_int, _ip, _fp, C.c_int, _fp, C.c_int]
        L.jrcb_make_frame_detector.argtypes = [C.c_int, C.c_int, C.c_double, C.c_uint, C.c_uint]
        L.jrcb_make_frame_sync.argtypes = [C.c_int, C.c_int, C.c_uint, _fp, C.c_int]
        L.jrcb_make_stream_decoder.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.jrcb_post_msg.argtypes = [_vp, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.jrcb_add_stream_start.argtypes = [_vp, C.c_int, C.c_uint64, C.c_long, C.c_long, C.c_long, C.c_double]
        L.jrcb_add_stream_end.argtypes = [_vp, C.c_int, C.c_uint64, C.c_double, _fp, C.c_int]
        L.jrcb_destroy.argtypes = [_vp]
        L.jrcb_add_in_tag.argtypes = [_vp, C.c_int, C.c_uint64, C.c_char_p, C.c_int, C.c_long, C.c_double]
        L.jrcb_run.argtypes = [_vp, C.c_int, _ip, C.c_int, C.POINTER(_vp), C.c_int, C.POINTER(_vp)]
        L.jrcb_consumed.argtypes = [_vp, C.c_int]
        L.jrcb_state_json.argtypes = [_vp, C.c_char_p, C.c_int]
        L.jrcb_call_setter.argtypes = [_vp, C.c_char_p, C.c_double]
        _lib = L
    return _lib


def _c(a):
    return np.ascontiguousarray(a, np.complex64)


def _f(a):
    return a.ctypes.data_as(_fp)


class Block:
    def __init__(self, handle):
        if not handle:
            msg = lib().jrcb_last_error().decode()
            raise (ValueError if "invalid_argument" in msg else RuntimeError)(msg)    # the exception type make() threw
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None):
            lib().jrcb_destroy(self.h)
            self.h = None

    def tag(self, port, offset, key, value):
        if isinstance(value, float):
            lib().jrcb_add_in_tag(self.h, port, offset, key.encode(), 2, 0, value)
        else:
            lib().jrcb_add_in_tag(self.h, port, offset, key.encode(), 0, int(value), 0.0)

    def run(self, noutput_items, ins, outs):
        """ins / outs: lists of numpy arrays (already sized); returns items produced.  Raises like the block would."""
        nin = (C.c_int * max(1, len(ins)))(*[getattr(a, "_nitems", len(a)) for a in ins])
        pin = (_vp * max(1, len(ins)))(*[a.ctypes.data for a in ins])
        pout = (_vp * max(1, len(outs)))(*[a.ctypes.data for a in outs])
        n = lib().jrcb_run(self.h, noutput_items, nin, len(ins), pin, len(outs), pout)
        if n == -1001:
            raise ValueError(lib().jrcb_last_error().decode())
        if n == -1000:
            raise RuntimeError(lib().jrcb_last_error().decode())
        return n

    def post(self, port, data, kind=1):
        """deliver a message: kind 0 = pmt symbol, 1 = PDU pair (dict . blob), 2 = some other pmt"""
        data = bytes(data)
        lib().jrcb_post_msg(self.h, port.encode(), kind, data, len(data))

    def stream_start(self, offset, data_bytes, mcs, packet_type, snr, port=0):
        lib().jrcb_add_stream_start(self.h, port, offset, data_bytes, mcs, packet_type, snr)

    def stream_end(self, offset, snr_data, chan_mean=(), port=0):
        cm = np.ascontiguousarray(chan_mean, np.complex64)
        lib().jrcb_add_stream_end(self.h, port, offset, snr_data, cm.view(np.float32).ctypes.data_as(_fp) if cm.size else None, cm.size)

    def consumed(self, port):
        return lib().jrcb_consumed(self.h, port)

    def state(self):
        buf = C.create_string_buffer(1 << 20)
        n = lib().jrcb_state_json(self.h, buf, len(buf))
        assert n > 0
        return json.loads(buf.value.decode())

    def query(self, name):
        """integer properties of the block behind the setter entry point (n_devices, frames_done)"""
        return lib().jrcb_call_setter(self.h, name.encode(), 0.0)

    def set(self, name, v):
        r = lib().jrcb_call_setter(self.h, name.encode(), float(v))
        if r == -1000:
            raise RuntimeError(lib().jrcb_last_error().decode())
        assert r == 0, name


def radar(fft_len, N_tx, N_rx, N_sym, N_pre, bg_removal=False, bg_recording=False, record_len=8, interp=1, interleave=False, radar_chan_file=""):
    return Block(lib().jrcb_make_radar2(fft_len, N_tx, N_rx, N_sym, N_pre, int(bg_removal), int(bg_recording), record_len, interp, int(interleave),
                                        radar_chan_file.encode()))


def radar_chain(fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, rb, ab, ndr, nda, snr_thr, pow_thr, stats_path="",
                stats_record=False, interleave=False, frames_per_batch=16, batches_in_flight=3, bg_removal=False, bg_recording=False, record_len=0):
    rb = np.ascontiguousarray(rb, np.float32)
    ab = np.ascontiguousarray(ab, np.float32)
    if bg_removal or bg_recording:
        return Block(lib().jrcb_make_radar_chain_bg(fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, int(interleave), _f(rb), len(rb),
                                                    _f(ab), len(ab), ndr, nda, snr_thr, pow_thr, stats_path.encode(), int(stats_record),
                                                    frames_per_batch, batches_in_flight, int(bg_removal), int(bg_recording), record_len))
    return Block(lib().jrcb_make_radar_chain(fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, int(interleave), _f(rb), len(rb),
                                             _f(ab), len(ab), ndr, nda, snr_thr, pow_thr, stats_path.encode(), int(stats_record),
                                             frames_per_batch, batches_in_flight))


def transpose(input_len, output_len, interp):
    return Block(lib().jrcb_make_transpose(input_len, output_len, interp))


def estimator(vlen, rb, ab, ndr, nda, snr_thr, pow_thr, stats_path="", stats_record=False):
    rb = np.ascontiguousarray(rb, np.float32)
    ab = np.ascontiguousarray(ab, np.float32)
    return Block(lib().jrcb_make_estimator(vlen, _f(rb), len(rb), _f(ab), len(ab), ndr, nda, snr_thr, pow_thr, stats_path.encode(), int(stats_record)))


def cp_remover(fft_len, cp_len):
    return Block(lib().jrcb_make_cp_remover(fft_len, cp_len))


def target_simulator(range_m, velocity, rcs, azimuth, position_rx, samp_rate, center_freq, self_coupling_db=-40.0,
                     rndm_phaseshift=False, self_coupling=False):
    r, v, s, a, p = (np.ascontiguousarray(np.atleast_1d(x), np.float32) for x in (range_m, velocity, rcs, azimuth, position_rx))
    if not (v.size == s.size == a.size == r.size):       # the C harness takes one count; mismatches are tested in Python
        raise ValueError("length mismatch")
    return Block(lib().jrcb_make_target_simulator(_f(r), _f(v), _f(s), _f(a), r.size, _f(p), p.size, int(samp_rate),
                                                  float(center_freq), float(self_coupling_db), int(rndm_phaseshift), int(self_coupling)))


def frame_generator(fft_len, occupied_carriers, pilot_carriers, pilot_symbols, sync_words, shifted=True):
    def flat(sets, dt):
        sz = np.array([len(x) for x in sets], np.int32)
        fl = np.ascontiguousarray(np.concatenate([np.asarray(x, dt).ravel() for x in sets] + [np.zeros(1, dt)]), dt)
        return sz, fl
    osz, ofl = flat(occupied_carriers, np.int32)
    psz, pfl = flat(pilot_carriers, np.int32)
    ssz, sfl = flat(pilot_symbols, np.complex64)
    sw = np.ascontiguousarray(np.concatenate([np.asarray(sync_words, np.complex64).ravel(), np.zeros(1, np.complex64)]))
    return Block(lib().jrcb_make_frame_generator(fft_len, len(osz), osz.ctypes.data_as(_ip), ofl.ctypes.data_as(_ip), len(psz), psz.ctypes.data_as(_ip),
                                                 pfl.ctypes.data_as(_ip), len(ssz), ssz.ctypes.data_as(_ip), sfl.view(np.float32).ctypes.data_as(_fp),
                                                 len(sync_words), sw.view(np.float32).ctypes.data_as(_fp), int(shifted)))


def zero_pad(pad_front, pad_tail):
    return Block(lib().jrcb_make_zero_pad(pad_front, pad_tail))


def moving_avg(length, scale, max_iter):
    return Block(lib().jrcb_make_moving_avg(length, scale, max_iter))


def frame_detector(fft_len, cp_len, threshold, min_n_peaks, ignore_gap):
    return Block(lib().jrcb_make_frame_detector(fft_len, cp_len, threshold, min_n_peaks, ignore_gap))


def frame_sync(fft_len, cp_len, sync_length, ltf_seq_time):
    t = _c(ltf_seq_time)
    return Block(lib().jrcb_make_frame_sync(fft_len, cp_len, sync_length, t.view(np.float32).ctypes.data_as(_fp), t.size))


def stream_encoder(mcs, data_len):
    return Block(lib().jrcb_make_stream_encoder(mcs, data_len))


def stream_decoder(n_data_carriers, comm_log_file="", stats_record=False):
    return Block(lib().jrcb_make_stream_decoder(n_data_carriers, comm_log_file.encode(), int(stats_record)))


def peak_detect(samp_rate, interp, threshold, samp_protect):
    return Block(lib().jrcb_make_peak_detect(samp_rate, interp, threshold, samp_protect))


def equalizer(o, algo=0, freq=24e9, bw=125e6, n_ltf=4, chan_est_file=""):
    dc = np.ascontiguousarray(o["data_subcarriers"], np.int32)
    pc = np.ascontiguousarray(o["pilot_subcarriers"], np.int32)
    ps, ltf, ml = _c(o["pilot_symbols"]), _c(o["ltf_64"]), _c(o["ltf_mapped_sc__ss_sym"])
    return Block(lib().jrcb_make_equalizer(algo, freq, bw, 64, 16, dc.ctypes.data_as(_ip), len(dc), pc.ctypes.data_as(_ip), len(pc),
                                           _f(ps.view(np.float32)), ps.shape[0], _f(ltf.view(np.float32)), _f(ml.view(np.float32)),
                                           ml.shape[1], n_ltf, chan_est_file.encode()))


def precoder(o, T=4, chan_est_file="", smoothing=False, radar_log_file="", radar_aided=False, phased=False, radar_streams=False,
             dc=None, pc=None):
    dc = np.ascontiguousarray(o["data_subcarriers"] if dc is None else dc, np.int32)
    pc = np.ascontiguousarray(o["pilot_subcarriers"] if pc is None else pc, np.int32)
    ps, sw, ml = _c(o["pilot_symbols"]), _c(o["l_stf_ltf_64"]), _c(o["ltf_mapped_sc__ss_sym"])
    return Block(lib().jrcb_make_precoder(64, T, dc.ctypes.data_as(_ip), len(dc), pc.ctypes.data_as(_ip), len(pc),
                                          _f(ps.view(np.float32)), ps.shape[0], _f(sw.view(np.float32)), sw.shape[0],
                                          _f(ml.view(np.float32)), chan_est_file.encode(), int(smoothing), radar_log_file.encode(),
                                          int(radar_aided), int(phased), int(radar_streams)))
