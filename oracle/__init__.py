"""CPU oracle for the gr-mimo-ofdm-jrc hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference has no tests/golden vectors and cannot be built in this image
(GNU Radio 3.8 / Eigen3 / Boost / FFTW3f / VOLK are absent; no stand-ins are written for them).
The functions here restate the reference algorithm (file:line cited in oracle/jrc_oracle.c) and are
checked against closed forms and the constant tables minted from the reference's embedded Python
module (tests/golden/); the SIGNAL field, scrambler, convolutional encoder and CRC also against the bit-level
examples IEEE 802.11 publishes for them (tests/test_published_vectors.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libjrc_oracle.so")
_lib = None

c_float_p = C.POINTER(C.c_float)


class RaResult(C.Structure):
    _fields_ = [("peak_range_idx", C.c_int), ("peak_angle_idx", C.c_int),
                ("angle_null_idx", C.c_int),
                ("discard_range_idx", C.c_int), ("discard_angle_idx", C.c_int),
                ("n_noise_samples", C.c_int),
                ("peak_power", C.c_float), ("noise_power", C.c_float), ("snr_est", C.c_float),
                ("range_val", C.c_float), ("angle_val", C.c_float),
                ("published", C.c_int)]


def build(force=False):
    """Compile oracle/jrc_oracle*.c with gcc (the checker, not the product)."""
    srcs = [os.path.join(_HERE, f) for f in ("jrc_oracle.c", "jrc_oracle_comm.c", "jrc_oracle_tsim.c", "jrc_oracle_codec.c", "jrc_oracle_sync.c", "jrc_oracle.h", "Makefile")]
    def fresh():
        return os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)
    if not force and fresh():
        return _LIB_PATH
    import fcntl
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    with open(os.path.join(_HERE, "_build", ".build.lock"), "w") as fh:       # one builder at a time across processes
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            if force or not fresh():
                subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        L.orc_radar_create.restype = C.c_void_p
        L.orc_radar_create.argtypes = [C.c_int] * 10
        L.orc_radar_destroy.argtypes = [C.c_void_p]
        L.orc_radar_set_background_record.argtypes = [C.c_void_p, C.c_int]
        L.orc_radar_ring_size.argtypes = [C.c_void_p]
        L.orc_radar_work.argtypes = [C.c_void_p, C.POINTER(c_float_p), C.POINTER(c_float_p), C.c_long, c_float_p]
        L.orc_fft_vcc.argtypes = [C.c_int, C.c_int, C.c_int, c_float_p, C.c_long, c_float_p, c_float_p]
        L.orc_fft_vcc_f32.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, c_float_p, c_float_p]
        L.orc_matrix_transpose.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p]
        L.orc_ra_estimate.argtypes = [C.c_int, C.c_int, c_float_p, c_float_p, C.c_int, c_float_p, C.c_int,
                                      C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(RaResult)]
        L.orc_cp_remove.argtypes = [C.c_int, C.c_int, C.c_long, c_float_p, c_float_p]
        L.orc_fft_peak_detect.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_long, c_float_p,
                                          c_float_p, c_float_p, c_float_p]
        L.orc_radar_chain.argtypes = [C.c_void_p, C.POINTER(c_float_p), C.POINTER(c_float_p), C.c_int,
                                      c_float_p, c_float_p, c_float_p, c_float_p]
    return _lib


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a


class Radar:
    """mimo_ofdm_radar_impl (lib/mimo_ofdm_radar_impl.cc:66-340) restated; keeps the background ring."""

    def __init__(self, fft_len, N_tx, N_rx, N_sym, N_pre, background_removal=False,
                 background_recording=False, record_len=8, interp_factor=1, enable_tx_interleave=False):
        self.N, self.T, self.R, self.S, self.Npre, self.Ir = fft_len, N_tx, N_rx, N_sym, N_pre, interp_factor
        self._h = lib().orc_radar_create(fft_len, N_tx, N_rx, N_sym, N_pre, int(background_removal),
                                         int(background_recording), record_len, interp_factor,
                                         int(enable_tx_interleave))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_radar_destroy(self._h)
            self._h = None

    def set_background_record(self, on):
        lib().orc_radar_set_background_record(self._h, int(on))

    def ring_size(self):
        return lib().orc_radar_ring_size(self._h)

    def _ptrs(self, arrs):
        keep = [_c64(a) for a in arrs]
        arr = (c_float_p * len(keep))(*[_fp(a) for a in keep])
        return keep, arr

    def work(self, tx, rx, tx_discard=0):
        """tx: T arrays [n_items, N]; rx: R arrays [n_items, N] -> out [P, N*Ir] complex64"""
        ktx, ptx = self._ptrs(tx)
        krx, prx = self._ptrs(rx)
        out = np.empty((self.T * self.R, self.N * self.Ir), np.complex64)
        lib().orc_radar_work(self._h, ptx, prx, tx_discard, _fp(out))
        return out

    def chain(self, tx, rx, interp_angle):
        """A1->A2->A3->A4 in float32 on the CPU (baseline timing + chain parity). Returns the map."""
        ktx, ptx = self._ptrs(tx)
        krx, prx = self._ptrs(rx)
        P, NR, NA = self.T * self.R, self.N * self.Ir, self.T * self.R * interp_angle
        hpad = np.empty((P, NR), np.complex64)
        rng = np.empty((P, NR), np.complex64)
        tr = np.empty((NR, NA), np.complex64)
        mp = np.empty((NR, NA), np.complex64)
        lib().orc_radar_chain(self._h, ptx, prx, interp_angle, _fp(hpad), _fp(rng), _fp(tr), _fp(mp))
        return mp


def fft_vcc(x, forward=True, shift=False, window=None, f32=False):
    """gr::fft::fft_vcc semantics along the last axis."""
    x = _c64(x)
    n = x.shape[-1]
    batch = x.size // n
    out = np.empty_like(x)
    if f32:
        assert window is None
        lib().orc_fft_vcc_f32(n, int(forward), int(shift), batch, _fp(x), _fp(out))
    else:
        w = None if window is None else np.ascontiguousarray(window, np.float32)
        lib().orc_fft_vcc(n, int(forward), int(shift), None if w is None else _fp(w), batch, _fp(x), _fp(out))
    return out


def matrix_transpose(x, input_len, output_len, interp_factor):
    """matrix_transpose_impl::work; x: [ninput_items, input_len] -> [input_len, output_len*interp]"""
    x = _c64(x)
    nin = x.shape[0]
    out = np.empty((input_len, output_len * interp_factor), np.complex64)
    r = lib().orc_matrix_transpose(input_len, output_len, interp_factor, nin, _fp(x), _fp(out))
    if r < 0:
        raise RuntimeError("[MATRIX TRANSPOSE] input_len and output_len do not match to packet length")
    return out


def ra_estimate(m, range_bins, angle_bins, noise_discard_range_m, noise_discard_angle_deg,
                snr_threshold=0.0, power_threshold=0.0):
    """range_angle_estimator_impl::work on a [n_inputs, vlen] complex map -> RaResult"""
    m = _c64(m)
    rb = np.ascontiguousarray(range_bins, np.float32)
    ab = np.ascontiguousarray(angle_bins, np.float32)
    res = RaResult()
    lib().orc_ra_estimate(m.shape[1], m.shape[0], _fp(m), _fp(rb), len(rb), _fp(ab), len(ab),
                          noise_discard_range_m, noise_discard_angle_deg, snr_threshold, power_threshold,
                          C.byref(res))
    return res


def cp_remove(x, fft_len, cp_len):
    x = _c64(x).ravel()
    nout = x.size // (fft_len + cp_len)
    out = np.empty((nout, fft_len), np.complex64)
    r = lib().orc_cp_remove(fft_len, cp_len, x.size, _fp(x), _fp(out))
    assert r == nout
    return out


def fft_peak_detect(x, samp_rate, interp_factor, threshold, samp_protect):
    """returns (k, freq, phase, mag); k == -1 -> outputs are NaN placeholders (reference leaves them unset)"""
    x = _c64(x).ravel()
    f = np.full(1, np.nan, np.float32)
    p = np.full(1, np.nan, np.float32)
    m = np.full(1, np.nan, np.float32)
    k = lib().orc_fft_peak_detect(samp_rate, interp_factor, threshold, samp_protect, x.size, _fp(x),
                                  _fp(f), _fp(p), _fp(m))
    return k, float(f[0]), float(p[0]), float(m[0])


# ---------------------------------------------------------------------------------------------------------
# comm side: SIG codec, C1 equalizer, C2 precoder, C3 steering  (oracle/jrc_oracle_comm.c)
# ---------------------------------------------------------------------------------------------------------
c_int_p = C.POINTER(C.c_int)


class EqCfg(C.Structure):
    _fields_ = [("estimator", C.c_int), ("freq", C.c_double), ("bw", C.c_double),
                ("fft_len", C.c_int), ("cp_len", C.c_int), ("n_data", C.c_int), ("n_pilot", C.c_int),
                ("data_carriers", c_int_p), ("pilot_carriers", c_int_p),
                ("n_pilot_rows", C.c_int), ("pilot_symbols", c_float_p), ("ltf_seq", c_float_p),
                ("mapped_ltf", c_float_p), ("mapped_cols", C.c_int), ("n_mimo_ltf", C.c_int)]


class EqEvent(C.Structure):
    _fields_ = [("kind", C.c_int), ("offset", C.c_long),
                ("data_bytes", C.c_uint64), ("mcs", C.c_uint64), ("packet_type", C.c_uint64),
                ("snr", C.c_double), ("freq_offset", C.c_double), ("snr_data", C.c_double),
                ("n_chan_mean", C.c_int), ("chan_mean", C.c_float * 32)]


class PreCfg(C.Structure):
    _fields_ = [("fft_len", C.c_int), ("n_tx", C.c_int), ("n_data", C.c_int), ("n_pilot", C.c_int),
                ("data_carriers", c_int_p), ("pilot_carriers", c_int_p),
                ("n_pilot_rows", C.c_int), ("pilot_symbols", c_float_p),
                ("n_sync", C.c_int), ("sync_words", c_float_p), ("mapped_ltf", c_float_p)]


def _comm_lib():
    L = lib()
    if not getattr(L, "_comm_ready", False):
        L.orc_n_ofdm_sym.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_sig_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, c_float_p]
        L.orc_viterbi_k7.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8)]
        L.orc_sig_parse.argtypes = [C.POINTER(C.c_uint8), C.c_int, c_int_p, c_int_p, c_int_p, c_int_p]
        L.orc_eq_create.restype = C.c_void_p
        L.orc_eq_create.argtypes = [C.POINTER(EqCfg)]
        L.orc_eq_destroy.argtypes = [C.c_void_p]
        L.orc_eq_set_estimator.argtypes = [C.c_void_p, C.c_int]
        L.orc_eq_work.argtypes = [C.c_void_p, C.c_int, C.c_int, c_float_p, C.POINTER(C.c_long),
                                  C.POINTER(C.c_double), C.c_int, c_float_p, c_int_p,
                                  C.POINTER(EqEvent), C.c_int, c_int_p, c_float_p, c_int_p]
        L.orc_steering_from_channel.argtypes = [C.c_int, c_float_p, C.c_int, c_float_p]
        L.orc_dft_matrix.argtypes = [C.c_int, c_float_p]
        L.orc_precoder_work.argtypes = [C.POINTER(PreCfg), C.c_int, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        c_float_p, c_float_p, c_float_p, C.POINTER(c_float_p)]
        L._comm_ready = True
    return L


def n_ofdm_sym(mcs, n_data_carriers, nbytes):
    return _comm_lib().orc_n_ofdm_sym(mcs, n_data_carriers, nbytes)


def sig_encode(n_data_carriers, mcs, packet_type, length):
    out = np.zeros(n_data_carriers, np.float32)
    assert _comm_lib().orc_sig_encode(n_data_carriers, mcs, packet_type, length, _fp(out)) == 0
    return out


def viterbi_k7(coded_bits):
    coded = np.ascontiguousarray(coded_bits, np.uint8)
    n = coded.size // 2
    dec = np.zeros(n, np.uint8)
    _comm_lib().orc_viterbi_k7(coded.ctypes.data_as(C.POINTER(C.c_uint8)), n, dec.ctypes.data_as(C.POINTER(C.c_uint8)))
    return dec


def sig_parse(bits, n_data_carriers):
    bits = np.ascontiguousarray(bits, np.uint8)
    mcs, pt, ln, ns = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ok = _comm_lib().orc_sig_parse(bits.ctypes.data_as(C.POINTER(C.c_uint8)), n_data_carriers,
                                   C.byref(mcs), C.byref(pt), C.byref(ln), C.byref(ns))
    return bool(ok), mcs.value, pt.value, ln.value, ns.value


def _ip(a):
    return a.ctypes.data_as(c_int_p)


class Equalizer:
    """mimo_ofdm_equalizer_impl restated (lib/mimo_ofdm_equalizer_impl.cc:66-648)"""

    def __init__(self, estimator_algo, freq, bw, fft_len, cp_len, data_carriers, pilot_carriers, pilot_symbols,
                 ltf_seq, mapped_ltf_symbols, n_mimo_ltf):
        self._keep = dict(
            dc=np.ascontiguousarray(data_carriers, np.int32), pc=np.ascontiguousarray(pilot_carriers, np.int32),
            ps=_c64(pilot_symbols), ltf=_c64(ltf_seq), ml=_c64(mapped_ltf_symbols))
        k = self._keep
        self.fft_len, self.n_data, self.n_tx = fft_len, len(k["dc"]), k["ml"].shape[1] // n_mimo_ltf
        cfg = EqCfg(int(estimator_algo), freq, bw, fft_len, cp_len, len(k["dc"]), len(k["pc"]), _ip(k["dc"]), _ip(k["pc"]),
                    k["ps"].shape[0], _fp(k["ps"]), _fp(k["ltf"]), _fp(k["ml"]), k["ml"].shape[1], n_mimo_ltf)
        self._h = _comm_lib().orc_eq_create(C.byref(cfg))

    def __del__(self):
        if getattr(self, "_h", None):
            _comm_lib().orc_eq_destroy(self._h)
            self._h = None

    def set_estimator(self, algo):
        _comm_lib().orc_eq_set_estimator(self._h, int(algo))

    def general_work(self, symbols, frame_start_tags=(), noutput_items=None):
        """symbols [n_in, fft_len]; frame_start_tags: [(item_offset, phase)].  returns dict(out, consumed, events, chan_est)"""
        x = _c64(symbols).reshape(-1, self.fft_len)
        nin = x.shape[0]
        nout = nin if noutput_items is None else noutput_items
        out = np.zeros((max(nout, 1), self.n_data), np.complex64)
        offs = (C.c_long * max(1, len(frame_start_tags)))(*[int(t[0]) for t in frame_start_tags])
        vals = (C.c_double * max(1, len(frame_start_tags)))(*[float(t[1]) for t in frame_start_tags])
        cons, nev, cw = C.c_int(), C.c_int(), C.c_int()
        ev = (EqEvent * 8)()
        ce = np.zeros((self.fft_len, self.n_tx), np.complex64)
        n = _comm_lib().orc_eq_work(self._h, nout, nin, _fp(x), offs, vals, len(frame_start_tags), _fp(out),
                                    C.byref(cons), ev, 8, C.byref(nev), _fp(ce), C.byref(cw))
        events = []
        for e in ev[:nev.value]:
            d = dict(kind=e.kind, offset=e.offset)
            if e.kind == 1:
                d.update(data_bytes=e.data_bytes, mcs=e.mcs, packet_type=e.packet_type, snr=e.snr, freq_offset=e.freq_offset)
            else:
                cm = np.array(e.chan_mean[:2 * e.n_chan_mean], np.float32).view(np.complex64)
                d.update(snr_data=e.snr_data, chan_mean=cm)
            events.append(d)
        return dict(out=out[:n], consumed=cons.value, events=events, chan_est=ce if cw.value else None)


def steering_from_channel(h, phased=False):
    """T x T steering matrix from a channel row h (lib/mimo_precoder_impl.cc:846-861): returns Q[t, j]"""
    h = _c64(h)
    T = h.size
    q = np.zeros(T * T, np.complex64)
    _comm_lib().orc_steering_from_channel(T, _fp(h), int(phased), _fp(q))
    return q.reshape(T, T).T.copy()          # stored column-major


def dft_matrix(T):
    q = np.zeros(T * T, np.complex64)
    _comm_lib().orc_dft_matrix(T, _fp(q))
    return q.reshape(T, T).T.copy()


class Precoder:
    """mimo_precoder_impl restated (lib/mimo_precoder_impl.cc:78-741); deterministic inputs only"""

    def __init__(self, fft_len, N_tx, N_ss, data_carriers, pilot_carriers, pilot_symbols, sync_words, mapped_ltf_symbols):
        self._keep = dict(dc=np.ascontiguousarray(data_carriers, np.int32), pc=np.ascontiguousarray(pilot_carriers, np.int32),
                          ps=_c64(pilot_symbols), sw=_c64(sync_words), ml=_c64(mapped_ltf_symbols))
        k = self._keep
        self.N, self.T, self.n_sync, self.n_data = fft_len, N_tx, k["sw"].shape[0], len(k["dc"])
        self.cfg = PreCfg(fft_len, N_tx, len(k["dc"]), len(k["pc"]), _ip(k["dc"]), _ip(k["pc"]), k["ps"].shape[0],
                          _fp(k["ps"]), k["sw"].shape[0], _fp(k["sw"]), _fp(k["ml"]))

    def calculate_output_stream_length(self, ninput_items):
        return self.n_sync + 1 + self.T + ninput_items // self.n_data

    def work(self, symbols, mcs, packet_type, pdu_len, steer_mode=0, Q_mean=None, Q_sc=None, radar_streams=None):
        """Q_mean [T,T] / Q_sc [N,T,T] indexed [t, j]; radar_streams [(T-1), n_sym, N] or None -> out [T, n_total, N]"""
        x = _c64(symbols).ravel()
        n_sym = x.size // self.n_data
        n_total = n_sym + self.n_sync + self.T + 1
        out = np.zeros((self.T, n_total, self.N), np.complex64)
        ptrs = (c_float_p * self.T)(*[_fp(out[t]) for t in range(self.T)])
        qm = None if Q_mean is None else np.ascontiguousarray(_c64(Q_mean).T)            # -> column-major
        qs = None if Q_sc is None else np.ascontiguousarray(np.transpose(_c64(Q_sc), (0, 2, 1)))
        rs = None if radar_streams is None else _c64(radar_streams)
        r = _comm_lib().orc_precoder_work(C.byref(self.cfg), x.size, _fp(x), mcs, packet_type, pdu_len, steer_mode,
                                          None if qm is None else _fp(qm), None if qs is None else _fp(qs),
                                          None if rs is None else _fp(rs), ptrs)
        if r < 0:
            raise RuntimeError("[MIMO PRECODER] something is wrong!!")
        assert r == n_total
        return out


def _farr(v):
    return np.ascontiguousarray(np.atleast_1d(v), dtype=np.float32)


def dft_any(x, forward=True):
    """gr::fft::fft_complex of any length restated (unnormalised; double inside, rounded once to float)"""
    L = lib()
    L.orc_dft_any.argtypes = [C.c_int, C.c_int, c_float_p, c_float_p]
    x = _c64(x)
    out = np.empty_like(x)
    L.orc_dft_any(x.size, 1 if forward else 0, _fp(x), _fp(out))
    return out


class TargetSimulator:
    """target_simulator_impl (lib/target_simulator_impl.cc:66-385) restated: 1 input stream, len(position_rx) outputs."""

    def __init__(self, range_m, velocity, rcs, azimuth, position_rx, samp_rate, center_freq, self_coupling_db=-40.0,
                 rndm_phaseshift=False, self_coupling=False):
        L = lib()
        L.orc_tsim_create.restype = C.c_void_p
        L.orc_tsim_create.argtypes = [C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, C.c_int, c_float_p, C.c_int,
                                      C.c_float, C.c_float, C.c_int, C.c_int]
        L.orc_tsim_destroy.argtypes = [C.c_void_p]
        L.orc_tsim_work.argtypes = [C.c_void_p, c_float_p, C.c_int, C.POINTER(c_float_p), c_float_p, C.c_int]
        for f in (L.orc_tsim_filt_doppler, L.orc_tsim_filt_time):
            f.restype = c_float_p
        L.orc_tsim_filt_doppler.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_tsim_filt_time.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        self._L = L
        r, v, s, a, p = (_farr(x) for x in (range_m, velocity, rcs, azimuth, position_rx))
        assert r.size == v.size == s.size == a.size
        self.K, self.R = r.size, p.size
        self._h = L.orc_tsim_create(self.K, _fp(r), _fp(v), _fp(s), _fp(a), self.R, _fp(p), int(samp_rate),
                                    float(center_freq), float(self_coupling_db), int(rndm_phaseshift), int(self_coupling))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_tsim_destroy(self._h)
            self._h = None

    def work(self, x, target_phase=None, sum_targets=False):
        x = _c64(x)
        n = x.size
        out = np.zeros((self.R, n), np.complex64)
        ptrs = (c_float_p * self.R)(*[_fp(out[l]) for l in range(self.R)])
        tp = None
        if target_phase is not None:
            tpa = _c64(target_phase)
            assert tpa.size == self.K
            tp = _fp(tpa)
        self._L.orc_tsim_work(self._h, _fp(x), n, ptrs, tp, int(sum_targets))
        return out

    def filt_doppler(self, n, k):
        p = self._L.orc_tsim_filt_doppler(self._h, n, k)
        return np.ctypeslib.as_array(p, shape=(2 * n,)).copy().view(np.complex64)

    def filt_time(self, n, l, k):
        p = self._L.orc_tsim_filt_time(self._h, n, l, k)
        return np.ctypeslib.as_array(p, shape=(2 * n,)).copy().view(np.complex64)


# ---- bit codec (oracle/jrc_oracle_codec.c) -----------------------------------------------------------------------
_u8p = C.POINTER(C.c_uint8)


def _codec_lib():
    L = lib()
    if not getattr(L, "_codec_ready", False):
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [_u8p, C.c_size_t]
        L.orc_stream_encode.argtypes = [C.c_int, C.c_int, _u8p, C.c_int, C.c_int, c_float_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_viterbi_windowed.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p]
        L.orc_stream_decode.argtypes = [C.c_int, C.c_int, C.c_int, c_float_p, _u8p]
        L.orc_constellation_point.argtypes = [C.c_int, C.c_int, c_float_p, c_float_p]
        L.orc_constellation_decide.argtypes = [C.c_int, C.c_float, C.c_float]
        L._codec_ready = True
    return L


def _u8(a):
    return np.ascontiguousarray(np.frombuffer(bytes(a), np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8)


def crc32(data):
    d = _u8(data)
    return int(_codec_lib().orc_crc32(d.ctypes.data_as(_u8p), d.size))


def packet_params(mcs, n_dc, data_size_byte):
    """ofdm_mcs + packet_param (lib/utils.cc:26-111): dict of n_bpsc, n_cbps, n_dbps, n_ofdm_sym, n_data_bits, n_pad_bits"""
    bpsc = [1, 1, 2, 2, 4, 4][mcs]
    cbps = n_dc * bpsc
    dbps = cbps // 2 if mcs % 2 == 0 else cbps * 3 // 4
    ns = int(np.ceil((16 + 8 * data_size_byte + 6) / float(dbps)))
    return dict(n_bpsc=bpsc, n_cbps=cbps, n_dbps=dbps, n_ofdm_sym=ns, n_data_bits=ns * dbps,
                n_pad_bits=ns * dbps - (16 + 8 * data_size_byte + 6), n_encoded_bits=ns * cbps)


def stream_encode(mcs, n_dc, psdu, scrambler_init=1):
    """stream_encoder_impl::general_work (lib/stream_encoder_impl.cc:76-270): returns (symbols [n_sym*n_dc], tags dict)"""
    L = _codec_lib()
    p = _u8(psdu)
    pp = packet_params(mcs, n_dc, p.size + 4)
    out = np.zeros(pp["n_ofdm_sym"] * n_dc, np.complex64)
    ns, pl = C.c_int(), C.c_int()
    n = L.orc_stream_encode(mcs, n_dc, p.ctypes.data_as(_u8p), p.size, scrambler_init, _fp(out), C.byref(ns), C.byref(pl))
    if n < 0:
        return None, None
    assert n == out.size
    return out, dict(packet_len=n, packet_type=int(p[0]) if p.size else 0, mcs=mcs, pdu_len=pl.value)


def viterbi_windowed(mcs, n_sym, n_cbps, n_data_bits, bits):
    L = _codec_lib()
    b = _u8(bits)
    assert b.size == n_sym * n_cbps
    dec = np.zeros(n_data_bits + 64, np.uint8)
    n = L.orc_viterbi_windowed(mcs, n_sym, n_cbps, n_data_bits, b.ctypes.data_as(_u8p), dec.ctypes.data_as(_u8p))
    return dec[:n]


def stream_decode(mcs, n_dc, data_size_byte, symbols):
    """stream_decoder_impl decode()+descramble(): returns (crc_ok, payload bytes without the CRC), or (None, None) if refused"""
    L = _codec_lib()
    s = _c64(symbols).ravel()
    pp = packet_params(mcs, n_dc, data_size_byte)
    assert s.size >= pp["n_ofdm_sym"] * n_dc
    out = np.zeros(max(data_size_byte, 8), np.uint8)
    ok = L.orc_stream_decode(mcs, n_dc, data_size_byte, _fp(s), out.ctypes.data_as(_u8p))
    if ok < 0:
        return None, None
    return bool(ok), out[:max(data_size_byte - 4, 0)].tobytes()


def constellation_point(bpsc, value):
    re, im = C.c_float(), C.c_float()
    _codec_lib().orc_constellation_point(bpsc, value, C.byref(re), C.byref(im))
    return complex(re.value, im.value)


def constellation_decide(bpsc, z):
    return _codec_lib().orc_constellation_decide(bpsc, float(np.real(z)), float(np.imag(z)))


# ---- sync front-end (oracle/jrc_oracle_sync.c) -----------------------------------------------------------------
_u64p = C.POINTER(C.c_uint64)
_dblp = C.POINTER(C.c_double)


def _sync_lib():
    L = lib()
    if not getattr(L, "_sync_ready", False):
        L.orc_moving_avg_work.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, c_float_p, c_float_p]
        L.orc_sync_metrics.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_float_p, c_float_p, c_float_p]
        L.orc_fd_create.restype = C.c_void_p
        L.orc_fd_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.orc_fd_destroy.argtypes = [C.c_void_p]
        L.orc_fd_work.argtypes = [C.c_void_p, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, C.POINTER(C.c_int),
                                  _u64p, _dblp, C.c_int, C.POINTER(C.c_int)]
        L.orc_fs_create.restype = C.c_void_p
        L.orc_fs_create.argtypes = [C.c_int, C.c_int, C.c_int, c_float_p, C.c_int]
        L.orc_fs_destroy.argtypes = [C.c_void_p]
        L.orc_fs_frame_start.argtypes = [C.c_void_p]
        L.orc_fs_freq_offset.restype = C.c_double
        L.orc_fs_freq_offset.argtypes = [C.c_void_p]
        L.orc_fs_work.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, _u64p, _dblp, C.c_int, c_float_p,
                                  C.POINTER(C.c_int), _u64p, _dblp, C.POINTER(C.c_int)]
        L._sync_ready = True
    return L


def moving_avg(x, length, scale=1.0, max_iter=16000, history=None):
    """moving_avg_impl::work for one call: x = the new items; history = the length-1 items before them (zeros at stream start)"""
    L = _sync_lib()
    x = _c64(x).ravel()
    h = np.zeros(length - 1, np.complex64) if history is None else _c64(history)
    buf = np.concatenate([h, x]).astype(np.complex64)
    out = np.zeros(x.size, np.complex64)
    n = L.orc_moving_avg_work(length, float(scale), max_iter, x.size, _fp(buf), _fp(out))
    return out[:n]


def sync_metrics(x, delay, window, pwindow, pscale):
    L = _sync_lib()
    x = _c64(x).ravel()
    xd, ia, ic = np.zeros_like(x), np.zeros_like(x), np.zeros(x.size, np.float32)
    L.orc_sync_metrics(_fp(x), x.size, delay, window, pwindow, float(pscale), _fp(xd), _fp(ia), _fp(ic))
    return xd, ia, ic


class FrameDetector:
    """frame_detector_impl (lib/frame_detector_impl.cc:40-205) restated; work() = one general_work call"""

    def __init__(self, fft_len, cp_len, threshold, min_n_peaks, ignore_gap):
        self._L = _sync_lib()
        self._h = self._L.orc_fd_create(fft_len, cp_len, float(threshold), int(min_n_peaks), int(ignore_gap))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_fd_destroy(self._h)
            self._h = None

    def work(self, x, in_abs, in_cor, noutput):
        x, in_abs = _c64(x), _c64(in_abs)
        in_cor = np.ascontiguousarray(in_cor, np.float32)
        n = min(x.size, in_abs.size, in_cor.size)
        out = np.zeros(max(noutput, 1), np.complex64)
        cons, nt = C.c_int(), C.c_int()
        to, tc = (C.c_uint64 * 4)(), (C.c_double * 4)()
        no = self._L.orc_fd_work(self._h, noutput, n, _fp(x), _fp(in_abs), _fp(in_cor), _fp(out), C.byref(cons), to, tc, 4, C.byref(nt))
        return out[:no], cons.value, [(int(to[i]), float(tc[i])) for i in range(nt.value)]

    def run(self, x, in_abs, in_cor, chunk=1 << 30):
        """drive work() until the inputs are used up, like a scheduler that always offers everything it has"""
        pos, outs, tags = 0, [], []
        while pos < len(x):
            n = min(chunk, len(x) - pos)
            o, c, t = self.work(x[pos:pos + n], in_abs[pos:pos + n], in_cor[pos:pos + n], n)
            outs.append(o)
            tags += t
            if c == 0 and o.size == 0:
                break
            pos += c
        return np.concatenate(outs) if outs else np.zeros(0, np.complex64), tags


class FrameSync:
    """frame_sync_impl (lib/frame_sync_impl.cc:45-289) restated; work() = one general_work call"""

    def __init__(self, fft_len, cp_len, sync_length, ltf_seq_time):
        self._L = _sync_lib()
        t = _c64(ltf_seq_time)
        self._h = self._L.orc_fs_create(fft_len, cp_len, int(sync_length), _fp(t), t.size)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_fs_destroy(self._h)
            self._h = None

    @property
    def frame_start(self):
        return self._L.orc_fs_frame_start(self._h)

    @property
    def freq_offset(self):
        return self._L.orc_fs_freq_offset(self._h)

    def work(self, x, x_delayed, tags, noutput):
        """tags: [(absolute offset on input 0, value)]"""
        x, xd = _c64(x), _c64(x_delayed)
        out = np.zeros(max(noutput, 1), np.complex64)
        nt = len(tags)
        to = (C.c_uint64 * max(nt, 1))(*[int(t[0]) for t in tags])
        tv = (C.c_double * max(nt, 1))(*[float(t[1]) for t in tags])
        cons, nto = C.c_int(), C.c_int()
        oo, ov = (C.c_uint64 * 1)(), (C.c_double * 1)()
        no = self._L.orc_fs_work(self._h, noutput, x.size, xd.size, _fp(x), _fp(xd), to, tv, nt, _fp(out), C.byref(cons), oo, ov, C.byref(nto))
        if no < 0:
            raise RuntimeError("[FRAME SYNC] Something is wrong!")
        return out[:no], cons.value, [(int(oo[0]), float(ov[0]))] if nto.value else []

    def run(self, x, x_delayed, tags, chunk=8192):
        pos, outs, otags, idle = 0, [], [], 0
        n = min(len(x), len(x_delayed))
        while pos < n and idle < 3:
            m = min(chunk, n - pos)
            o, c, t = self.work(x[pos:pos + m], x_delayed[pos:pos + m], tags, m)
            outs.append(o)
            otags += t
            idle = idle + 1 if (c == 0 and o.size == 0) else 0
            pos += c
        return np.concatenate(outs) if outs else np.zeros(0, np.complex64), otags


def _norm_sets(sets, fft_len, shifted):
    out = []
    for st in sets:
        v = [int(c) + fft_len if int(c) < 0 else int(c) for c in st]
        if any(c > fft_len or c < 0 for c in v):
            raise ValueError("carrier index out of bounds")
        out.append([(c + fft_len // 2) % fft_len for c in v] if shifted else v)
    return out


class FrameGenerator:
    """ofdm_frame_generator_impl (lib/ofdm_frame_generator_impl.cc:55-216) restated: the SISO carrier allocator"""

    def __init__(self, fft_len, occupied_carriers, pilot_carriers, pilot_symbols, sync_words, ltf_len=0, len_tag_key="packet_len",
                 output_is_shifted=True):
        self.N = fft_len
        self.occ = _norm_sets(occupied_carriers, fft_len, output_is_shifted)
        self.pil = _norm_sets(pilot_carriers, fft_len, output_is_shifted)
        self.psym = [np.asarray(p, np.complex64) for p in pilot_symbols]
        self.sync = _c64(np.asarray(sync_words, np.complex64).reshape(-1, fft_len)) if len(sync_words) else np.zeros((0, fft_len), np.complex64)
        for i in range(max(len(self.pil), len(self.psym))):
            if len(self.pil[i % len(self.pil)]) != len(self.psym[i % len(self.psym)]):
                raise ValueError("pilot_carriers do not match pilot_symbols")
        self.sps = sum(len(o) for o in self.occ)

    def calculate_output_stream_length(self, nin):
        nout = (nin // self.sps) * len(self.occ)
        i, k = 0, 0
        while i < nin % self.sps:
            nout += 1
            i += len(self.occ[k % len(self.occ)])
            k += 1
        return nout + len(self.sync)

    def work(self, x):
        L = lib()
        ip = C.POINTER(C.c_int)
        L.orc_frame_generator.argtypes = [C.c_int, C.c_int, ip, ip, C.c_int, ip, ip, C.c_int, c_float_p, C.c_int, c_float_p, C.c_int,
                                          c_float_p, c_float_p, C.c_int]
        x = _c64(x).ravel()
        nout = self.calculate_output_stream_length(x.size)
        out = np.zeros((nout, self.N), np.complex64)
        osz = np.array([len(o) for o in self.occ], np.int32)
        ofl = np.array([c for o in self.occ for c in o], np.int32)
        psz = np.array([len(o) for o in self.pil], np.int32)
        pfl = np.array([c for o in self.pil for c in o] or [0], np.int32)
        psf = _c64(np.concatenate(self.psym) if sum(p.size for p in self.psym) else np.zeros(1, np.complex64))
        sw = _c64(self.sync if self.sync.size else np.zeros((1, self.N), np.complex64))
        n = L.orc_frame_generator(self.N, len(self.occ), osz.ctypes.data_as(ip), ofl.ctypes.data_as(ip), len(self.pil), psz.ctypes.data_as(ip),
                                  pfl.ctypes.data_as(ip), len(self.psym), _fp(psf), len(self.sync), _fp(sw), x.size, _fp(x), _fp(out), nout)
        assert n == nout
        return out
