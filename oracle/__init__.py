"""CPU oracle for the gr-mimo-ofdm-jrc hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference has no tests/golden vectors and cannot be built in this image
(GNU Radio 3.8 / Eigen3 / Boost / FFTW3f / VOLK are absent; no stand-ins are written for them).
The functions here restate the reference algorithm (file:line cited in oracle/jrc_oracle.c) and are
checked against closed forms and the constant tables minted from the reference's embedded Python
module (tests/golden/).

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
    srcs = [os.path.join(_HERE, f) for f in ("jrc_oracle.c", "jrc_oracle_comm.c", "jrc_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE])
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
