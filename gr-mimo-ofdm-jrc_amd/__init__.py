"""gr-mimo-ofdm-jrc_amd — MI355X (gfx950) implementation of the gr-mimo-ofdm-jrc radar/equalizer hot path.

The product is the C-ABI shared library built from csrc/*.hip (see include/jrc.h).  This module is the
thin Python host side: a ctypes binding plus classes that mirror the reference's block interface
(same names, constructor arguments and work() semantics as include/mimo_ofdm_jrc/*.h in the reference)
so the parity tests read like tests of the reference blocks.

There is NO CPU fallback: importing works anywhere (so the symbol checks can run), but creating a
Context without a usable HIP device raises JrcError.  Nothing here imports oracle/.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JRC_LIB_PATH") or os.path.join(_HERE, "lib", "libjrc_hip.so")    # JRC_LIB_PATH: kernel-variant experiments (tools/ra_variants.py)
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

JRC_OK = 0
JRC_ERR_NO_DEVICE = -1
JRC_ERR_HIP = -2
JRC_ERR_INVALID_ARG = -3
JRC_ERR_UNSUPPORTED = -4
JRC_ERR_LENGTH_MISMATCH = -5
JRC_ERR_SHORT_INPUT = -6

_cfp = C.POINTER(C.c_float)
_vp = C.c_void_p


class JrcError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("jrc status %d: %s" % (status, msg))
        self.status = status


class RaResult(C.Structure):
    """jrc_ra_result (include/jrc.h)"""
    _fields_ = [("peak_range_idx", C.c_int32), ("peak_angle_idx", C.c_int32),
                ("angle_null_idx", C.c_int32),
                ("discard_range_idx", C.c_int32), ("discard_angle_idx", C.c_int32),
                ("n_noise_samples", C.c_int32),
                ("peak_power", C.c_float), ("noise_power", C.c_float), ("snr_est", C.c_float),
                ("range_val", C.c_float), ("angle_val", C.c_float),
                ("published", C.c_int32)]


class ChainCfg(C.Structure):
    """jrc_chain_cfg (include/jrc.h)"""
    _fields_ = [("fft_len", C.c_int32), ("N_tx", C.c_int32), ("N_rx", C.c_int32), ("N_sym", C.c_int32),
                ("N_pre", C.c_int32), ("interp_range", C.c_int32), ("interp_angle", C.c_int32),
                ("enable_tx_interleave", C.c_int32), ("n_items", C.c_int32),
                ("noise_discard_range_m", C.c_float), ("noise_discard_angle_deg", C.c_float),
                ("snr_threshold", C.c_float), ("power_threshold", C.c_float)]


_lib = None

# The library launches on streams of its own (non-blocking: they do not order themselves against the caller's default stream, like any
# stream a GNU Radio block would own).  A caller that prepares device buffers with torch — tests, tools — therefore has to finish
# torch's work before handing them over.  set_torch_stream_sync(True) makes every call into the library do that first: it waits for
# torch's CURRENT stream only (never for the library's streams, whose ordering stays what the C ABI gives it), and after the call makes
# torch's stream depend on the contexts' streams (a stream dependency, no host wait), so that torch reads what the call enqueued.  Off by
# default (bench.py keeps its own discipline, and a block fed from host buffers never needs it); tests/conftest.py switches it on.
_TORCH_STREAM_SYNC = False


def set_torch_stream_sync(on):
    global _TORCH_STREAM_SYNC
    _TORCH_STREAM_SYNC = bool(on)


_live_contexts = None       # weak set of Context objects (their streams are what torch's stream is made to wait for after a call)


def _torch():
    torch = sys.modules.get("torch")
    return torch if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized() else None


def _sync_torch_stream():
    torch = _torch()
    if torch is not None:
        torch.cuda.current_stream().synchronize()


def _torch_waits_for_library():
    """the other direction, as a stream dependency (no host wait): what torch queues from here on runs behind what the contexts' streams hold now"""
    torch = _torch()
    if torch is None or not _live_contexts or _lib is None:
        return
    cur = torch.cuda.current_stream()
    for c in list(_live_contexts):
        h = getattr(c, "h", None)
        if h:
            sp = _lib._cdll.jrc_stream(h)
            if sp:
                cur.wait_stream(torch.cuda.ExternalStream(sp, device=c.device))


class _Fn:
    """one entry point of the shared library; attribute access (argtypes / restype) goes to the ctypes function"""
    __slots__ = ("f",)

    def __init__(self, f):
        object.__setattr__(self, "f", f)

    def __call__(self, *a):
        if not _TORCH_STREAM_SYNC:
            return self.f(*a)
        _sync_torch_stream()
        r = self.f(*a)
        _torch_waits_for_library()
        return r

    def __getattr__(self, k):
        return getattr(self.f, k)

    def __setattr__(self, k, v):
        setattr(self.f, k, v)


class _Lib:
    def __init__(self, cdll):
        self.__dict__["_cdll"] = cdll
        self.__dict__["_fns"] = {}

    def __getattr__(self, name):
        fn = self._fns.get(name)
        if fn is None:
            fn = self._fns[name] = _Fn(getattr(self._cdll, name))
        return fn


def load(build_if_missing=False):
    """dlopen libjrc_hip.so.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            _build.build()
        else:
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP extension is mandatory, there is no CPU fallback)" % LIB_PATH)
    # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7, NEEDED as "libamdhip64.so"): it must be
    # loaded first so that this library binds to the same HIP runtime instance instead of a second copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = _Lib(C.CDLL(LIB_PATH))
    L.jrc_abi_version.restype = C.c_int
    L.jrc_device_count.restype = C.c_int
    L.jrc_create.argtypes = [C.c_int, C.POINTER(_vp)]
    L.jrc_destroy.argtypes = [_vp]
    L.jrc_strerror.restype = C.c_char_p
    L.jrc_strerror.argtypes = [C.c_int]
    L.jrc_last_error.restype = C.c_char_p
    L.jrc_last_error.argtypes = [_vp]
    L.jrc_device_name.argtypes = [_vp, C.c_char_p, C.c_size_t]
    L.jrc_sync.argtypes = [_vp]
    L.jrc_stream.restype = _vp
    L.jrc_stream.argtypes = [_vp]
    L.jrc_dev_malloc.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
    L.jrc_dev_free.argtypes = [_vp, _vp]
    L.jrc_dev_memset.argtypes = [_vp, _vp, C.c_int, C.c_size_t]
    L.jrc_memcpy_h2d.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.jrc_memcpy_d2h.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.jrc_radar_create.argtypes = [_vp] + [C.c_int] * 10 + [C.POINTER(_vp)]
    L.jrc_radar_destroy.argtypes = [_vp]
    L.jrc_radar_set_background_record.argtypes = [_vp, C.c_int]
    L.jrc_radar_ring_size.argtypes = [_vp]
    L.jrc_radar_work.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp), C.c_size_t, C.c_size_t, C.c_size_t, _vp]
    L.jrc_radar_chanest_dev.argtypes = [_vp] + [C.c_int] * 8 + [_vp, _vp, _vp]
    L.jrc_radar_chanest_td_dev.argtypes = [_vp] + [C.c_int] * 7 + [C.c_long, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
    L.jrc_chain_run_td_dev.argtypes = [_vp, C.c_int, _vp, _vp, C.c_int, C.c_long, _vp, _vp, _vp, _vp]
    L.jrc_fft_vcc.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp]
    L.jrc_fft_vcc_dev.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp, _vp]
    L.jrc_matrix_transpose.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]
    L.jrc_matrix_transpose_dev.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp]
    L.jrc_ra_estimate.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(RaResult)]
    L.jrc_cp_remove.argtypes = [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp]
    L.jrc_cp_remove_fft.argtypes = [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp]
    L.jrc_cp_remove_fft_dev.argtypes = [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp]
    L.jrc_ofdm_mod.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp]
    L.jrc_ofdm_mod_dev.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp, _vp]
    L.jrc_fft_peak_detect.argtypes = [_vp, C.c_int, C.c_float, C.c_float, C.c_int, C.c_size_t, _vp,
                                      _cfp, _cfp, _cfp, C.POINTER(C.c_int)]
    L.jrc_chain_create.argtypes = [_vp, C.POINTER(ChainCfg), _vp, _vp, C.c_int, C.POINTER(_vp)]
    L.jrc_chain_destroy.argtypes = [_vp]
    for fn in ("jrc_chain_frame_bytes", "jrc_chain_chanest_bytes", "jrc_chain_map_bytes"):
        getattr(L, fn).restype = C.c_size_t
        getattr(L, fn).argtypes = [_vp]
    L.jrc_chain_run_dev.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
    L.jrc_chain_fetch_results.argtypes = [_vp, C.c_int, _vp, C.POINTER(RaResult), _vp]
    L.jrc_chain_fetch_results_begin.argtypes = [_vp, C.c_int, _vp, _vp]
    L.jrc_chain_fetch_results_end.argtypes = [_vp, C.POINTER(RaResult), C.POINTER(C.c_int)]
    L.jrc_range_doppler_dev.argtypes = [_vp, C.POINTER(ChainCfg), C.c_int, C.c_int, _vp, _vp, _vp, _vp]
    L.jrc_chain_set_timing.argtypes = [_vp, C.c_int]
    L.jrc_chain_launches_per_run.argtypes = [_vp, C.c_int]
    L.jrc_chain_feed_create.argtypes = [_vp, C.POINTER(ChainCfg), _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]
    L.jrc_chain_feed_destroy.argtypes = [_vp]
    L.jrc_chain_feed_destroy.restype = None
    for fn in ("jrc_chain_feed_frame_bytes", "jrc_chain_feed_map_bytes"):
        getattr(L, fn).argtypes = [_vp]
        getattr(L, fn).restype = C.c_size_t
    L.jrc_chain_feed_acquire.argtypes = [_vp, C.POINTER(_vp)]
    L.jrc_chain_feed_submit.argtypes = [_vp, _vp, C.c_int]
    L.jrc_chain_feed_submit_rx.argtypes = [_vp, _vp, C.c_int]
    L.jrc_chain_feed_set_tx.argtypes = [_vp, _vp]
    L.jrc_chain_feed_poll.argtypes = [_vp]
    L.jrc_chain_feed_collect.argtypes = [_vp, C.POINTER(RaResult), _vp, C.POINTER(C.c_int)]
    L.jrc_chain_feed_pending.argtypes = [_vp]
    L.jrc_chain_feed_stats.argtypes = [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.jrc_chain_get_timing.argtypes = [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.jrc_chain_set_background.argtypes = [_vp, C.c_int, C.c_int, C.c_int]
    L.jrc_chain_share_background.argtypes = [_vp, _vp]
    L.jrc_chain_background_size.argtypes = [_vp]
    L.jrc_chain_prime_background_dev.argtypes = [_vp, C.c_int, _vp, _vp]
    L.jrc_chain_set_write_map.argtypes = [_vp, C.c_int]
    L.jrc_chain_set_map_format.argtypes = [_vp, C.c_int]
    L.jrc_chain_feed_set_background.argtypes = [_vp, C.c_int, C.c_int, C.c_int]
    L.jrc_chain_feed_set_write_map.argtypes = [_vp, C.c_int]
    L.jrc_chain_feed_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(ChainCfg), _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]
    L.jrc_chain_feed_n_devices.argtypes = [_vp]
    L.jrc_chain_feed_last_error.argtypes = [_vp]
    L.jrc_chain_feed_last_error.restype = C.c_char_p
    L.jrc_chain_feed_submit_many.argtypes = [_vp, C.POINTER(_vp), C.POINTER(C.c_int), C.c_int]
    _lib = L
    return L


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


def _ptr(a):
    return a.ctypes.data_as(_vp)


class Context:
    """jrc_ctx: one per host thread, bound to one GPU."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        st = self.lib.jrc_create(device, C.byref(h))
        if st != JRC_OK:
            raise JrcError(st, self.lib.jrc_strerror(st).decode())
        self.h = h
        self.device = device
        import weakref
        global _live_contexts
        if _live_contexts is None:
            _live_contexts = weakref.WeakSet()
        _live_contexts.add(self)
        # jrc_destroy frees the context's own twiddles, scratch, pinned memory and stream — never the objects created on it (chains, feeds,
        # blocks: device buffers, streams, events of their own).  So the context remembers them (weakly) and close() destroys them first.
        self._children = weakref.WeakSet()

    def _adopt(self, child):
        self._children.add(child)

    def check(self, st):
        if st < 0:
            msg = self.lib.jrc_last_error(self.h).decode() or self.lib.jrc_strerror(st).decode()
            if st == JRC_ERR_LENGTH_MISMATCH:
                raise RuntimeError(msg)       # std::runtime_error in the reference
            if st == JRC_ERR_INVALID_ARG:
                raise ValueError(msg)         # std::invalid_argument in the reference
            raise JrcError(st, msg)
        return st

    def device_name(self):
        buf = C.create_string_buffer(256)
        self.check(self.lib.jrc_device_name(self.h, buf, 256))
        return buf.value.decode()

    def sync(self):
        self.check(self.lib.jrc_sync(self.h))

    def close(self):
        h = getattr(self, "h", None)
        if h:
            for child in list(getattr(self, "_children", ())):   # objects created on this context, while it is still alive to destroy them with
                try:
                    child.close()
                except Exception:
                    pass
            self.h = None                     # first: nothing may look the stream of a context up while or after it is destroyed
            self.lib.jrc_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")) if _device_count() > 1 else 0)
    return _default_ctx


def _device_count():
    return load().jrc_device_count()


# ---------------------------------------------------------------------------------------------------------
# Block mirrors.  Names / constructor arguments follow the reference's public headers
# (include/mimo_ofdm_jrc/*.h); work() takes and returns numpy arrays instead of GNU Radio buffers.
# ---------------------------------------------------------------------------------------------------------
class mimo_ofdm_radar:
    """include/mimo_ofdm_jrc/mimo_ofdm_radar.h:35-64; work = lib/mimo_ofdm_radar_impl.cc:131-340"""

    def __init__(self, fft_len, N_tx, N_rx, N_sym, N_pre, background_removal=False, background_recording=False,
                 record_len=8, interp_factor=1, enable_tx_interleave=False, radar_chan_file="",
                 len_tag_key="packet_len", debug=False, ctx=None):
        self.ctx = ctx or default_context()
        self.fft_len, self.N_tx, self.N_rx, self.N_sym, self.N_pre = fft_len, N_tx, N_rx, N_sym, N_pre
        self.interp_factor = interp_factor
        h = _vp()
        self.ctx.check(self.ctx.lib.jrc_radar_create(self.ctx.h, fft_len, N_tx, N_rx, N_sym, N_pre,
                                                     int(background_removal), int(background_recording), record_len,
                                                     interp_factor, int(enable_tx_interleave), C.byref(h)))
        self.h = h
        self.ctx._adopt(self)

    def set_background_record(self, background_record):
        self.ctx.check(self.ctx.lib.jrc_radar_set_background_record(self.h, int(background_record)))

    def ring_size(self):
        return self.ctx.lib.jrc_radar_ring_size(self.h)

    def general_work(self, tx, rx, tx_discard=0):
        """tx: N_tx arrays [n_items, fft_len]; rx: N_rx arrays -> [P, fft_len*interp_factor] complex64.
        The block emits P items and a packet_len=P tag (lib/mimo_ofdm_radar_impl.cc:303-309)."""
        ktx = [_c64(a) for a in tx]
        krx = [_c64(a) for a in rx]
        assert len(ktx) == self.N_tx and len(krx) == self.N_rx
        ptx = (_vp * len(ktx))(*[_ptr(a) for a in ktx])
        prx = (_vp * len(krx))(*[_ptr(a) for a in krx])
        n_tx = min(a.size // self.fft_len for a in ktx)
        n_rx = min(a.size // self.fft_len for a in krx)
        P = self.N_tx * self.N_rx
        out = np.empty((P, self.fft_len * self.interp_factor), np.complex64)
        n = self.ctx.check(self.ctx.lib.jrc_radar_work(self.h, ptx, prx, n_tx, n_rx, tx_discard, _ptr(out)))
        assert n == P
        return out

    def close(self):
        if getattr(self, "h", None) and getattr(getattr(self, "ctx", None), "h", None):   # (Context.close() closes its children before it destroys itself; a context already gone means this object went with it)
            self.ctx.lib.jrc_radar_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class fft_vcc:
    """stock gr::fft::fft_vcc(fft_size, forward, window, shift) as used by the flowgraphs"""

    def __init__(self, fft_size, forward, window=None, shift=False, ctx=None):
        self.ctx = ctx or default_context()
        self.fft_size, self.forward, self.shift = fft_size, bool(forward), bool(shift)
        self.window = None if window is None or len(window) == 0 else np.ascontiguousarray(window, np.float32)
        if self.window is not None and len(self.window) != fft_size:
            raise ValueError("window size must equal fft_size")

    def work(self, x):
        x = _c64(x)
        batch = x.size // self.fft_size
        out = np.empty_like(x)
        self.ctx.check(self.ctx.lib.jrc_fft_vcc(self.ctx.h, self.fft_size, int(self.forward), int(self.shift),
                                                None if self.window is None else _ptr(self.window), batch,
                                                _ptr(x), _ptr(out)))
        return out


class matrix_transpose:
    """include/mimo_ofdm_jrc/matrix_transpose.h; work = lib/matrix_transpose_impl.cc:69-110"""

    def __init__(self, input_len, output_len, interp_factor, debug=False, len_key="packet_len", ctx=None):
        self.ctx = ctx or default_context()
        self.input_len, self.output_len, self.interp_factor = input_len, output_len, interp_factor

    def calculate_output_stream_length(self, ninput_items):
        return self.input_len

    def work(self, x):
        x = _c64(x).reshape(-1, self.input_len)
        out = np.empty((self.input_len, self.output_len * self.interp_factor), np.complex64)
        self.ctx.check(self.ctx.lib.jrc_matrix_transpose(self.ctx.h, self.input_len, self.output_len,
                                                         self.interp_factor, x.shape[0], _ptr(x), _ptr(out)))
        return out


class range_angle_estimator:
    """include/mimo_ofdm_jrc/range_angle_estimator.h; work = lib/range_angle_estimator_impl.cc:121-284.
    work() returns the jrc_ra_result; `published` says whether the reference would have emitted the
    {range, angle, power, snr} message on port "params"."""

    def __init__(self, vlen, range_bins, angle_bins, noise_discard_range_m, noise_discard_angle_deg,
                 snr_threshold, power_threshold, stats_path="", stats_record=False, len_key="packet_len",
                 debug=False, ctx=None):
        self.ctx = ctx or default_context()
        self.vlen = vlen
        self.range_bins = np.ascontiguousarray(range_bins, np.float32)
        self.angle_bins = np.ascontiguousarray(angle_bins, np.float32)
        self.ndr, self.nda = float(noise_discard_range_m), float(noise_discard_angle_deg)
        self.snr_threshold, self.power_threshold = float(snr_threshold), float(power_threshold)

    def set_snr_threshold(self, v):
        self.snr_threshold = float(v)

    def set_power_threshold(self, v):
        self.power_threshold = float(v)

    def work(self, m):
        m = _c64(m).reshape(-1, self.vlen)
        res = RaResult()
        self.ctx.check(self.ctx.lib.jrc_ra_estimate(self.ctx.h, self.vlen, m.shape[0], _ptr(m),
                                                    _ptr(self.range_bins), len(self.range_bins),
                                                    _ptr(self.angle_bins), len(self.angle_bins),
                                                    self.ndr, self.nda, self.snr_threshold, self.power_threshold,
                                                    C.byref(res)))
        return res


class ofdm_cyclic_prefix_remover:
    """include/mimo_ofdm_jrc/ofdm_cyclic_prefix_remover.h; work = lib/ofdm_cyclic_prefix_remover_impl.cc:69-99"""

    def __init__(self, fft_len, cp_len, len_key="packet_len", ctx=None):
        self.ctx = ctx or default_context()
        self.fft_len, self.cp_len = fft_len, cp_len

    def calculate_output_stream_length(self, ninput_items):
        return ninput_items // (self.fft_len + self.cp_len)

    def work(self, x, fused_fft=False):
        x = _c64(x).ravel()
        nout = x.size // (self.fft_len + self.cp_len)
        out = np.empty((nout, self.fft_len), np.complex64)
        fn = self.ctx.lib.jrc_cp_remove_fft if fused_fft else self.ctx.lib.jrc_cp_remove
        n = self.ctx.check(fn(self.ctx.h, self.fft_len, self.cp_len, x.size, _ptr(x), _ptr(out)))
        assert n == nout
        return out


def ofdm_mod(x, fft_len, cp_len, window=None, ctx=None):
    """TX OFDM modulator: fft_vxx reverse+shift(+window) then cyclic prefixer; x [n_sym, fft_len] -> [n_sym, cp+fft_len]"""
    ctx = ctx or default_context()
    x = _c64(x).reshape(-1, fft_len)
    w = None if window is None else np.ascontiguousarray(window, np.float32)
    out = np.empty((x.shape[0], cp_len + fft_len), np.complex64)
    n = ctx.check(ctx.lib.jrc_ofdm_mod(ctx.h, fft_len, cp_len, None if w is None else _ptr(w), x.shape[0], _ptr(x), _ptr(out)))
    assert n == x.shape[0]
    return out


class fft_peak_detect:
    """include/mimo_ofdm_jrc/fft_peak_detect.h; work = lib/fft_peak_detect_impl.cc:77-111"""

    def __init__(self, samp_rate, interp_factor, threshold, samp_protect, max_freq=(), cut_max_freq=False,
                 len_key="packet_len", ctx=None):
        self.ctx = ctx or default_context()
        self.samp_rate, self.interp_factor = int(samp_rate), float(interp_factor)
        self.threshold, self.samp_protect = float(threshold), int(samp_protect)

    def set_threshold(self, t):
        self.threshold = float(t)

    def set_samp_protect(self, s):
        self.samp_protect = int(s)

    def work(self, x):
        """returns (k, freq, phase, mag); k == -1 -> the reference leaves its outputs unset (NaN here)"""
        x = _c64(x).ravel()
        f = np.full(1, np.nan, np.float32)
        p = np.full(1, np.nan, np.float32)
        m = np.full(1, np.nan, np.float32)
        k = C.c_int(-2)
        n = self.ctx.check(self.ctx.lib.jrc_fft_peak_detect(
            self.ctx.h, self.samp_rate, self.interp_factor, self.threshold, self.samp_protect, x.size, _ptr(x),
            f.ctypes.data_as(_cfp), p.ctypes.data_as(_cfp), m.ctypes.data_as(_cfp), C.byref(k)))
        assert n == 1
        return k.value, float(f[0]), float(p[0]), float(m[0])


# ---------------------------------------------------------------------------------------------------------
# Fused, device-resident radar chain (torch tensors are only device memory + stream plumbing)
# ---------------------------------------------------------------------------------------------------------
def radar_axes(fft_len, samp_rate, interp_range, n_pairs, interp_angle):
    """range / angle bin axes exactly as the radar flowgraph hands them to the estimator
    (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:1390-1400), cast to float32."""
    nr, na = fft_len * interp_range, n_pairs * interp_angle
    range_bins = np.linspace(0, 3e8 * fft_len / (2 * samp_rate), nr)
    angle_bins = np.arcsin(2 / na * (np.arange(0, na) - np.floor(na / 2) + 0.5)) * 180 / np.pi
    return range_bins.astype(np.float32), angle_bins.astype(np.float32)


class RadarChain:
    """jrc_chain: A1 -> A2 -> A3 -> A4 -> A5 for a batch of frames resident in HBM."""

    def __init__(self, fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, range_bins, angle_bins,
                 noise_discard_range_m, noise_discard_angle_deg, snr_threshold=0.0, power_threshold=0.0,
                 n_items=None, enable_tx_interleave=False, max_frames=64, ctx=None):
        self.ctx = ctx or default_context()
        self.cfg = ChainCfg(fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, int(enable_tx_interleave),
                            n_items if n_items is not None else N_pre + N_sym,
                            noise_discard_range_m, noise_discard_angle_deg, snr_threshold, power_threshold)
        self.P, self.NR, self.NA = N_tx * N_rx, fft_len * interp_range, N_tx * N_rx * interp_angle
        self.max_frames = max_frames
        rb = np.ascontiguousarray(range_bins, np.float32)
        ab = np.ascontiguousarray(angle_bins, np.float32)
        assert len(rb) == self.NR and len(ab) == self.NA
        h = _vp()
        self.ctx.check(self.ctx.lib.jrc_chain_create(self.ctx.h, C.byref(self.cfg), _ptr(rb), _ptr(ab), max_frames,
                                                     C.byref(h)))
        self.h = h
        self.ctx._adopt(self)
        L = self.ctx.lib
        self.frame_bytes = L.jrc_chain_frame_bytes(h)
        self.chanest_bytes = L.jrc_chain_chanest_bytes(h)
        self.map_bytes = L.jrc_chain_map_bytes(h)

    def frame_shape(self):
        c = self.cfg
        return (c.N_tx + c.N_rx, c.n_items, c.fft_len)

    def alloc(self, n_frames, device, with_map=True, power_map=False):
        """device buffers as torch tensors (complex64 viewed as float32 pairs); with_map=False for detect-only mode, power_map=True
        for the float |z|^2 format"""
        import torch
        c = self.cfg
        if power_map:
            b = self.alloc(n_frames, device, with_map=False)
            b["map"] = torch.empty((n_frames, self.NR, self.NA), dtype=torch.float32, device=device)
            return b
        return dict(
            frames=torch.empty((n_frames, c.N_tx + c.N_rx, c.n_items, c.fft_len, 2), dtype=torch.float32, device=device),
            chanest=torch.empty((n_frames, self.P, c.fft_len, 2), dtype=torch.float32, device=device),
            map=torch.empty((n_frames, self.NR, self.NA, 2), dtype=torch.float32, device=device) if with_map else None,
            results=torch.empty((n_frames, C.sizeof(RaResult)), dtype=torch.uint8, device=device),
        )

    def run(self, bufs, n_frames, stream=None):
        """asynchronous on `stream` (an int hipStream_t handle, e.g. torch.cuda.current_stream().cuda_stream)"""
        self.ctx.check(self.ctx.lib.jrc_chain_run_dev(self.h, n_frames, bufs["frames"].data_ptr(),
                                                      bufs["chanest"].data_ptr(),
                                                      bufs["map"].data_ptr() if bufs.get("map") is not None else None,
                                                      bufs["results"].data_ptr(), stream))

    def set_background(self, background_removal, background_recording, record_len):
        """mimo_ofdm_radar's background_removal / background_recording / record_len (include/mimo_ofdm_jrc/mimo_ofdm_radar.h:52-56) for
        the batched chain: the frames of a batch are consecutive frames of one radar stream"""
        self.ctx.check(self.ctx.lib.jrc_chain_set_background(self.h, int(background_removal), int(background_recording), record_len))
        self._bg = (bool(background_removal), record_len)

    def set_background_record(self, background_record):
        """mimo_ofdm_radar::set_background_record (lib/mimo_ofdm_radar_impl.cc:122-125), between batches"""
        removal, record_len = getattr(self, "_bg", (False, 0))
        self.set_background(removal, background_record, record_len)

    def share_background(self, owner):
        self.ctx.check(self.ctx.lib.jrc_chain_share_background(self.h, owner.h))

    def background_size(self):
        return self.ctx.check(self.ctx.lib.jrc_chain_background_size(self.h))

    def prime_background(self, frames, n_frames, stream=None):
        """channel estimate + history update only (frames: torch tensor laid out like bufs["frames"])"""
        self.ctx.check(self.ctx.lib.jrc_chain_prime_background_dev(self.h, n_frames, frames.data_ptr(), stream))

    def set_write_map(self, write_map):
        """write_map=False: detect-only mode — no range-angle map is stored, results are bit-identical"""
        self.ctx.check(self.ctx.lib.jrc_chain_set_write_map(self.h, int(write_map)))

    def set_map_format(self, power):
        """power=True: bufs["map"] is the float |z|^2 map [n_frames, NR, NA] (alloc(..., power_map=True)), results bit-identical"""
        self.ctx.check(self.ctx.lib.jrc_chain_set_map_format(self.h, 1 if power else 0))
        self.map_bytes = self.ctx.lib.jrc_chain_map_bytes(self.h)

    def run_td(self, bufs, tx, rx_td, n_frames, cp_len, stream=None):
        """A6 + A7 + A1 fused in front of the chain: tx = torch [n_frames, T, n_items, fft_len, 2] (frequency domain),
        rx_td = torch [n_frames, R, rx_stream_len, 2] time-domain RX streams with cyclic prefixes; asynchronous on `stream`"""
        assert tx.is_contiguous() and rx_td.is_contiguous()
        self.ctx.check(self.ctx.lib.jrc_chain_run_td_dev(self.h, n_frames, tx.data_ptr(), rx_td.data_ptr(), cp_len, rx_td.shape[2],
                                                         bufs["chanest"].data_ptr(),
                                                         bufs["map"].data_ptr() if bufs.get("map") is not None else None,
                                                         bufs["results"].data_ptr(), stream))

    def results(self, bufs, n_frames, stream=None):
        arr = (RaResult * n_frames)()
        self.ctx.check(self.ctx.lib.jrc_chain_fetch_results(self.h, n_frames, bufs["results"].data_ptr(), arr, stream))
        return list(arr)

    def results_begin(self, d_results, n_frames, stream=None):
        """starts the copy of a run's records behind the work queued so far, without blocking the stream (at most two in flight);
        d_results: torch uint8 tensor [n_frames, sizeof(RaResult)] that stays untouched until the matching results_end()"""
        self.ctx.check(self.ctx.lib.jrc_chain_fetch_results_begin(self.h, n_frames, d_results.data_ptr(), stream))

    def results_end(self, into=None):
        """the oldest copy begun: waits for it and returns the completed records (a list; with `into` = a (RaResult * max_frames)() array of
        the caller's, the records are left there and their count is returned: no Python object per record on a hot path)"""
        arr = into if into is not None else (RaResult * self.max_frames)()
        n = C.c_int(0)
        self.ctx.check(self.ctx.lib.jrc_chain_fetch_results_end(self.h, arr, C.byref(n)))
        return n.value if into is not None else list(arr[:n.value])

    def range_doppler(self, bufs, n_frames, interp_doppler=1, stream=None):
        """row D: [n_frames, P, N*Ir, S*Id] complex range-Doppler map of the frames in bufs["frames"] (torch, on device)"""
        import torch
        c = self.cfg
        dev = bufs["frames"].device
        # the work buffer stays with the object: the kernels run on the context's stream (or `stream`), which torch's caching
        # allocator does not know about, so a temporary could be handed out again while they still use it
        key = (n_frames, str(dev))
        if getattr(self, "_rd_work_key", None) != key:
            self._rd_work = torch.empty((n_frames, self.P, c.N_sym, self.NR, 2), dtype=torch.float32, device=dev)
            self._rd_work_key = key
        out = torch.empty((n_frames, self.P, self.NR, c.N_sym * interp_doppler, 2), dtype=torch.float32, device=dev)
        self.ctx.check(self.ctx.lib.jrc_range_doppler_dev(self.ctx.h, C.byref(self.cfg), interp_doppler, n_frames,
                                                         bufs["frames"].data_ptr(), self._rd_work.data_ptr(), out.data_ptr(), stream))
        if stream is None:
            self.ctx.sync()               # `out` is complete on return; with an explicit stream the caller orders its own use
        return out

    def launches_per_run(self, n_frames):
        return self.ctx.check(self.ctx.lib.jrc_chain_launches_per_run(self.h, n_frames))

    def set_timing(self, on):
        self.ctx.check(self.ctx.lib.jrc_chain_set_timing(self.h, int(on)))

    def get_timing(self):
        ms = (C.c_float * 3)()
        n = C.c_int(0)
        self.ctx.check(self.ctx.lib.jrc_chain_get_timing(self.h, ms, C.byref(n)))
        return dict(radar_chanest=ms[0], range_angle_fused=ms[1], ra_finalize=ms[2], launches=n.value)

    def close(self):
        if getattr(self, "h", None) and getattr(getattr(self, "ctx", None), "h", None):   # (Context.close() closes its children before it destroys itself; a context already gone means this object went with it)
            self.ctx.lib.jrc_chain_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


FEED_GRAPH = 1


class ChainFeed:
    """jrc_chain_feed: host-fed pipeline over the radar chain — frames in host memory in, per-frame results (and, on
    request, the first maps of each batch) out, `n_slots` batches in flight on their own streams."""

    def __init__(self, fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, range_bins, angle_bins,
                 noise_discard_range_m, noise_discard_angle_deg, snr_threshold=0.0, power_threshold=0.0,
                 n_items=None, enable_tx_interleave=False, n_slots=3, frames_per_slot=32, maps_per_slot=0,
                 graph=False, ctx=None, devices=None):
        """devices=[0, 1, ...]: one host process feeds several GPUs (jrc_chain_feed_create_multi): `n_slots` slots per device, batch k on
        devices[k % len(devices)], results in submission order"""
        # a multi-device feed owns one context per listed GPU: no default context (on device 0) is opened for it
        self.ctx = None if devices else (ctx or default_context())
        self.lib = self.ctx.lib if self.ctx is not None else load()
        self.h = None
        self.cfg = ChainCfg(fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, int(enable_tx_interleave),
                            n_items if n_items is not None else N_pre + N_sym,
                            noise_discard_range_m, noise_discard_angle_deg, snr_threshold, power_threshold)
        self.P, self.NR, self.NA = N_tx * N_rx, fft_len * interp_range, N_tx * N_rx * interp_angle
        self.n_slots, self.frames_per_slot, self.maps_per_slot = n_slots, frames_per_slot, maps_per_slot
        rb = np.ascontiguousarray(range_bins, np.float32)
        ab = np.ascontiguousarray(angle_bins, np.float32)
        assert len(rb) == self.NR and len(ab) == self.NA
        h = _vp()
        L = self.lib
        if devices:
            dv = (C.c_int * len(devices))(*devices)
            st = L.jrc_chain_feed_create_multi(dv, len(devices), C.byref(self.cfg), _ptr(rb), _ptr(ab), n_slots, frames_per_slot,
                                               maps_per_slot, FEED_GRAPH if graph else 0, C.byref(h))
            if st != JRC_OK:
                raise JrcError(st, L.jrc_strerror(st).decode())
            self.n_slots = n_slots * len(devices)
        else:
            self.ctx.check(L.jrc_chain_feed_create(self.ctx.h, C.byref(self.cfg), _ptr(rb), _ptr(ab), n_slots, frames_per_slot,
                                                   maps_per_slot, FEED_GRAPH if graph else 0, C.byref(h)))
        self.h = h
        if self.ctx is not None:
            self.ctx._adopt(self)
        self.frame_bytes = L.jrc_chain_feed_frame_bytes(h)
        self.map_bytes = L.jrc_chain_feed_map_bytes(h)

    def _check(self, st):
        """negative status -> JrcError carrying the FEED's last error (its contexts are its own: jrc_chain_feed_last_error)"""
        if st < 0:
            msg = self.lib.jrc_chain_feed_last_error(self.h).decode() or self.lib.jrc_strerror(st).decode()
            if st == JRC_ERR_LENGTH_MISMATCH:
                raise RuntimeError(msg)       # the same mapping as Context.check
            if st == JRC_ERR_INVALID_ARG:
                raise ValueError(msg)
            raise JrcError(st, msg)
        return st

    def frame_shape(self):
        c = self.cfg
        return (c.N_tx + c.N_rx, c.n_items, c.fft_len)

    def acquire(self):
        """the next slot's pinned staging as a numpy complex64 array [frames_per_slot, T+R, n_items, fft_len] to fill in place"""
        p = _vp()
        self._check(self.lib.jrc_chain_feed_acquire(self.h, C.byref(p)))
        n = self.frames_per_slot * self.frame_bytes // 8
        buf = (C.c_float * (2 * n)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.complex64).reshape((self.frames_per_slot,) + self.frame_shape())

    def submit(self, frames=None, n_frames=None, rx_only=False):
        """frames: complex64 [n, T+R, n_items, fft_len] in host memory, or None after acquire() + in-place fill.
        rx_only: the frames' TX ports equal the rows given to set_tx(): only their receive ports are uploaded (jrc_chain_feed_submit_rx)"""
        fn = self.lib.jrc_chain_feed_submit_rx if rx_only else self.lib.jrc_chain_feed_submit
        if frames is None:
            n = self.frames_per_slot if n_frames is None else n_frames
            self._check(fn(self.h, None, n))
            return
        fr = np.ascontiguousarray(frames, np.complex64)
        n = fr.shape[0] if n_frames is None else n_frames
        assert fr.size * 8 >= n * self.frame_bytes
        self._check(fn(self.h, _ptr(fr), n))

    def set_tx(self, tx):
        """tx: complex64 [T, n_items, fft_len] = the reference ports every following rx_only frame shares (None: off)"""
        if tx is None:
            self._check(self.lib.jrc_chain_feed_set_tx(self.h, None))
            return
        t = np.ascontiguousarray(tx, np.complex64)
        c = self.cfg
        assert t.shape == (c.N_tx, c.n_items, c.fft_len)
        self._check(self.lib.jrc_chain_feed_set_tx(self.h, _ptr(t)))

    def poll(self):
        """True when collect() would not block"""
        return self.lib.jrc_chain_feed_poll(self.h) == 1

    def submit_many(self, batches):
        """batches: list of complex64 arrays [n_k, T+R, n_items, fft_len]; one per free slot at most; staged and enqueued by the per-device
        host threads in parallel"""
        keep = [np.ascontiguousarray(b, np.complex64) for b in batches]
        ptrs = (_vp * len(keep))(*[_ptr(b) for b in keep])
        ns = (C.c_int * len(keep))(*[b.shape[0] for b in keep])
        self._check(self.lib.jrc_chain_feed_submit_many(self.h, ptrs, ns, len(keep)))

    def n_devices(self):
        return self.lib.jrc_chain_feed_n_devices(self.h)

    def collect(self, want_maps=False):
        """oldest batch in flight -> (list of RaResult, maps or None); ([], None) when nothing is in flight"""
        arr = (RaResult * self.frames_per_slot)()
        n = C.c_int(0)
        maps = None
        if want_maps and self.maps_per_slot:
            maps = np.empty((self.maps_per_slot, self.NR, self.NA), np.complex64)
        r = self.lib.jrc_chain_feed_collect(self.h, arr, _ptr(maps) if maps is not None else None, C.byref(n))
        self._check(min(r, 0))
        if maps is not None:
            maps = maps[:min(n.value, self.maps_per_slot)]
        return list(arr[:n.value]), maps

    def set_background(self, background_removal, background_recording, record_len):
        self._check(self.lib.jrc_chain_feed_set_background(self.h, int(background_removal), int(background_recording), record_len))

    def set_write_map(self, write_map):
        self._check(self.lib.jrc_chain_feed_set_write_map(self.h, int(write_map)))

    def pending(self):
        return self.lib.jrc_chain_feed_pending(self.h)

    def stats(self):
        g, d = C.c_long(0), C.c_long(0)
        self._check(self.lib.jrc_chain_feed_stats(self.h, C.byref(g), C.byref(d)))
        return dict(graph_replays=g.value, direct_submits=d.value)

    def close(self):
        if getattr(self, "h", None) and (self.ctx is None or getattr(self.ctx, "h", None)):   # multi-device feeds own their contexts
            self.lib.jrc_chain_feed_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------
# comm side: C1 mimo_ofdm_equalizer, C2 mimo_precoder, C3 steering
# ---------------------------------------------------------------------------------------------------------
LS, STA = 0, 1                      # ChannelEstimator (include/mimo_ofdm_jrc/mimo_ofdm_equalizer.h:27-30)
NDP, DATA = 1, 2                    # PACKET_TYPE (include/mimo_ofdm_jrc/stream_encoder.h)
_i32p = C.POINTER(C.c_int32)


class EqCfg(C.Structure):
    """jrc_eq_cfg"""
    _fields_ = [("estimator", C.c_int32), ("freq", C.c_double), ("bw", C.c_double),
                ("fft_len", C.c_int32), ("cp_len", C.c_int32), ("n_data", C.c_int32), ("n_pilot", C.c_int32),
                ("data_carriers", _i32p), ("pilot_carriers", _i32p), ("n_pilot_rows", C.c_int32),
                ("pilot_symbols", _vp), ("ltf_seq", _vp), ("mapped_ltf", _vp),
                ("mapped_cols", C.c_int32), ("n_mimo_ltf", C.c_int32)]


class EqEvent(C.Structure):
    """jrc_eq_event"""
    _fields_ = [("kind", C.c_int32), ("n_chan_mean", C.c_int32), ("offset", C.c_int64),
                ("data_bytes", C.c_uint64), ("mcs", C.c_uint64), ("packet_type", C.c_uint64),
                ("snr", C.c_double), ("freq_offset", C.c_double), ("snr_data", C.c_double),
                ("chan_mean", C.c_float * 32)]


class PreCfg(C.Structure):
    """jrc_pre_cfg"""
    _fields_ = [("fft_len", C.c_int32), ("N_tx", C.c_int32), ("n_data", C.c_int32), ("n_pilot", C.c_int32),
                ("data_carriers", _i32p), ("pilot_carriers", _i32p), ("n_pilot_rows", C.c_int32),
                ("pilot_symbols", _vp), ("n_sync", C.c_int32), ("sync_words", _vp), ("mapped_ltf", _vp)]


def _load_comm():
    L = load()
    if not getattr(L, "_comm_ready", False):
        L.jrc_equalizer_create.argtypes = [_vp, C.POINTER(EqCfg), C.c_int, C.POINTER(_vp)]
        L.jrc_equalizer_destroy.argtypes = [_vp]
        L.jrc_equalizer_set_estimator.argtypes = [_vp, C.c_int]
        L.jrc_equalizer_set_bandwidth.argtypes = [_vp, C.c_double]
        L.jrc_equalizer_set_frequency.argtypes = [_vp, C.c_double]
        L.jrc_equalizer_work.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                         C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(EqEvent), C.c_int,
                                         C.POINTER(C.c_int), _vp, C.POINTER(C.c_int)]
        L.jrc_equalizer_frames_dev.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]
        L.jrc_steering_from_channel.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_int, _vp]
        L.jrc_dft_matrix.argtypes = [_vp, C.c_int, _vp]
        L.jrc_precoder_create.argtypes = [_vp, C.POINTER(PreCfg), C.POINTER(_vp)]
        L.jrc_precoder_destroy.argtypes = [_vp]
        L.jrc_precoder_output_length.argtypes = [_vp, C.c_int]
        L.jrc_precoder_work.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.POINTER(_vp)]
        L.jrc_precoder_frames_dev.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
        L.jrc_n_ofdm_sym.argtypes = [C.c_int, C.c_int, C.c_int]
        L.jrc_sig_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _cfp]
        L._comm_ready = True
    return L


def n_ofdm_sym(mcs, n_data_carriers, nbytes):
    return _load_comm().jrc_n_ofdm_sym(mcs, n_data_carriers, nbytes)


def sig_encode(n_data_carriers, mcs, packet_type, length):
    """generate_signal_field (lib/mimo_precoder_impl.cc:985-1060): the BPSK SIG symbol on the data carriers (real parts)"""
    out = np.zeros(n_data_carriers, np.float32)
    st = _load_comm().jrc_sig_encode(n_data_carriers, mcs, packet_type, length, out.ctypes.data_as(_cfp))
    if st < 0:
        raise JrcError(st, "jrc_sig_encode")
    return out


def _event_dict(e):
    d = dict(kind=e.kind, offset=e.offset)
    if e.kind == 1:
        d.update(data_bytes=e.data_bytes, mcs=e.mcs, packet_type=e.packet_type, snr=e.snr, freq_offset=e.freq_offset)
    else:
        d.update(snr_data=e.snr_data, chan_mean=np.array(e.chan_mean[:2 * e.n_chan_mean], np.float32).view(np.complex64))
    return d


class mimo_ofdm_equalizer:
    """include/mimo_ofdm_jrc/mimo_ofdm_equalizer.h:64-78; general_work = lib/mimo_ofdm_equalizer_impl.cc:191-648.
    Tags come back as event dicts (stream_start / stream_end); chan_est is what the reference writes to
    chan_est_file for an NDP frame."""

    def __init__(self, estimator_algo, freq, bw, fft_len, cp_len, data_carriers, pilot_carriers, pilot_symbols,
                 long_seq, mapped_ltf_symbols, n_mimo_ltf, chan_est_file="", comm_log_file="", stats_record=False,
                 debug=False, n_streams=1, ctx=None):
        self.ctx = ctx or default_context()
        L = _load_comm()
        self._keep = dict(dc=np.ascontiguousarray(data_carriers, np.int32), pc=np.ascontiguousarray(pilot_carriers, np.int32),
                          ps=_c64(pilot_symbols), ltf=_c64(long_seq), ml=_c64(mapped_ltf_symbols))
        k = self._keep
        self.fft_len, self.n_data, self.n_streams = fft_len, len(k["dc"]), n_streams
        self.n_tx = k["ml"].shape[1] // n_mimo_ltf
        cfg = EqCfg(int(estimator_algo), freq, bw, fft_len, cp_len, len(k["dc"]), len(k["pc"]),
                    k["dc"].ctypes.data_as(_i32p), k["pc"].ctypes.data_as(_i32p), k["ps"].shape[0],
                    _ptr(k["ps"]), _ptr(k["ltf"]), _ptr(k["ml"]), k["ml"].shape[1], n_mimo_ltf)
        h = _vp()
        self.ctx.check(L.jrc_equalizer_create(self.ctx.h, C.byref(cfg), n_streams, C.byref(h)))
        self.h = h
        self.ctx._adopt(self)

    def set_estimator(self, algo):
        self.ctx.check(self.ctx.lib.jrc_equalizer_set_estimator(self.h, int(algo)))

    def set_bandwidth(self, bw):
        self.ctx.check(self.ctx.lib.jrc_equalizer_set_bandwidth(self.h, float(bw)))

    def set_frequency(self, freq):
        self.ctx.check(self.ctx.lib.jrc_equalizer_set_frequency(self.h, float(freq)))

    def general_work(self, symbols, frame_start_tags=(), noutput_items=None, stream=0):
        x = _c64(symbols).reshape(-1, self.fft_len)
        nin = x.shape[0]
        nout = nin if noutput_items is None else noutput_items
        out = np.zeros((max(nout, 1), self.n_data), np.complex64)
        nt = len(frame_start_tags)
        offs = (C.c_int64 * max(1, nt))(*[int(t[0]) for t in frame_start_tags])
        vals = (C.c_double * max(1, nt))(*[float(t[1]) for t in frame_start_tags])
        cons, nev, cw = C.c_int(), C.c_int(), C.c_int()
        ev = (EqEvent * 8)()
        ce = np.zeros((self.fft_len, self.n_tx), np.complex64)
        n = self.ctx.check(self.ctx.lib.jrc_equalizer_work(self.h, stream, nout, nin, _ptr(x), offs, vals, nt, _ptr(out),
                                                           C.byref(cons), ev, 8, C.byref(nev), _ptr(ce), C.byref(cw)))
        return dict(out=out[:n], consumed=cons.value, events=[_event_dict(e) for e in ev[:nev.value]],
                    chan_est=ce if cw.value else None)

    def frames_dev(self, d_in, d_phase, n_symbols, max_out, stream=None):
        """batched device-resident frames (torch tensors): d_in [n_streams, n_symbols, fft_len, 2] f32, d_phase [n_streams] f64"""
        import torch
        ns = d_in.shape[0]
        out = torch.empty((ns, max_out, self.n_data, 2), dtype=torch.float32, device=d_in.device)
        n_out = torch.empty((ns,), dtype=torch.int32, device=d_in.device)
        ev = torch.empty((ns, 2, C.sizeof(EqEvent)), dtype=torch.uint8, device=d_in.device)   # the kernel defines every slot (kind 0 = unused)
        self.ctx.check(self.ctx.lib.jrc_equalizer_frames_dev(self.h, ns, n_symbols, d_in.data_ptr(), d_phase.data_ptr(), max_out,
                                                             out.data_ptr(), n_out.data_ptr(), ev.data_ptr(), stream))
        return out, n_out, ev

    def close(self):
        if getattr(self, "h", None) and getattr(getattr(self, "ctx", None), "h", None):   # (Context.close() closes its children before it destroys itself; a context already gone means this object went with it)
            self.ctx.lib.jrc_equalizer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def steering_from_channel(h, phased=False, ctx=None):
    """h: [n, T] (or [T]) channel rows -> Q[n, t, j] (lib/mimo_precoder_impl.cc:846-861)"""
    ctx = ctx or default_context()
    h = _c64(h)
    single = h.ndim == 1
    h2 = h.reshape(-1, h.shape[-1])
    n, T = h2.shape
    q = np.zeros((n, T * T), np.complex64)
    ctx.check(_load_comm().jrc_steering_from_channel(ctx.h, T, n, _ptr(h2), int(phased), _ptr(q)))
    Q = np.transpose(q.reshape(n, T, T), (0, 2, 1)).copy()        # stored column-major
    return Q[0] if single else Q


class mimo_precoder:
    """include/mimo_ofdm_jrc/mimo_precoder.h; work = lib/mimo_precoder_impl.cc:275-741.  The steering-matrix source
    (CSV / radar log parsing, file mtime cache) stays in the host wrapper; work() takes the matrices explicitly."""

    def __init__(self, fft_len, N_tx, N_ss, data_carriers, pilot_carriers, pilot_symbols, sync_words, mapped_ltf_symbols,
                 chan_est_file="", chan_est_smoothing=False, radar_log_file="", radar_aided=False, phased_steering=False,
                 use_radar_streams=False, len_tag_key="packet_len", debug=False, ctx=None):
        self.ctx = ctx or default_context()
        L = _load_comm()
        self._keep = dict(dc=np.ascontiguousarray(data_carriers, np.int32), pc=np.ascontiguousarray(pilot_carriers, np.int32),
                          ps=_c64(pilot_symbols), sw=_c64(sync_words), ml=_c64(mapped_ltf_symbols))
        k = self._keep
        if any(len(w) != fft_len for w in k["sw"]):
            raise ValueError("[MIMO PRECODER] sync words must be fft length")
        if k["ml"].shape != (fft_len, N_tx * N_tx):
            raise ValueError("[MIMO PRECODER] MIMO LTF symbols should have (fft length x Ntx) rows!!")
        if k["ps"].shape[1] != len(k["pc"]):
            raise ValueError("pilot_carriers do not match pilot_symbols")
        self.N, self.T, self.n_sync, self.n_data = fft_len, N_tx, k["sw"].shape[0], len(k["dc"])
        cfg = PreCfg(fft_len, N_tx, len(k["dc"]), len(k["pc"]), k["dc"].ctypes.data_as(_i32p), k["pc"].ctypes.data_as(_i32p),
                     k["ps"].shape[0], _ptr(k["ps"]), k["sw"].shape[0], _ptr(k["sw"]), _ptr(k["ml"]))
        h = _vp()
        self.ctx.check(L.jrc_precoder_create(self.ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h
        self.ctx._adopt(self)

    def calculate_output_stream_length(self, ninput_items):
        return self.ctx.lib.jrc_precoder_output_length(self.h, ninput_items)

    def work(self, symbols, mcs, packet_type, pdu_len, steer_mode=0, Q_mean=None, Q_sc=None, radar_streams=None):
        x = _c64(symbols).ravel()
        n_sym = x.size // self.n_data
        n_total = n_sym + self.n_sync + self.T + 1
        out = np.zeros((self.T, n_total, self.N), np.complex64)
        ptrs = (_vp * self.T)(*[_ptr(out[t]) for t in range(self.T)])
        qm = None if Q_mean is None else np.ascontiguousarray(_c64(Q_mean).T)
        qs = None if Q_sc is None else np.ascontiguousarray(np.transpose(_c64(Q_sc), (0, 2, 1)))
        rs = None if radar_streams is None else _c64(radar_streams)
        st = self.ctx.lib.jrc_precoder_work(self.h, x.size, _ptr(x), mcs, packet_type, pdu_len, steer_mode,
                                            None if qm is None else _ptr(qm), None if qs is None else _ptr(qs),
                                            None if rs is None else _ptr(rs), ptrs)
        if st == -8:
            raise RuntimeError(self.ctx.lib.jrc_last_error(self.ctx.h).decode())   # std::runtime_error (:327-333)
        self.ctx.check(st)
        assert st == n_total
        return out

    def frames_dev(self, d_in, mcs, packet_type, pdu_len, steer_mode=0, d_Q_mean=None, d_Q_sc=None, d_radar_streams=None, d_out=None,
                   stream=None):
        """batched, device-resident: d_in torch float32 [n_frames, n_sym*n_data, 2]; Q matrices torch, already column-major per
        matrix ([T*T, 2] / [fft_len, T*T, 2]); returns torch [n_frames, T, n_total, fft_len, 2] (asynchronous on `stream`)"""
        import torch
        F, nin = d_in.shape[0], d_in.shape[1]
        n_total = nin // self.n_data + self.n_sync + self.T + 1
        if d_out is None:
            d_out = torch.empty((F, self.T, n_total, self.N, 2), dtype=torch.float32, device=d_in.device)
        ptr = lambda t: None if t is None else t.data_ptr()
        st = self.ctx.lib.jrc_precoder_frames_dev(self.h, F, nin, d_in.data_ptr(), mcs, packet_type, pdu_len, steer_mode, ptr(d_Q_mean),
                                                  ptr(d_Q_sc), ptr(d_radar_streams), d_out.data_ptr(), stream)
        if st == -8:
            raise RuntimeError(self.ctx.lib.jrc_last_error(self.ctx.h).decode())   # std::runtime_error (:327-333)
        self.ctx.check(st)
        return d_out

    def close(self):
        if getattr(self, "h", None) and getattr(getattr(self, "ctx", None), "h", None):   # (Context.close() closes its children before it destroys itself; a context already gone means this object went with it)
            self.ctx.lib.jrc_precoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------
# target_simulator on the device
class TsimCfg(C.Structure):
    _fields_ = [("n_targets", C.c_int), ("range", _cfp), ("velocity", _cfp), ("rcs", _cfp), ("azimuth", _cfp),
                ("n_rx", C.c_int), ("position_rx", _cfp), ("samp_rate", C.c_int), ("center_freq", C.c_float),
                ("self_coupling_db", C.c_float), ("rndm_phaseshift", C.c_int), ("self_coupling", C.c_int),
                ("sum_targets", C.c_int), ("max_bursts", C.c_int)]


def _load_tsim():
    L = load()
    if not getattr(L, "_tsim_ready", False):
        L.jrc_tsim_create.restype = _vp
        L.jrc_tsim_create.argtypes = [_vp, C.POINTER(TsimCfg)]
        L.jrc_tsim_destroy.argtypes = [_vp]
        L.jrc_tsim_destroy.restype = None
        L.jrc_tsim_set_targets.argtypes = [_vp, C.c_int, _cfp, _cfp, _cfp, _cfp]
        L.jrc_tsim_work.argtypes = [_vp, _vp, C.c_int, C.POINTER(_vp), _vp]
        L.jrc_tsim_run_dev.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp]
        L.jrc_tsim_run_sum_dev.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.POINTER(_vp), _vp, C.POINTER(_vp), C.c_int, _vp]
        L.jrc_tsim_burst_capacity.argtypes = [_vp]
        L._tsim_ready = True
    return L


def _f32(v):
    return np.ascontiguousarray(np.atleast_1d(v), dtype=np.float32)


class target_simulator:
    """include/mimo_ofdm_jrc/target_simulator.h make(range, velocity, rcs, azimuth, position_rx, samp_rate, center_freq,
    self_coupling_db, rndm_phaseshift, self_coupling, len_key, debug); work = lib/target_simulator_impl.cc:202-385.
    One input stream, len(position_rx) output streams.  `sum_targets` (not in the reference) accumulates the targets
    instead of letting the last one overwrite the others; `max_bursts` sizes the batched device form run_dev()."""

    def __init__(self, range, velocity, rcs, azimuth, position_rx, samp_rate, center_freq, self_coupling_db=-40.0,
                 rndm_phaseshift=False, self_coupling=False, len_key="packet_len", debug=False, sum_targets=False,
                 max_bursts=1, seed=None, ctx=None):
        self.ctx = ctx or default_context()
        L = _load_tsim()
        self._keep = [_f32(v) for v in (range, velocity, rcs, azimuth, position_rx)]
        r, v, s, a, p = self._keep
        if not (r.size == v.size == s.size == a.size):
            raise ValueError("[TARGET SIM] range, velocity, rcs and azimuth must have the same length")
        self.K, self.R = int(r.size), int(p.size)
        self.samp_rate = int(samp_rate)
        self.rndm_phaseshift = bool(rndm_phaseshift)
        cfg = TsimCfg(self.K, r.ctypes.data_as(_cfp), v.ctypes.data_as(_cfp), s.ctypes.data_as(_cfp), a.ctypes.data_as(_cfp),
                      self.R, p.ctypes.data_as(_cfp), self.samp_rate, float(center_freq), float(self_coupling_db),
                      int(self.rndm_phaseshift), int(bool(self_coupling)), int(bool(sum_targets)), int(max_bursts))
        self.h = L.jrc_tsim_create(self.ctx.h, C.byref(cfg))
        self.ctx._adopt(self)
        if not self.h:
            raise ValueError(self.ctx.lib.jrc_last_error(self.ctx.h).decode())
        self._rng = np.random.default_rng(seed)        # the reference seeds std::rand with time(NULL) (:196)
        self.nitems_written = 0

    def setup_targets(self, range, velocity, rcs, azimuth):
        k = [_f32(v) for v in (range, velocity, rcs, azimuth)]
        self.ctx.check(self.ctx.lib.jrc_tsim_set_targets(self.h, int(k[0].size), *[x.ctypes.data_as(_cfp) for x in k]))
        self._keep[:4] = k
        self.K = int(k[0].size)

    def calculate_output_stream_length(self, ninput_items):
        return ninput_items

    def draw_phases(self):
        """:316-321 — exp(j 2 pi ((rand() % 1000 + 1) / 1000)) per target"""
        u = (self._rng.integers(0, 1000, self.K) + 1) / 1000.0
        ang = (2 * np.pi * u.astype(np.float32)).astype(np.float32)
        return (np.cos(ang) + 1j * np.sin(ang)).astype(np.complex64)

    def rx_time_tag(self):
        """(:331-335) value of the rx_time tag of the burst about to be produced: (uint64 secs, double frac)"""
        secs = self.nitems_written // self.samp_rate
        frac = float(np.float32(self.nitems_written) / np.float32(self.samp_rate)) - secs
        return int(secs), frac

    def work(self, x, target_phase=None):
        x = _c64(x).ravel()
        n = x.size
        out = np.zeros((self.R, n), np.complex64)
        if n == 0:
            return out
        if self.rndm_phaseshift and target_phase is None:
            target_phase = self.draw_phases()
        tp = None if target_phase is None else _c64(target_phase)
        ptrs = (_vp * self.R)(*[_ptr(out[l]) for l in range(self.R)])
        st = self.ctx.check(self.ctx.lib.jrc_tsim_work(self.h, _ptr(x), n, ptrs, None if tp is None else _ptr(tp)))
        self.nitems_written += n
        return out

    def run_dev(self, d_in, d_out, n_bursts, n_input, target_phase=None, accumulate_out=False, stream=None):
        """d_in: torch complex64 [n_bursts][n_input]; d_out: [n_bursts][n_rx][n_input] (device tensors)"""
        tp = None if target_phase is None else _c64(target_phase)
        self.ctx.check(self.ctx.lib.jrc_tsim_run_dev(self.h, n_bursts, n_input, _vp(d_in.data_ptr()), _vp(d_out.data_ptr()),
                                                     None if tp is None else _ptr(tp), int(accumulate_out), stream))

    @staticmethod
    def run_sum_dev(sims, d_ins, d_out, n_bursts, n_input, target_phases=None, accumulate_out=False, stream=None):
        """jrc_tsim_run_sum_dev: the simulators of a flowgraph's TX ports, their outputs summed per RX antenna on the spectrum, in one pass.
        d_ins: one device tensor [n_bursts][n_input] per simulator; raises JrcError(JRC_ERR_UNSUPPORTED) when the burst length does not take
        the direct route (run them one by one with accumulate_out then)."""
        ctx = sims[0].ctx
        hs = (_vp * len(sims))(*[s.h for s in sims])
        ins = (_vp * len(sims))(*[_vp(t.data_ptr()) for t in d_ins])
        keep = [None if (target_phases is None or p is None) else _c64(p) for p in (target_phases or [None] * len(sims))]
        phs = (_vp * len(sims))(*[None if k is None else k.ctypes.data_as(_vp) for k in keep]) if target_phases is not None else None
        ctx.check(ctx.lib.jrc_tsim_run_sum_dev(hs, len(sims), n_bursts, n_input, ins, _vp(d_out.data_ptr()), phs, int(accumulate_out), stream))

    def close(self):
        if getattr(self, "h", None) and getattr(getattr(self, "ctx", None), "h", None):   # (Context.close() closes its children before it destroys itself; a context already gone means this object went with it)
            self.ctx.lib.jrc_tsim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------
# bit codec (stream_encoder / stream_decoder) on the device
_u8p = C.POINTER(C.c_uint8)
MAX_PAYLOAD_SIZE = 3100                     # lib/utils.h


def _load_codec():
    L = load()
    if not getattr(L, "_codec_ready", False):
        L.jrc_stream_n_ofdm_sym.argtypes = [C.c_int, C.c_int, C.c_int]
        L.jrc_stream_encode.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int]
        L.jrc_stream_encode_dev.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_long, _vp, _vp, _vp, C.c_long, _vp, _vp]
        L.jrc_stream_decode.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.POINTER(C.c_int)]
        L.jrc_stream_decode_dev.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_long, _vp, _vp, _vp, C.c_long, _vp, _vp]
        L._codec_ready = True
    return L


def stream_n_ofdm_sym(mcs, n_data_carriers, data_bytes):
    """packet_param::n_ofdm_sym for a frame of `data_bytes` (payload + CRC) bytes (lib/utils.cc:79-111); < 0 = not a valid frame"""
    return _load_codec().jrc_stream_n_ofdm_sym(int(mcs), int(n_data_carriers), int(data_bytes))


class stream_encoder:
    """include/mimo_ofdm_jrc/stream_encoder.h:62 make(mod_encode, data_len, N_ss_radar, debug); general_work =
    lib/stream_encoder_impl.cc:76-270.  PDUs arrive on the `pdu_in` message port in the reference; here work(pdu) takes the
    bytes and returns the symbols plus the four tags the block attaches (packet_len, packet_type, mcs, pdu_len)."""

    def __init__(self, mod_encode, data_len, N_ss_radar=0, debug=False, ctx=None):
        self.ctx = ctx or default_context()
        self.L = _load_codec()
        if not 0 <= int(mod_encode) <= 5:
            raise ValueError("wrong encoding")
        self.mcs, self.data_len = int(mod_encode), int(data_len)
        self.d_scrambler = 1                                   # :53

    def set_mcs(self, mod_encode):
        if not 0 <= int(mod_encode) <= 5:
            raise ValueError("wrong encoding")
        self.mcs = int(mod_encode)

    def work(self, pdu):
        p = np.frombuffer(bytes(pdu), np.uint8).copy() if not isinstance(pdu, np.ndarray) else np.ascontiguousarray(pdu, np.uint8)
        if p.size + 4 > MAX_PAYLOAD_SIZE:                      # printed and dropped (:139-143); the scrambler seed is not advanced
            return None, None
        ns = self.L.jrc_stream_n_ofdm_sym(self.mcs, self.data_len, p.size + 4)
        out = np.zeros(ns * self.data_len, np.complex64)
        n = self.ctx.check(self.L.jrc_stream_encode(self.ctx.h, self.mcs, self.data_len, _ptr(p) if p.size else None, p.size,
                                                    self.d_scrambler, _ptr(out), out.size))
        assert n == out.size
        self.d_scrambler += 1                                  # scramble(..., d_scrambler++) (:171-175)
        if self.d_scrambler > 127:
            self.d_scrambler = 1
        tags = dict(packet_len=n, packet_type=int(p[0]) if p.size else 0, mcs=self.mcs, pdu_len=p.size + 4)
        return out, tags

    def encode_dev(self, d_psdu, psdu_stride, d_len, d_scrambler, d_out, sym_stride, d_n_sym, n_frames, stream=None):
        self.ctx.check(self.L.jrc_stream_encode_dev(self.ctx.h, self.mcs, self.data_len, n_frames, _vp(d_psdu.data_ptr()), psdu_stride,
                                                    _vp(d_len.data_ptr()), _vp(d_scrambler.data_ptr()), _vp(d_out.data_ptr()),
                                                    sym_stride, _vp(d_n_sym.data_ptr()), stream))


class stream_decoder:
    """include/mimo_ofdm_jrc/stream_decoder.h:49 make(n_data_carriers, comm_log_file, stats_record, debug); decode() =
    lib/stream_decoder_impl.cc:205-405.  work(symbols, stream_start) decodes one frame described by the equalizer's
    stream_start dictionary (data_bytes, mcs, packet_type, snr) and returns what the block publishes on `sym`:
    (crc_ok, payload bytes); the rolling packet-error rate over 25 frames (:64, :186) is kept in .per."""

    def __init__(self, n_data_carriers, comm_log_file="", stats_record=False, debug=False, ctx=None):
        self.ctx = ctx or default_context()
        self.L = _load_codec()
        self.n_data_carriers = int(n_data_carriers)
        self._per = []

    @property
    def per(self):
        w = self._per[-25:]
        return 100.0 * (sum(w) / len(w)) if w else 0.0

    def work(self, symbols, stream_start):
        mcs, nbytes = int(stream_start["mcs"]), int(stream_start["data_bytes"])
        s = _c64(symbols).ravel()
        out = np.zeros(max(nbytes, 8), np.uint8)
        ok = C.c_int(0)
        n = self.L.jrc_stream_decode(self.ctx.h, mcs, self.n_data_carriers, nbytes, _ptr(s), s.size, _ptr(out), C.byref(ok))
        if n == JRC_ERR_UNSUPPORTED:
            return None, None                                  # frame refused, nothing is decoded (:133-146)
        self.ctx.check(n)
        self._per.append(0 if ok.value else 1)                 # per_stats(0|1) (:283, :322)
        return bool(ok.value), out[:n].tobytes()

    def decode_dev(self, d_sym, sym_stride, d_mcs, d_data_bytes, d_payload, payload_stride, d_status, n_frames, stream=None):
        self.ctx.check(self.L.jrc_stream_decode_dev(self.ctx.h, self.n_data_carriers, n_frames, _vp(d_sym.data_ptr()), sym_stride,
                                                    _vp(d_mcs.data_ptr()), _vp(d_data_bytes.data_ptr()), _vp(d_payload.data_ptr()),
                                                    payload_stride, _vp(d_status.data_ptr()), stream))


# ---------------------------------------------------------------------------------------------------------------
# sync front-end (moving_avg, frame_detector, frame_sync) on the device
_u64p = C.POINTER(C.c_uint64)
_dblp = C.POINTER(C.c_double)


def _load_sync():
    L = load()
    if not getattr(L, "_sync_ready", False):
        L.jrc_moving_avg.argtypes = [_vp, C.c_int, C.c_float, C.c_int, C.c_int, _vp, _vp]
        L.jrc_moving_avg_dev.argtypes = [_vp, C.c_int, C.c_float, C.c_int, _vp, _vp, _vp]
        L.jrc_sync_metrics_dev.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp, _vp, _vp]
        L.jrc_frame_detector_create.restype = _vp
        L.jrc_frame_detector_create.argtypes = [_vp, C.c_int, C.c_int, C.c_double, C.c_uint, C.c_uint]
        L.jrc_frame_detector_destroy.argtypes = [_vp]
        L.jrc_frame_detector_destroy.restype = None
        L.jrc_frame_detector_work.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(C.c_int), _u64p, _dblp, C.c_int,
                                              C.POINTER(C.c_int)]
        L.jrc_frame_sync_create.restype = _vp
        L.jrc_frame_sync_create.argtypes = [_vp, C.c_int, C.c_int, C.c_uint, _vp, C.c_int]
        L.jrc_frame_sync_destroy.argtypes = [_vp]
        L.jrc_frame_sync_destroy.restype = None
        L.jrc_frame_sync_work.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _u64p, _dblp, C.c_int, _vp, C.POINTER(C.c_int),
                                          _u64p, _dblp, C.POINTER(C.c_int)]
        L.jrc_frame_sync_state.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L._sync_ready = True
    return L


class moving_avg:
    """include/mimo_ofdm_jrc/moving_avg.h make(length, scale, max_iter, debug); work = lib/moving_avg_impl.cc:62-98.  A sync block
    with history length-1: work(x) takes the new items and keeps the history between calls."""

    def __init__(self, length, scale=1.0, max_iter=4096, debug=False, ctx=None):
        self.ctx = ctx or default_context()
        self.L = _load_sync()
        self.length, self.scale, self.max_iter = int(length), float(scale), int(max_iter)
        self._hist = np.zeros(self.length - 1, np.complex64)

    def set_length_and_scale(self, length, scale):
        self.length, self.scale = int(length), float(scale)
        self._hist = np.zeros(self.length - 1, np.complex64)

    def work(self, x):
        x = _c64(x).ravel()
        buf = np.concatenate([self._hist, x]).astype(np.complex64)
        out = np.zeros(x.size, np.complex64)
        n = self.ctx.check(self.L.jrc_moving_avg(self.ctx.h, self.length, self.scale, self.max_iter, x.size, _ptr(buf), _ptr(out)))
        if n and self.length > 1:
            self._hist = buf[n:n + self.length - 1].copy()     # the scheduler keeps the last length-1 consumed items as history
        return out[:n]


def sync_metrics(x, delay, window, pwindow, pscale, ctx=None):
    """the detector's three input streams for a capture (stock blocks of the comm flowgraph): (x delayed, in_abs, in_cor)"""
    import torch
    ctx = ctx or default_context()
    L = _load_sync()
    x = _c64(x).ravel()
    d_x = torch.from_numpy(x.view(np.float32).copy()).cuda()
    d_xd, d_ia = torch.empty_like(d_x), torch.empty_like(d_x)
    d_ic = torch.empty(x.size, dtype=torch.float32, device="cuda")
    ctx.check(L.jrc_sync_metrics_dev(ctx.h, x.size, delay, window, pwindow, float(pscale), _vp(d_x.data_ptr()), _vp(d_xd.data_ptr()),
                                     _vp(d_ia.data_ptr()), _vp(d_ic.data_ptr()), None))
    ctx.sync()
    return d_xd.cpu().numpy().view(np.complex64), d_ia.cpu().numpy().view(np.complex64), d_ic.cpu().numpy()


class frame_detector:
    """include/mimo_ofdm_jrc/frame_detector.h make(fft_len, cp_len, threshold, min_n_peaks, ignore_gap, debug); general_work =
    lib/frame_detector_impl.cc:70-205.  work() is one general_work call; run() offers everything until nothing moves."""

    def __init__(self, fft_len, cp_len, threshold, min_n_peaks, ignore_gap, debug=False, ctx=None):
        self.ctx = ctx or default_context()
        self.L = _load_sync()
        self.h = self.L.jrc_frame_detector_create(self.ctx.h, fft_len, cp_len, float(threshold), int(min_n_peaks), int(ignore_gap))
        self.ctx._adopt(self)
        if not self.h:
            raise ValueError(self.ctx.lib.jrc_last_error(self.ctx.h).decode())

    def work(self, x, in_abs, in_cor, noutput):
        x, in_abs = _c64(x).ravel(), _c64(in_abs).ravel()
        in_cor = np.ascontiguousarray(in_cor, np.float32)
        n = min(x.size, in_abs.size, in_cor.size)
        out = np.zeros(max(noutput, 1), np.complex64)
        cons, nt = C.c_int(), C.c_int()
        to, tc = (C.c_uint64 * 8)(), (C.c_double * 8)()
        no = self.ctx.check(self.L.jrc_frame_detector_work(self.h, noutput, n, _ptr(x), _ptr(in_abs), _ptr(in_cor), _ptr(out),
                                                           C.byref(cons), to, tc, 8, C.byref(nt)))
        return out[:no], cons.value, [(int(to[i]), float(tc[i])) for i in range(nt.value)]

    def run(self, x, in_abs, in_cor, chunk=1 << 30):
        pos, outs, tags = 0, [], []
        while pos < len(x):
            n = min(chunk, len(x) - pos)
            o, c, t = self.work(x[pos:pos + n], in_abs[pos:pos + n], in_cor[pos:pos + n], n)
            outs.append(o)
            tags += t
            if c == 0 and o.size == 0:
                break
            pos += c
        return (np.concatenate(outs) if outs else np.zeros(0, np.complex64)), tags

    def close(self):
        if getattr(self, "h", None) and getattr(getattr(self, "ctx", None), "h", None):   # (Context.close() closes its children before it destroys itself; a context already gone means this object went with it)
            self.L.jrc_frame_detector_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class frame_sync:
    """include/mimo_ofdm_jrc/frame_sync.h make(fft_len, cp_len, sync_length, ltf_seq_time, debug); general_work =
    lib/frame_sync_impl.cc:89-229.  work() is one general_work call (at most 8192 items, :111)."""

    def __init__(self, fft_len, cp_len, sync_length, ltf_seq_time, debug=False, ctx=None):
        self.ctx = ctx or default_context()
        self.L = _load_sync()
        t = _c64(ltf_seq_time).ravel()
        self.h = self.L.jrc_frame_sync_create(self.ctx.h, fft_len, cp_len, int(sync_length), _ptr(t), t.size)
        self.ctx._adopt(self)
        if not self.h:
            raise ValueError(self.ctx.lib.jrc_last_error(self.ctx.h).decode())

    def _state(self):
        st, fs, fo = C.c_int(), C.c_int(), C.c_float()
        self.L.jrc_frame_sync_state(self.h, C.byref(st), C.byref(fs), C.byref(fo))
        return st.value, fs.value, fo.value

    state = property(lambda self: self._state()[0])
    frame_start = property(lambda self: self._state()[1])
    freq_offset = property(lambda self: self._state()[2])

    def work(self, x, x_delayed, tags, noutput):
        x, xd = _c64(x).ravel(), _c64(x_delayed).ravel()
        out = np.zeros(max(noutput, 1), np.complex64)
        nt = len(tags)
        to = (C.c_uint64 * max(nt, 1))(*[int(t[0]) for t in tags])
        tv = (C.c_double * max(nt, 1))(*[float(t[1]) for t in tags])
        cons, nto = C.c_int(), C.c_int()
        oo, ov = (C.c_uint64 * 1)(), (C.c_double * 1)()
        no = self.ctx.check(self.L.jrc_frame_sync_work(self.h, noutput, x.size, xd.size, _ptr(x) if x.size else None,
                                                       _ptr(xd) if xd.size else None, to, tv, nt, _ptr(out), C.byref(cons), oo, ov,
                                                       C.byref(nto)))
        return out[:no], cons.value, ([(int(oo[0]), float(ov[0]))] if nto.value else [])

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
        return (np.concatenate(outs) if outs else np.zeros(0, np.complex64)), otags

    def close(self):
        if getattr(self, "h", None) and getattr(getattr(self, "ctx", None), "h", None):   # (Context.close() closes its children before it destroys itself; a context already gone means this object went with it)
            self.L.jrc_frame_sync_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class zero_pad:
    """include/mimo_ofdm_jrc/zero_pad.h make(debug, pad_front, pad_tail); work = lib/zero_pad_impl.cc:62-94 (tagged stream block):
    the burst framed by pad_front / pad_tail samples of N(0, 1e-2) complex noise."""

    def __init__(self, debug=False, pad_front=0, pad_tail=0, seed=0, ctx=None):
        self.ctx = ctx or default_context()
        L = load()
        L.jrc_zero_pad.argtypes = [_vp, C.c_int, C.c_uint, C.c_uint, C.c_uint64, _vp, _vp]
        self.pad_front, self.pad_tail, self._seed = int(pad_front), int(pad_tail), int(seed)

    def calculate_output_stream_length(self, ninput_items):
        return ninput_items + self.pad_front + self.pad_tail

    def work(self, x):
        x = _c64(x).ravel()
        out = np.zeros(self.calculate_output_stream_length(x.size), np.complex64)
        n = self.ctx.check(self.ctx.lib.jrc_zero_pad(self.ctx.h, x.size, self.pad_front, self.pad_tail, self._seed,
                                                     _ptr(x) if x.size else None, _ptr(out)))
        self._seed += 1                                  # a new draw per packet, like the reference's per-call random_device
        assert n == out.size
        return out


class SyncCfg(C.Structure):
    _fields_ = [("fft_len", C.c_int), ("cp_len", C.c_int), ("threshold", C.c_double), ("min_n_peaks", C.c_uint), ("ignore_gap", C.c_uint),
                ("sync_length", C.c_int), ("n_taps", C.c_int), ("d_ltf_taps", _vp), ("delay", C.c_int), ("window", C.c_int),
                ("power_window", C.c_int), ("power_scale", C.c_float)]


class SyncFrame(C.Structure):
    _fields_ = [("start", C.c_int), ("len", C.c_int), ("coarse_cfo", C.c_float), ("frame_start", C.c_int), ("fine_cfo", C.c_float),
                ("tag_value", C.c_double), ("n_out", C.c_int), ("pad_", C.c_int)]


class SyncFrontEnd:
    """Device-resident sync front end (jrc_sync_frontend_dev): detection metrics -> frame_detector -> frame_sync run to completion
    on a capture in HBM; frame k lands in row k of `frames` ([max_frames, max_symbols, fft_len] complex, torch tensor)."""

    def __init__(self, fft_len, cp_len, threshold, min_n_peaks, ignore_gap, sync_length, ltf_seq_time, max_frames=64, max_symbols=64,
                 ctx=None):
        import torch
        self.ctx = ctx or default_context()
        L = _load_sync()
        L.jrc_sync_frontend_work_bytes.restype = C.c_size_t
        L.jrc_sync_frontend_work_bytes.argtypes = [C.c_int]
        L.jrc_sync_frontend_dev.argtypes = [_vp, C.POINTER(SyncCfg), C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
        self.L = L
        t = _c64(ltf_seq_time).ravel()
        self.d_taps = torch.from_numpy(t.view(np.float32).copy()).cuda()
        self.cfg = SyncCfg(fft_len, cp_len, float(threshold), int(min_n_peaks), int(ignore_gap), int(sync_length), t.size,
                           _vp(self.d_taps.data_ptr()), fft_len // 4, fft_len // 2, int(1.5 * (fft_len // 2)), 1 / 1.5)
        self.max_frames, self.max_symbols, self.fft_len = max_frames, max_symbols, fft_len
        self.frames = torch.zeros((max_frames, max_symbols, fft_len, 2), dtype=torch.float32, device="cuda")
        self.d_info = torch.zeros((max_frames, C.sizeof(SyncFrame)), dtype=torch.uint8, device="cuda")
        self.d_n = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()     # the fills above ran on torch's stream; the library launches on its own
        self._work = None

    def run(self, d_x, n_samples, stream=None):
        """d_x: torch float32 [n_samples, 2] on the device; asynchronous"""
        import torch
        need = self.L.jrc_sync_frontend_work_bytes(n_samples)
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.uint8, device="cuda")
        self.ctx.check(self.L.jrc_sync_frontend_dev(self.ctx.h, C.byref(self.cfg), n_samples, _vp(d_x.data_ptr()), _vp(self._work.data_ptr()),
                                                    self.max_frames, self.max_symbols, _vp(self.frames.data_ptr()), _vp(self.d_info.data_ptr()),
                                                    _vp(self.d_n.data_ptr()), stream))

    def results(self):
        self.ctx.sync()
        n = int(self.d_n.cpu().item())
        raw = self.d_info[:n].cpu().numpy().tobytes()
        info = [SyncFrame.from_buffer_copy(raw[i * C.sizeof(SyncFrame):(i + 1) * C.sizeof(SyncFrame)]) for i in range(n)]
        return n, info


class ofdm_frame_generator:
    """include/mimo_ofdm_jrc/ofdm_frame_generator.h make(fft_len, occupied_carriers, pilot_carriers, pilot_symbols, sync_words, ltf_len,
    len_tag_key, output_is_shifted); work = lib/ofdm_frame_generator_impl.cc:155-216 (tagged stream block)."""

    def __init__(self, fft_len, occupied_carriers, pilot_carriers, pilot_symbols, sync_words, ltf_len=0, len_tag_key="packet_len",
                 output_is_shifted=True, ctx=None):
        self.ctx = ctx or default_context()
        L = load()
        ip = C.POINTER(C.c_int)
        L.jrc_frame_generator_create.restype = _vp
        L.jrc_frame_generator_create.argtypes = [_vp, C.c_int, C.c_int, ip, ip, C.c_int, ip, ip, C.c_int, ip, _vp, C.c_int, _vp, C.c_int]
        L.jrc_frame_generator_destroy.argtypes = [_vp]
        L.jrc_frame_generator_destroy.restype = None
        L.jrc_frame_generator_output_length.argtypes = [_vp, C.c_int]
        L.jrc_frame_generator_work.argtypes = [_vp, C.c_int, _vp, _vp]
        L.jrc_frame_generator_dev.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp]
        self.L, self.fft_len = L, fft_len

        def flat(sets, dtype):
            sizes = np.array([len(x) for x in sets], np.int32)
            vals = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype).ravel() for x in sets]) if sizes.sum() else np.zeros(1, dtype), dtype)
            return sizes, vals
        if len(occupied_carriers) == 0:
            raise ValueError("Occupied carriers must be of type vector of vector i.e. ((),).")
        if len(pilot_carriers) == 0:
            raise ValueError("Pilot carriers must be of type vector of vector i.e. ((),).")
        if len(pilot_symbols) == 0:
            raise ValueError("Pilot symbols must be of type vector of vector i.e. ((),).")
        osz, ofl = flat(occupied_carriers, np.int32)
        psz, pfl = flat(pilot_carriers, np.int32)
        ssz, sfl = flat(pilot_symbols, np.complex64)
        sw = _c64(np.asarray(sync_words, np.complex64)) if len(sync_words) else np.zeros((0, fft_len), np.complex64)
        if sw.size and sw.shape[-1] != fft_len:
            raise ValueError("sync words must be fft length")
        swp = sw if sw.size else np.zeros(1, np.complex64)
        self.h = L.jrc_frame_generator_create(self.ctx.h, fft_len, len(osz), osz.ctypes.data_as(ip), ofl.ctypes.data_as(ip), len(psz),
                                              psz.ctypes.data_as(ip), pfl.ctypes.data_as(ip), len(ssz), ssz.ctypes.data_as(ip), _ptr(sfl),
                                              sw.shape[0] if sw.size else 0, _ptr(swp), int(bool(output_is_shifted)))
        if not self.h:
            raise ValueError(self.ctx.lib.jrc_last_error(self.ctx.h).decode())
        self.ctx._adopt(self)

    def calculate_output_stream_length(self, ninput_items):
        return self.L.jrc_frame_generator_output_length(self.h, ninput_items)

    def work(self, x):
        x = _c64(x).ravel()
        out = np.zeros((self.calculate_output_stream_length(x.size), self.fft_len), np.complex64)
        n = self.ctx.check(self.L.jrc_frame_generator_work(self.h, x.size, _ptr(x) if x.size else None, _ptr(out)))
        assert n == out.shape[0]
        return out

    def close(self):
        if getattr(self, "h", None) and getattr(getattr(self, "ctx", None), "h", None):   # (Context.close() closes its children before it destroys itself; a context already gone means this object went with it)
            self.L.jrc_frame_generator_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
