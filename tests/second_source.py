"""Second-source restatements of the comm-side reference blocks (TEST INFRASTRUCTURE, CPU tier).

Written directly from the reference's C++ (`/root/reference/lib/*.cc`, cited per function), NOT from `oracle/*.c`: the reference
cannot be compiled in this image (GNU Radio / Eigen / Boost headers absent), so the C oracle is "parity unpinned"; these
independent numpy-float32 / pure-Python restatements are a second reading of the same source that the oracle must agree with
(`tests/test_second_source.py`).  Two readings that agree bit for bit are still not an execution of the reference - parity
stays "partial" (README, docs/history.md §5) - but a misreading now has to be made twice, the same way, in two languages.

Arithmetic model (what a g++ build of the reference does on x86-64, the image's toolchain: gcc 11.4 / glibc 2.35):
  * gr_complex = std::complex<float>; `*` is the inline form of libgcc __mulsc3 (four products, one subtraction, one addition, each
    rounded to float, no FMA on baseline x86-64); `/` is __divsc3 out of libgcc_s.so.1 (12.3 here: quotient formed in double, rounded
    once); complex / float divides each part.
  * std::exp(gr_complex(0, x)) is glibc cexpf: (cosf(x), sinf(x)) for |x| > FLT_MIN; std::arg is atan2f; std::abs is hypotf.
    The libm of this image is called through ctypes for those (numpy's own float32 sin/cos are not glibc's).
  * expressions mixing double and float follow the C++ promotions written in the source.
Every float operation below is one numpy float32 ufunc call or one np.float32 scalar operation, so it is rounded exactly once."""
import ctypes
import ctypes.util

import numpy as np

f32 = np.float32
c64 = np.complex64

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n, _na in (("sinf", 1), ("cosf", 1), ("atan2f", 2), ("hypotf", 2)):
    _f = getattr(_libm, _n)
    _f.restype = ctypes.c_float
    _f.argtypes = [ctypes.c_float] * _na
_libm.log10.restype = ctypes.c_double
_libm.log10.argtypes = [ctypes.c_double]

NDP, DATA = 1, 2                        # enum PACKET_TYPE (include/mimo_ofdm_jrc/stream_encoder.h)
LS, STA = 0, 1                          # enum ChannelEstimator (include/mimo_ofdm_jrc/mimo_ofdm_equalizer.h)
FLT_MIN = 1.17549435e-38


# ------------------------------------------------------------------------------------------------ complex<float> arithmetic
def cplx(re, im):
    out = np.empty(np.broadcast(re, im).shape, c64)
    out.real = re
    out.imag = im
    return out


def parts(z):
    z = np.asarray(z, c64)
    return z.real.astype(f32), z.imag.astype(f32)


def cmul(x, y):
    """__mulsc3 for finite operands: (ac - bd, ad + bc), every product and sum rounded to float"""
    a, b = parts(x)
    c, d = parts(y)
    return cplx(a * c - b * d, a * d + b * c)


def cdiv(x, y):
    """__divsc3 as a g++ build of the reference executes it on this image: g++ links -lgcc_s ahead of -lgcc, and libgcc_s.so.1 here is
    GCC 12.3, whose float version forms the textbook quotient in double and rounds once (gcc's static libgcc.a 11.4 would use Smith's
    method in float).  tests/test_second_source.py checks this bit for bit against the __divsc3 the box's libgcc_s exports."""
    a, b = parts(x)
    c, d = parts(y)
    aa, bb, cc, dd = (np.asarray(v, np.float64) for v in (a, b, c, d))
    with np.errstate(all="ignore"):
        den = cc * cc + dd * dd
        return cplx(((aa * cc + bb * dd) / den).astype(f32), ((bb * cc - aa * dd) / den).astype(f32))


def cdiv_real(x, r):
    """std::complex<float> / float: each part divided"""
    a, b = parts(x)
    return cplx(a / f32(r), b / f32(r))


def cconj(x):
    a, b = parts(x)
    return cplx(a, -b)


def cadd(x, y):
    a, b = parts(x)
    c, d = parts(y)
    return cplx(a + c, b + d)


def csub(x, y):
    a, b = parts(x)
    c, d = parts(y)
    return cplx(a - c, b - d)


def cexp_j(x):
    """std::exp(gr_complex(0, x)) -> glibc cexpf: expf(0) * (cosf(x), sinf(x)); the double argument is narrowed by the constructor"""
    xf = float(f32(x))
    if abs(xf) > FLT_MIN:
        return c64(complex(_libm.cosf(xf), _libm.sinf(xf)))
    return c64(complex(1.0, xf))


def carg(z):
    """std::arg(std::complex<float>) = atan2f(imag, real)"""
    return float(_libm.atan2f(float(np.imag(z)), float(np.real(z))))


def cabs(z):
    """std::abs(std::complex<float>) = cabsf = hypotf"""
    return float(_libm.hypotf(float(np.real(z)), float(np.imag(z))))


def csum_in_order(v):
    """`acc += v[k]` over k in index order, float parts"""
    re, im = f32(0), f32(0)
    for z in np.asarray(v, c64).ravel():
        re = f32(re + f32(z.real))
        im = f32(im + f32(z.imag))
    return c64(complex(re, im))


# ------------------------------------------------------------------------------------------------ lib/utils.cc
RATE_FIELD = {0: 0x0D, 1: 0x0F, 2: 0x05, 3: 0x07, 4: 0x09, 5: 0x0B}         # ofdm_mcs::ofdm_mcs (:47-101)
RATE_BITMAP_TO_MCS = {11: 0, 15: 1, 10: 2, 14: 3, 9: 4, 13: 5}                # decode_signal_field switch (:728-774)


def mcs_params(mcs, n_dc):
    """ofdm_mcs (lib/utils.cc:47-101): n_bpsc, n_cbps, n_dbps"""
    n_bpsc = (1, 1, 2, 2, 4, 4)[mcs]
    n_cbps = n_dc * n_bpsc
    n_dbps = n_cbps // 2 if mcs in (0, 2, 4) else n_cbps * 3 // 4
    return n_bpsc, n_cbps, n_dbps


def packet_n_sym(mcs, n_dc, data_size_byte):
    """packet_param (lib/utils.cc:26-36): n_ofdm_sym = ceil((16 + 8 len + 6) / n_dbps)"""
    return int(np.ceil((16 + 8 * data_size_byte + 6) / float(mcs_params(mcs, n_dc)[2])))


def conv_encode(bits):
    """convolutional_encoding (lib/utils.cc:204-214)"""
    state, out = 0, []
    for b in bits:
        state = ((state << 1) & 0x7e) | int(b)
        out += [bin(state & 0o155).count("1") % 2, bin(state & 0o117).count("1") % 2]
    return np.array(out, np.uint8)


def signal_field(n_dc, mcs, packet_type, length):
    """mimo_precoder_impl::generate_signal_field (lib/mimo_precoder_impl.cc:983-1060): 24 header bits, rate-1/2 code over
    n_data_bits = n_ofdm_sym(BPSK_1_2, 0 bytes) * n_dc/2 bits of a zeroed buffer, one bit per data carrier, BPSK points -1 / +1"""
    hdr = np.zeros(max(n_dc // 2, 24), np.uint8)
    rf = RATE_FIELD[mcs]
    hdr[0:4] = [(rf >> 3) & 1, (rf >> 2) & 1, (rf >> 1) & 1, rf & 1]
    hdr[4] = 1 if packet_type == DATA else 0
    for i in range(12):
        hdr[5 + i] = (length >> i) & 1
    hdr[17] = int(hdr[:17].sum()) % 2
    n_dbps = n_dc // 2
    n_sym = int(np.ceil(22 / float(n_dbps)))
    enc = conv_encode(hdr[:n_sym * n_dbps])
    return np.where(enc[:n_dc] > 0, 1.0, -1.0).astype(f32)        # constellation_bpsk: 0 -> -1, 1 -> +1


# ------------------------------------------------------------------------------------------------ lib/viterbi_decoder.cc
class ViterbiWindowed:
    """viterbi_decoder (lib/viterbi_decoder.cc:99-331) with the sixteen byte lanes of each __m128i written out: 64 unsigned-char
    metrics with wrap-around adds, signed-byte compare of the difference, 8-bit path registers, a traceback window of
    ntraceback chunks.  Symbols the decoder reads beyond the coded bits handed in are 0 (the reference reads whatever lies
    behind its buffer there: `rx_bits` is calloc(n_data_carriers), lib/mimo_ofdm_equalizer_impl.cc:175)."""

    def __init__(self):
        polys = (0x6d, 0x4f)                                                   # viterbi_chunks_init_sse2 (:318-338)
        self.bt = np.array([[bin((2 * i) & p).count("1") & 1 for i in range(32)] for p in polys], np.uint8)
        self.store_pos = 0

    def _half_butterfly(self, metric, path, s0, s1):
        """one of the two halves of viterbi_butterfly2_sse2 (:87-180): lane k < 32 pairs old states k and k + 32"""
        one, two = np.uint8(1), np.uint8(2)
        if s0 == 2:
            metsvm = self.bt[1] ^ np.uint8(s1)
            metsv = one - metsvm
        elif s1 == 2:
            metsvm = self.bt[0] ^ np.uint8(s0)
            metsv = one - metsvm
        else:
            metsvm = (self.bt[0] ^ np.uint8(s0)) + (self.bt[1] ^ np.uint8(s1))
            metsv = two - metsvm
        lo, hi = metric[:32], metric[32:]
        m0, m1, m2, m3 = lo + metsv, hi + metsvm, lo + metsvm, hi + metsv      # _mm_add_epi8 wraps
        d0 = (m0 - m1).view(np.int8) > 0                                       # _mm_cmpgt_epi8(_mm_sub_epi8(m0, m1), 0)
        d1 = (m2 - m3).view(np.int8) > 0
        sv0, sv1 = np.where(d0, m0, m1), np.where(d1, m2, m3)
        p16 = path.view(np.uint16)                                             # _mm_slli_epi16: bit 7 of a low byte spills upward
        sh = (p16 << np.uint16(1)).view(np.uint8)
        shift0, shift1 = sh[:32], sh[32:] + one
        t0, t1 = np.where(d0, shift0, shift1), np.where(d1, shift0, shift1)
        new_m, new_p = np.empty(64, np.uint8), np.empty(64, np.uint8)
        new_m[0::2], new_m[1::2] = sv0, sv1                                    # unpacklo / unpackhi: new state 2k, 2k + 1
        new_p[0::2], new_p[1::2] = t0, t1
        return new_m, new_p

    def _get_output(self, ntraceback):
        """viterbi_get_output_sse2 (:183-225)"""
        self.store_pos = (self.store_pos + 1) % ntraceback
        mm = self.metric.copy()
        self.pp[self.store_pos] = self.path
        best, bestmetric, minmetric = 0, int(mm[0]), int(mm[0])
        for i in range(1, 64):
            if int(mm[i]) > bestmetric:
                bestmetric, best = int(mm[i]), i
            if int(mm[i]) < minmetric:
                minmetric = int(mm[i])
        pos = self.store_pos
        for _ in range(ntraceback - 1):
            best = int(self.pp[pos][best]) >> 2
            pos = (pos - 1 + ntraceback) % ntraceback
        out = int(self.pp[pos][best])
        self.path = np.zeros(64, np.uint8)
        self.metric = self.metric - np.uint8(minmetric)
        return out

    def decode(self, mcs, n_ofdm_sym, n_cbps, n_data_bits, bits):
        """viterbi_decoder::decode (:258-291) after reset (:293-316) and depuncture (:228-256)"""
        with np.errstate(over="ignore"):
            return self._decode(mcs, n_ofdm_sym, n_cbps, n_data_bits, bits)

    def _decode(self, mcs, n_ofdm_sym, n_cbps, n_data_bits, bits):
        self.metric, self.path = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
        self.pp = np.zeros((24, 64), np.uint8)
        half = mcs in (0, 2, 4)
        ntraceback = 5 if half else 10
        bits = np.asarray(bits, np.uint8)
        if half:
            dep = list(bits[:n_ofdm_sym * n_cbps])
        else:
            pat, k2 = (1, 1, 1, 0, 0, 1), 6
            dep = []
            for i in range(n_ofdm_sym):
                for k in range(n_cbps):
                    while pat[len(dep) % k2] == 0:
                        dep.append(2)
                    dep.append(int(bits[i * n_cbps + k]))
                    while pat[len(dep) % k2] == 0:
                        dep.append(2)

        def sym(i):
            return int(dep[i]) if i < len(dep) else 0

        decoded, in_count, out_count = [], 0, 0
        while len(decoded) < n_data_bits:
            if in_count % 4 == 0:
                b = in_count & 0xfffffffc
                self.metric, self.path = self._half_butterfly(self.metric, self.path, sym(b), sym(b + 1))
                self.metric, self.path = self._half_butterfly(self.metric, self.path, sym(b + 2), sym(b + 3))
                if in_count > 0 and in_count % 16 == 8:
                    c = self._get_output(ntraceback)
                    if out_count >= ntraceback:
                        decoded += [(c >> (7 - i)) & 1 for i in range(8)]
                    out_count += 1
            in_count += 1
        return np.array(decoded, np.uint8)


def parse_signal(bits, n_dc):
    """decode_signal_field after the decoder (lib/mimo_ofdm_equalizer_impl.cc:669-781): (ok, mcs, packet_type, length, n_ofdm_sym)"""
    rate_bitmap = pt_bitmap = length = 0
    parity = 0
    for i in range(17):
        parity ^= int(bits[i])
        if i < 4 and bits[i]:
            rate_bitmap |= 1 << i
        if i == 4 and bits[i]:
            pt_bitmap |= 1 << (i - 4)
        if bits[i] and 4 < i < 17:
            length |= 1 << (i - 5)
    trailing_zeros_correct = all(bits[i] == 0 for i in range(17, 23))
    if parity != int(bits[17]) and trailing_zeros_correct:                    # sic (:702)
        return False, None, None, 0, 0
    if pt_bitmap not in (0, 1):
        return False, None, None, length, 0
    ptype = NDP if pt_bitmap == 0 else DATA
    if rate_bitmap not in RATE_BITMAP_TO_MCS:
        return False, None, ptype, length, 0
    mcs = RATE_BITMAP_TO_MCS[rate_bitmap]
    return True, mcs, ptype, length, packet_n_sym(mcs, n_dc, length)


# ------------------------------------------------------------------------------------------------ gr-digital constellations (3.8)
QPSK_A = f32(0.707107)                   # constellation_qpsk points (+-a, +-a), index 2 (im > 0) + (re > 0)


def decide_and_map(bpsc, z):
    """modulator_SIG->decision_maker + map_to_points, then the QPSK halving of the caller
    (lib/mimo_ofdm_equalizer_impl.cc:505-514, :563-570)"""
    re, im = f32(np.real(z)), f32(np.imag(z))
    if bpsc == 1:
        return c64(complex(1.0 if re > 0 else -1.0, 0.0))
    if bpsc == 2:
        p = c64(complex(QPSK_A if re > 0 else -QPSK_A, QPSK_A if im > 0 else -QPSK_A))
        return cdiv_real(p, 2.0)[()]
    # 16-QAM: the gr-digital 3.8 table is not in the reference tree (parity unpinned there); nearest level per axis, level = sqrt(0.1)
    lvl = f32(np.sqrt(f32(0.1)))

    def axis(v):
        return (f32(1) if abs(v) < f32(2) * lvl else f32(3)) * lvl * (f32(1) if v > 0 else f32(-1))
    return c64(complex(axis(re), axis(im)))


# ------------------------------------------------------------------------------------------------ lib/mimo_ofdm_equalizer_impl.cc
class EqualizerRef:
    """mimo_ofdm_equalizer_impl::general_work (lib/mimo_ofdm_equalizer_impl.cc:199-648) for one input stream"""

    def __init__(self, estimator_algo, freq, bw, fft_len, cp_len, data_carriers, pilot_carriers, pilot_symbols, ltf_seq,
                 mapped_ltf_symbols, n_mimo_ltf):
        self.algo, self.freq, self.bw, self.N, self.cp = estimator_algo, float(freq), float(bw), fft_len, cp_len
        self.pilot_c = [int(c) + fft_len // 2 for c in pilot_carriers]                       # :132-140
        self.data_c = [int(c) + fft_len // 2 for c in data_carriers]
        self.active = sorted(self.data_c + self.pilot_c)                                      # :153-159
        self.pilot_symbols = np.asarray(pilot_symbols, c64)
        self.ltf = np.asarray(ltf_seq, c64)
        self.mapped = np.asarray(mapped_ltf_symbols, c64)                                     # [N][T * N_ltf]
        self.NL = n_mimo_ltf
        self.T = self.mapped.shape[1] // n_mimo_ltf                                           # :168
        self.H = np.zeros(fft_len, c64)
        self.H_mimo = np.zeros(fft_len, c64)
        self.pre = np.zeros((fft_len, n_mimo_ltf), c64)     # the reference's stack VLA (:213); kept across calls here as in the build
        self.symbol_ind = 0
        self.sig_ok = False                                  # members are uninitialised in the reference until the first frame_start
        self.n_sym_sig = 0
        self.total_out = 0
        self.equalize_done = True
        self.signal_power_sum = self.noise_power_sum = 0.0
        self.snr_est_count = 0
        self.snr_est = self.freq_offset = self.epsilon0 = self.er = 0.0
        self.precoded_snr = 0.0
        self.chan_mean = np.zeros(0, c64)
        self.mcs = self.ptype = self.length = None
        self.dec = ViterbiWindowed()

    # estimate_residual_cfo (:908-922)
    def _residual_cfo(self, Y, chan, ref):
        est = cmul(chan[self.pilot_c], ref)
        prod = cmul(Y[self.pilot_c], cconj(est))
        return carg(csum_in_order(prod)), est

    def general_work(self, symbols, frame_start_tags=(), noutput_items=None):
        x = np.asarray(symbols, c64).reshape(-1, self.N)
        nin = x.shape[0]
        nout = nin if noutput_items is None else noutput_items
        tags = {int(o): float(v) for o, v in frame_start_tags}
        N, cp, NL = self.N, self.cp, self.NL
        out, events, chan_est = [], [], None
        n_in = 0
        while n_in < nin and len(out) < nout:                                                 # :219
            if n_in in tags:                                                                  # :221-245
                self.symbol_ind, self.total_out, self.n_sym_sig = 0, 0, 0
                phi = tags[n_in]
                self.freq_offset = phi * self.bw / (2 * np.pi)
                self.epsilon0 = phi * self.bw / (2 * np.pi * self.freq)
                self.er = 0.0
                self.sig_ok, self.equalize_done = True, False
                self.signal_power_sum = self.noise_power_sum = 0.0
                self.snr_est_count = 0
            if self.symbol_ind > self.n_sym_sig + 2 + NL or not self.sig_ok:                  # :250-255
                n_in += 1
                continue
            # sampling-offset derotation (:261-264): the whole argument in double, narrowed by gr_complex(0, .)
            rot = np.array([cexp_j(2 * np.pi * self.symbol_ind * ((N + cp) * 1.0 / N) * (self.epsilon0 + self.er) * (i - N // 2))
                            for i in range(N)], c64)
            Y = cmul(x[n_in], rot)
            ind = self.symbol_ind
            if ind == 0:                                                                      # :272-275
                self.H = Y.copy()
            elif ind == 1:                                                                    # :277-306
                signal = noise = 0.0
                for c in self.active:
                    noise += cabs(csub(self.H[c], Y[c])[()]) ** 2.0                           # std::pow(float, int) -> double
                    signal += cabs(cadd(self.H[c], Y[c])[()]) ** 2.0
                    h = cadd(self.H[c], Y[c])
                    self.H[c] = cdiv(h, cmul(self.ltf[c], c64(2.0)))[()]
                # (:288-303: a common phase error is estimated and applied to rx_symbol_Y, which is not used again)
                with np.errstate(all="ignore"):
                    self.snr_est = 10 * _libm.log10(float(np.float64(signal) / np.float64(noise) / 2))      # noise-free input: +inf, as in C
            elif ind == 2:                                                                    # :308-344
                cfo, _ = self._residual_cfo(Y, self.H, self.pilot_symbols[0])
                Y = cmul(Y, cexp_j(-cfo))
                Z = cdiv(Y[self.data_c], self.H[self.data_c])                                 # symbol_equalize (:900-906)
                rx_bits = (Z.real > 0).astype(np.uint8)                                       # constellation_bpsk::decision_maker
                nd = len(self.data_c)
                n_dbps = nd // 2                                                              # ofdm_mcs(BPSK_1_2, nd); packet_param(., 0, NDP)
                n_sym = int(np.ceil(22 / float(n_dbps)))
                dec = self.dec.decode(0, n_sym, nd, n_sym * n_dbps, rx_bits)
                self.sig_ok, self.mcs, self.ptype, self.length, self.n_sym_sig = parse_signal(dec, nd)
                if self.sig_ok:
                    events.append(dict(kind=1, offset=len(out), data_bytes=self.length, mcs=self.mcs, packet_type=self.ptype,
                                       snr=self.snr_est, freq_offset=self.freq_offset))
            elif 3 <= ind <= 2 + NL:                                                          # :346-463
                l = ind - 3
                self.pre[:, l] = Y
                if l == NL - 1:
                    if self.ptype == NDP:                                                     # :375-422: H = conj(X_ltf) y per subcarrier
                        chan_est = np.zeros((N, self.T), c64)
                        mean = np.zeros(self.T, c64)
                        for sc in range(N):
                            for t in range(self.T):
                                terms = cmul(cconj(self.mapped[sc, t * NL:(t + 1) * NL]), self.pre[sc])
                                chan_est[sc, t] = csum_in_order(terms)                        # Eigen's order: see docs/history.md §5 (unpinned)
                            if sc in self.active:
                                mean = cadd(mean, chan_est[sc])
                        self.chan_mean = cdiv_real(mean, len(self.active))                    # VectorXcf / int -> Literal = float
                    elif self.ptype == DATA:                                                  # :423-456: row(0).dot(y) / N_ltf
                        acc = c64(0)
                        for c in self.data_c + self.pilot_c:
                            terms = cmul(cconj(self.mapped[c, 0:NL]), self.pre[c])
                            self.H_mimo[c] = cdiv_real(csum_in_order(terms), NL)[()]
                            acc = cadd(acc, self.H_mimo[c])[()]
                        self.chan_mean = np.array([cdiv_real(acc, len(self.active))[()]], c64)
            else:                                                                             # data symbols :465-605
                ref = self.pilot_symbols[(ind - 3 - NL) % len(self.pilot_symbols)]
                Hsel = self.H if self.ptype == NDP else self.H_mimo
                cfo, est = self._residual_cfo(Y, Hsel, ref)
                Y = cmul(Y, cexp_j(-cfo))
                for k, p in enumerate(self.pilot_c):                                          # :484-493
                    self.signal_power_sum += float(np.real(cmul(est[k], cconj(est[k]))))
                    err = csub(est[k], Y[p])
                    self.noise_power_sum += float(np.real(cmul(err, cconj(err))))
                    self.snr_est_count += 1
                bpsc = mcs_params(self.mcs, len(self.data_c))[0]
                if self.ptype == NDP:
                    Z = cdiv(Y[self.data_c], self.H[self.data_c])
                    if self.algo == STA:                                                      # :499-535
                        alpha = f32(0.5)
                        for i, c in enumerate(self.data_c):
                            X = decide_and_map(bpsc, Z[i])
                            upd = cdiv(Y[c], X)
                            self.H[c] = cadd(cmul(c64(complex(f32(1) - alpha, 0)), self.H[c]), cmul(c64(complex(alpha, 0)), upd))[()]
                        for k, c in enumerate(self.pilot_c):
                            upd = cdiv(cmul(c64(complex(alpha, 0)), Y[c]), ref[k])
                            self.H[c] = cadd(cmul(c64(complex(f32(1) - alpha, 0)), self.H[c]), upd)[()]
                else:                                                                         # DATA :536-592
                    Z = np.zeros(len(self.data_c), c64)
                    for i, c in enumerate(self.data_c):
                        hh = f32(np.real(cmul(self.H_mimo[c], cconj(self.H_mimo[c]))))
                        csi = f32(float(hh) + self.noise_power_sum / self.snr_est_count)      # float + double -> double -> float csi_est
                        Z[i] = cdiv_real(cmul(Y[c], cconj(self.H_mimo[c])), csi)[()]
                    if self.algo == STA:
                        alpha = f32(0.4)
                        for i, c in enumerate(self.data_c):
                            X = decide_and_map(bpsc, Z[i])
                            upd = cdiv(cmul(c64(complex(alpha, 0)), Y[c]), X)
                            self.H_mimo[c] = cadd(cmul(c64(complex(f32(1) - alpha, 0)), self.H_mimo[c]), upd)[()]
                        for k, c in enumerate(self.pilot_c):
                            upd = cdiv(cmul(c64(complex(alpha, 0)), Y[c]), ref[k])
                            self.H_mimo[c] = cadd(cmul(c64(complex(f32(1) - alpha, 0)), self.H_mimo[c]), upd)[()]
                out.append(np.asarray(Z, c64).copy())                                         # :602-604
            n_in += 1
            self.symbol_ind += 1
        n_out = len(out)
        self.total_out += n_out                                                               # :609
        if self.total_out == self.n_sym_sig and self.sig_ok and not self.equalize_done:       # :616-632
            if self.snr_est_count != 0:
                with np.errstate(all="ignore"):
                    self.precoded_snr = 10 * _libm.log10(float(np.float64(self.signal_power_sum / self.snr_est_count) / np.float64(self.noise_power_sum / self.snr_est_count)))
            events.append(dict(kind=2, offset=n_out - 1, snr_data=self.precoded_snr, chan_mean=np.asarray(self.chan_mean, c64).copy()))
            self.equalize_done = True
        return dict(out=np.array(out, c64).reshape(n_out, len(self.data_c)), consumed=n_in, events=events, chan_est=chan_est)


# ------------------------------------------------------------------------------------------------ lib/mimo_precoder_impl.cc
def dft_matrix(T):
    """get_dft_matrix_eigen (lib/mimo_precoder_impl.cc:760-772): exp(gr_complex(0, -2 pi float(r c) / float(N))) / (gr_complex) sqrt(N)"""
    M = np.zeros((T, T), c64)
    root = c64(complex(f32(np.sqrt(float(T))), 0))                 # std::sqrt(int) -> double, narrowed by the cast
    for r in range(T):
        for c in range(T):
            arg = -2 * np.pi * float(f32(r * c)) / float(f32(T))   # GR_M_PI is a double literal: the quotient is formed in double
            M[r, c] = cdiv(cexp_j(arg), root)[()]
    return M


class PrecoderRef:
    """mimo_precoder_impl::work (lib/mimo_precoder_impl.cc:277-741), deterministic sub-paths: the steering matrix and the radar
    stream symbols are arguments (the reference reads / draws them: files :774-981, std::random_device :435-437)"""

    def __init__(self, fft_len, N_tx, data_carriers, pilot_carriers, pilot_symbols, sync_words, mapped_ltf_symbols):
        self.N, self.T = fft_len, N_tx
        self.NL = N_tx                                                                        # :117
        wrap = lambda c: ((c + fft_len if c < 0 else c) + fft_len // 2) % fft_len             # :126-152
        self.data_c = [wrap(int(c)) for c in data_carriers]
        self.pilot_c = [wrap(int(c)) for c in pilot_carriers]
        self.pilot_symbols = np.asarray(pilot_symbols, c64)
        self.sync = np.asarray(sync_words, c64)
        self.mapped = np.asarray(mapped_ltf_symbols, c64)
        self.F = dft_matrix(N_tx)

    def _matvec(self, W, s):
        """(T x J) times (J) in index order of j (Eigen's summation order: docs/history.md §5, unpinned)"""
        out = np.zeros(W.shape[0], c64)
        for t in range(W.shape[0]):
            out[t] = csum_in_order(cmul(W[t], s))
        return out

    def work(self, symbols, mcs, packet_type, pdu_len, steer_mode=0, Q_mean=None, Q_sc=None, radar_streams=None):
        N, T, NL = self.N, self.T, self.NL
        nd = len(self.data_c)
        x = np.asarray(symbols, c64).ravel()
        n_sym = x.size // nd                                                                  # :290
        if packet_n_sym(mcs, nd, pdu_len) != n_sym:                                           # :327-333
            raise RuntimeError("[MIMO PRECODER] something is wrong!!")
        n_sync = len(self.sync)
        out = np.zeros((T, n_sym + n_sync + T + 1, N), c64)                                   # :337
        for t in range(min(T, 2)):                                                            # :340-347
            out[t, :n_sync] = self.sync
        sig = signal_field(nd, mcs, packet_type, pdu_len)                                     # :353-371
        for t in range(min(T, 2)):
            out[t, n_sync, self.data_c] = sig
            out[t, n_sync, self.pilot_c] = self.pilot_symbols[0]
        k0 = n_sync + 1
        if packet_type == NDP:                                                                # :375-428
            for t in range(T):
                for l in range(NL):
                    out[t, k0 + l] = self.mapped[:, l + t * NL]
            for m in range(n_sym):
                for t in range(min(T, 2)):
                    out[t, k0 + NL + m, self.data_c] = x[m * nd:(m + 1) * nd]
                    out[t, k0 + NL + m, self.pilot_c] = self.pilot_symbols[m % len(self.pilot_symbols)]
            return out
        # DATA (:430-712)
        use_streams = radar_streams is not None
        S = np.zeros((T if use_streams else 1, n_sym, N), c64)                                # stream_symbols (:452-493)
        for m in range(n_sym):
            S[0, m, self.data_c] = x[m * nd:(m + 1) * nd]
            S[0, m, self.pilot_c] = self.pilot_symbols[m % len(self.pilot_symbols)]
        if use_streams:
            rs = np.asarray(radar_streams, c64)
            act = self.data_c + self.pilot_c
            S[1:, :, act] = rs[:, :, act]
        W_all = self.F if steer_mode == 0 else (np.asarray(Q_mean, c64) if steer_mode == 1 else None)
        for sc in range(N):                                                                   # MIMO preamble (:540-576)
            X = self.mapped[sc].reshape(T, NL)
            if not X.any():
                continue
            W = W_all if W_all is not None else np.asarray(Q_sc, c64)[sc]
            for l in range(NL):
                out[:, k0 + l, sc] = self._matvec(W, X[:, l])
        for m in range(n_sym):                                                                # data + pilots (:606-705)
            for sc in self.data_c + self.pilot_c:
                W = W_all if W_all is not None else np.asarray(Q_sc, c64)[sc]
                if use_streams:
                    out[:, k0 + NL + m, sc] = self._matvec(W, S[:, m, sc])
                else:
                    out[:, k0 + NL + m, sc] = cmul(W[:, 0], S[0, m, sc])                      # Q.col(0) * scalar
        return out


# ------------------------------------------------------------------------------------------------ lib/range_angle_estimator_impl.cc, lib/fft_peak_detect_impl.cc
_libm.log10f.restype = ctypes.c_float
_libm.log10f.argtypes = [ctypes.c_float]


def _pow2_abs(z):
    """std::pow(std::abs(z), 2) on a std::complex<float>: hypotf in float, the square in double (pow(float, int) promotes)"""
    h = float(_libm.hypotf(float(np.real(z)), float(np.imag(z))))
    return h * h


def ra_estimate_ref(m, range_bins, angle_bins, noise_discard_range_m, noise_discard_angle_deg, snr_threshold=0.0, power_threshold=0.0):
    """range_angle_estimator_impl::work (lib/range_angle_estimator_impl.cc:122-283) on a [n_inputs][vlen] complex64 map.
    `iter_geq == end()` dereferences past the vector in the reference (:169-170); as everywhere in this build that case is
    angle_null_idx = size - 1, then the clamp of :184-187 (docs/history.md §4)."""
    m = np.asarray(m, c64)
    n_inputs, vlen = m.shape
    rb, ab = np.asarray(range_bins, f32), np.asarray(angle_bins, f32)
    peak_power, pr, pa = f32(-1), -1, -1
    for i_range in range(n_inputs):                                                           # :137-151
        for i_angle in range(vlen):
            curr = f32(_pow2_abs(m[i_range, i_angle]))
            if curr > peak_power:
                peak_power, pr, pa = curr, i_range, i_angle
    angle_val, range_val = ab[pa], rb[pr]
    angle_null = f32(angle_val + f32(90))                                                     # :155-160
    if angle_null >= 90:
        angle_null = f32(angle_null - f32(180))
    it = int(np.searchsorted(ab, angle_null, side="left"))                                    # std::lower_bound
    if it == 0:                                                                               # :172-180
        null_idx = 0
    elif it == len(ab):
        null_idx = len(ab) - 1
    elif abs(float(angle_null) - float(ab[it - 1])) < abs(float(angle_null) - float(ab[it])):
        null_idx = it - 1
    else:
        null_idx = it
    if null_idx == len(ab) - 1:                                                               # :184-187
        null_idx = len(ab) - 2
    with np.errstate(all="ignore"):
        dr = int(f32(noise_discard_range_m) / f32(rb[1] - rb[0]))                             # :189 (truncation)
        da = int(f32(noise_discard_angle_deg) / f32(ab[(null_idx + 1) % len(ab)] - ab[null_idx]))
    if da <= 0:
        da = 1
    r0, r1 = pr + len(rb) // 2 - dr, pr + len(rb) // 2 + dr                                    # :197-201
    a0, a1 = null_idx - da, null_idx + da
    noise, n = f32(0), 0
    for i_range in range(r0, r1):                                                             # :209-221; C's % then the fix-up = Python's %
        r_idx = i_range % n_inputs
        for i_angle in range(a0, a1):
            noise = f32(float(noise) + _pow2_abs(m[r_idx, i_angle % vlen]))                   # float += double
            n += 1
    with np.errstate(all="ignore"):
        noise = f32(noise / f32(n)) if n else f32(np.nan)
        snr = f32(f32(10) * f32(_libm.log10f(float(f32(peak_power / noise)))))                # :227
    return dict(peak_range_idx=pr, peak_angle_idx=pa, angle_null_idx=null_idx, discard_range_idx=dr, discard_angle_idx=da,
                n_noise_samples=n, peak_power=peak_power, noise_power=noise, snr_est=snr, range_val=range_val, angle_val=angle_val,
                published=int(bool(snr >= f32(snr_threshold) and peak_power >= f32(power_threshold))))


def fft_peak_detect_ref(x, samp_rate, interp_factor, threshold, samp_protect):
    """fft_peak_detect_impl::work (lib/fft_peak_detect_impl.cc:77-113): (k, freq, phase, mag); k = -1 leaves the outputs untouched"""
    x = np.asarray(x, c64)
    n = x.size
    k, hold = -1, f32(-1)
    thr = 10.0 ** (float(f32(threshold)) / 10.0)                                              # std::pow(10, d_threshold / 10.0): double
    for p in range(samp_protect, n - samp_protect):
        h = f32(_libm.hypotf(float(x[p].real), float(x[p].imag)))
        if h > hold and float(h) * float(h) > thr:
            hold, k = h, p
    if k == -1:
        return -1, None, None, None
    fs_i = f32(f32(samp_rate) * f32(interp_factor))                                           # int * float -> float
    if k <= n // 2:
        freq = f32(f32(f32(k) / f32(n)) * fs_i)                                               # k / (float) n * (samp_rate * interp)
    else:
        freq = f32(-fs_i + f32(f32(k) * f32(fs_i / f32(n))))
    return k, freq, f32(_libm.atan2f(float(x[k].imag), float(x[k].real))), f32(_libm.hypotf(float(x[k].real), float(x[k].imag)))


# ------------------------------------------------------------------------------------------------ lib/stream_encoder_impl.cc, lib/stream_decoder_impl.cc
def stream_encode_values(mcs, n_dc, pdu, scrambler):
    """stream_encoder_impl::general_work (lib/stream_encoder_impl.cc:126-207) up to d_symbol_values: CRC-32 appended (boost::crc_32_type =
    the zlib polynomial, little-endian bytes), generate_bits (16 zero bits + bytes LSB first, lib/utils.cc:137-173), scramble (:175-187),
    reset_tail_bits (:190-193), convolutional_encoding (:207-217), puncturing (:220-250), split_symbols (:281-297; no interleaving).
    Returns (symbol values [n_ofdm_sym * n_dc], n_ofdm_sym, pdu_len tag)."""
    import zlib
    pdu = bytes(pdu)
    size = len(pdu) + 4
    n_bpsc, n_cbps, n_dbps = mcs_params(mcs, n_dc)
    n_sym = int(np.ceil((16 + 8 * size + 6) / float(n_dbps)))
    n_data_bits = n_sym * n_dbps
    n_pad = n_data_bits - (16 + 8 * size + 6)
    data = pdu + int(zlib.crc32(pdu) & 0xffffffff).to_bytes(4, "little")
    bits = np.zeros(n_data_bits, np.uint8)
    for i, byte in enumerate(data):
        for b in range(8):
            bits[16 + i * 8 + b] = (byte >> b) & 1
    state, scr = int(scrambler), np.zeros(n_data_bits, np.uint8)
    for i in range(n_data_bits):
        fb = (1 if state & 64 else 0) ^ (1 if state & 8 else 0)
        scr[i] = fb ^ bits[i]
        state = ((state << 1) & 0x7e) | fb
    scr[n_data_bits - n_pad - 6:n_data_bits - n_pad] = 0
    enc = conv_encode(scr)
    if mcs in (0, 2, 4):
        punct = enc
    else:
        punct = np.array([enc[i] for i in range(2 * n_data_bits) if i % 6 not in (3, 4)], np.uint8)
    assert punct.size == n_sym * n_cbps
    vals = np.zeros(n_sym * n_dc, np.uint8)
    for i in range(vals.size):
        for k in range(n_bpsc):
            vals[i] |= punct[i * n_bpsc + k] << k
    return vals, n_sym, size


def stream_decode_values(mcs, n_dc, data_size_byte, values):
    """stream_decoder_impl::decode (lib/stream_decoder_impl.cc:258-292) + descramble (:406-433) from the decided symbol values:
    bits LSB first per symbol, the windowed Viterbi decoder, descrambler seeded by the first seven decoded bits, CRC-32 residue.
    Returns (crc_ok, payload without the CRC)."""
    import zlib
    n_bpsc, n_cbps, n_dbps = mcs_params(mcs, n_dc)
    n_sym = int(np.ceil((16 + 8 * data_size_byte + 6) / float(n_dbps)))
    vals = np.asarray(values, np.uint8)[:n_sym * n_dc]
    bits = np.zeros(n_sym * n_cbps, np.uint8)
    for i, v in enumerate(vals):
        for k in range(n_bpsc):
            bits[i * n_bpsc + k] = (int(v) >> k) & 1
    dec = ViterbiWindowed().decode(mcs, n_sym, n_cbps, n_sym * n_dbps, bits)
    dec = np.concatenate([dec, np.zeros(64, np.uint8)])
    state = 0
    for i in range(7):
        if dec[i]:
            state |= 1 << (6 - i)
    out = bytearray(data_size_byte + 2 + 8)
    out[0] = state
    for i in range(7, data_size_byte * 8 + 16):
        fb = (1 if state & 64 else 0) ^ (1 if state & 8 else 0)
        bit = fb ^ int(dec[i] & 1)
        out[i // 8] |= bit << (i % 8)
        state = ((state << 1) & 0x7e) | fb
    body = bytes(out[2:2 + data_size_byte])
    # boost::crc_32_type over message + little-endian CRC leaves the residue 558161692 (= 0x2144DF1C) exactly when the CRC matches
    ok = (zlib.crc32(body) & 0xffffffff) == 558161692
    return ok, body[:max(data_size_byte - 4, 0)]


# ------------------------------------------------------------------------------------------------ lib/moving_avg_impl.cc, lib/frame_detector_impl.cc
def moving_avg_ref(buf, length, scale, n_out, max_iter=16000):
    """moving_avg_impl::work (lib/moving_avg_impl.cc:62-98): `buf` = the length-1 history items followed by the new ones (set_history)"""
    buf = np.asarray(buf, c64)
    n = min(n_out, max_iter)
    sr, si = f32(buf[0].real), f32(buf[0].imag)
    for i in range(1, length - 1):
        sr, si = f32(sr + buf[i].real), f32(si + buf[i].imag)
    out = np.zeros(n, c64)
    for i in range(n):
        sr, si = f32(sr + buf[i + length - 1].real), f32(si + buf[i + length - 1].imag)
        out[i] = complex(f32(sr * f32(scale)), f32(si * f32(scale)))
        sr, si = f32(sr - buf[i].real), f32(si - buf[i].imag)
    return out


class FrameDetectorRef:
    """frame_detector_impl::general_work (lib/frame_detector_impl.cc:70-193), one call per work(); nitems_read / nitems_written kept here"""

    def __init__(self, fft_len, cp_len, threshold, min_n_peaks, ignore_gap):
        self.fft_len, self.threshold, self.min_n_peaks, self.ignore_gap = fft_len, float(threshold), int(min_n_peaks), int(ignore_gap)
        self.MAX_PEAK_VALUE, self.MAX_PEAK_DISTANCE, self.MAX_SAMPLES = 2.0, 2 * (fft_len + cp_len), 540 * (fft_len + cp_len)
        self.state, self.n_peaks, self.cfo, self.copied, self.first_peak = "SEARCH", 0, f32(0), 0, 0
        self.nread = self.nwritten = 0

    def work(self, x, in_abs, in_cor, noutput):
        x, in_abs, in_cor = np.asarray(x, c64), np.asarray(in_abs, c64), np.asarray(in_cor, f32)
        ninput = min(x.size, in_abs.size, in_cor.size)
        tags = []

        def is_peak(v):
            return float(v) > self.threshold and float(v) < self.MAX_PEAK_VALUE

        def u64(v):                                   # nitems_read(0) + n - first_peak_ind is uint64_t arithmetic
            return v & 0xFFFFFFFFFFFFFFFF

        if self.state == "SEARCH":                                                             # :89-130
            n_in = 0
            while n_in < ninput:
                if is_peak(in_cor[n_in]):
                    if self.n_peaks < self.min_n_peaks:
                        self.n_peaks += 1
                        if self.n_peaks == 1:
                            self.first_peak = self.nread + n_in
                    elif u64(self.nread + n_in - self.first_peak) < self.MAX_PEAK_DISTANCE:
                        self.state, self.copied = "COPY", 0
                        self.cfo = f32(float(f32(carg(in_abs[n_in]))) / (self.fft_len / 4.0))
                        self.n_peaks, self.first_peak = 0, 0
                        tags.append((self.nwritten, float(self.cfo)))
                        break
                    else:
                        self.n_peaks, self.first_peak = 0, 0
                elif u64(self.nread + n_in - self.first_peak) > self.MAX_PEAK_DISTANCE:
                    self.n_peaks, self.first_peak = 0, 0
                n_in += 1
            self.nread += n_in
            return np.zeros(0, c64), n_in, tags
        out = []                                                                               # COPY :132-186
        n_out = 0
        while n_out < ninput and n_out < noutput and self.copied < self.MAX_SAMPLES:
            if is_peak(in_cor[n_out]):
                if self.n_peaks < self.min_n_peaks:
                    self.n_peaks += 1
                    if self.n_peaks == 1:
                        self.first_peak = self.nread + n_out
                elif u64(self.nread + n_out - self.first_peak) < self.MAX_PEAK_DISTANCE:
                    if self.copied > self.ignore_gap:
                        self.copied, self.n_peaks, self.first_peak = 0, 0, 0
                        self.cfo = f32(float(f32(carg(in_abs[n_out]))) / (self.fft_len / 4.0))
                        tags.append((self.nwritten + n_out, float(self.cfo)))
                        break
                else:
                    self.n_peaks, self.first_peak = 0, 0
            elif u64(self.nread + n_out - self.first_peak) > self.MAX_PEAK_DISTANCE:
                self.n_peaks, self.first_peak = 0, 0
            out.append(cmul(x[n_out], cexp_j(f32(-self.cfo * f32(self.copied))))[()])          # float * int -> float
            n_out += 1
            self.copied += 1
        if self.copied == self.MAX_SAMPLES:
            self.state = "SEARCH"
        self.nread += n_out
        self.nwritten += n_out
        return np.array(out, c64), n_out, tags


# ------------------------------------------------------------------------------------------------ lib/target_simulator_impl.cc
_libm.sin.restype = ctypes.c_double
_libm.sin.argtypes = [ctypes.c_double]
_libm.fmod.restype = ctypes.c_double
_libm.fmod.argtypes = [ctypes.c_double, ctypes.c_double]
_libm.pow.restype = ctypes.c_double
_libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
_libm.sqrtf.restype = ctypes.c_float
_libm.sqrtf.argtypes = [ctypes.c_float]
GR_M_PI = 3.14159265358979323846


class TargetSimulatorRef:
    """target_simulator_impl: setup_targets (lib/target_simulator_impl.cc:127-198) and work (:201-384), float / double promotions as written.
    The two gr::fft::fft_complex calls (FFTW3f, not in the tree) are numpy's complex128 transform of the float data rounded once to float,
    the VOLK products the generic kernel's (ac - bd, ad + bc) in float: the outputs agree with a GNU Radio build to FFT rounding, the
    set-up vectors and the two filters are meant bit for bit."""
    C_LIGHT = f32(3e8)                                  # target_simulator_impl.h:87 constexpr static float
    FOUR_PI_CUBED_SQRT = 44.54662397465366              # :33

    def __init__(self, range_m, velocity, rcs, azimuth, position_rx, samp_rate, center_freq, self_coupling_db, rndm_phaseshift, self_coupling):
        self.range, self.velocity, self.rcs, self.azimuth, self.position_rx = (np.asarray(v, f32) for v in (range_m, velocity, rcs, azimuth, position_rx))
        self.samp_rate, self.center_freq = int(samp_rate), f32(center_freq)
        self.self_coupling_db, self.rndm_phaseshift, self.self_coupling = f32(self_coupling_db), bool(rndm_phaseshift), bool(self_coupling)
        K, R = self.range.size, self.position_rx.size
        # :164  2 * d_velocity[k] * d_center_freq / c_light — int * float, all float
        self.doppler = np.array([f32(f32(f32(f32(2) * v) * self.center_freq) / self.C_LIGHT) for v in self.velocity], f32)
        # :177  (2.0 * range - position * std::sin(azimuth * GR_M_PI / 180.0)) / c_light — double, stored into a vector<float>
        self.timeshift = np.zeros((R, K), f32)
        for l in range(R):
            for k in range(K):
                s = _libm.sin(float(self.azimuth[k]) * GR_M_PI / 180.0)
                self.timeshift[l, k] = f32((2.0 * float(self.range[k]) - float(self.position_rx[l]) * s) / float(self.C_LIGHT))
        # :188  c_light * std::sqrt(rcs) [float] / FOUR_PI_CUBED_SQRT [double] / (range * range) [float product] / center_freq
        self.scale_ampl = np.array([f32(float(f32(self.C_LIGHT * f32(_libm.sqrtf(float(s))))) / self.FOUR_PI_CUBED_SQRT / float(f32(r * r)) / float(self.center_freq))
                                    for s, r in zip(self.rcs, self.range)], f32)
        self.buff_size = 2                               # constructor :89

    def filters(self, n):
        """:249-305"""
        K, R = self.range.size, self.position_rx.size
        sr = f32(self.samp_rate)
        i = np.arange(n)
        base = (i.astype(f32) * sr) / f32(n)             # i * (float)d_samp_rate / (float)n_input: int * float, / float
        self.freq = np.where(i < n // 2, base, base - sr).astype(f32)
        self.filt_doppler = np.zeros((K, n), c64)
        self.filt_time = np.zeros((R, K, n), c64)
        for k in range(K):
            phase = f32(0)                                # imag(d_phase_doppler)
            step = 2 * GR_M_PI * float(self.doppler[k]) / float(sr)
            for j in range(n):
                e = cexp_j(phase)
                self.filt_doppler[k, j] = cplx(f32(e.real) * self.scale_ampl[k], f32(e.imag) * self.scale_ampl[k])    # complex<float> * float
                phase = f32(_libm.fmod(float(phase) + step, 2 * GR_M_PI))
            for l in range(R):
                for j in range(n):
                    ph = f32(_libm.fmod(2 * GR_M_PI * float(self.timeshift[l, k]) * float(f32(self.freq[j] + self.center_freq)), 2 * GR_M_PI))
                    e = cexp_j(-ph)                       # std::exp(-d_phase_time): (0, ph) negated is (-0, -ph); cexpf(-0 - j ph) = (cos, -sin)
                    self.filt_time[l, k, j] = cplx(f32(e.real) / f32(n), f32(e.imag) / f32(n))
        self.buff_size = n

    @staticmethod
    def _fft(x, forward):
        z = np.asarray(x, c64).astype(np.complex128)
        y = np.fft.fft(z) if forward else np.fft.ifft(z) * z.size            # fft_complex(n, false) is the unnormalised backward transform
        return y.astype(c64)

    def work(self, x, target_phase=None):
        x = np.asarray(x, c64)
        n = x.size
        if self.buff_size != n:
            self.filters(n)
        R, K = self.position_rx.size, self.range.size
        out = np.zeros((R, n), c64)
        for l in range(R):
            o = np.zeros(n, c64)                          # :338
            for k in range(K):
                bt = cmul(x, self.filt_doppler[k])        # :345
                fo = self._fft(bt, True)
                bf = cmul(fo, self.filt_time[l, k])       # :352
                fo = self._fft(bf, False)
                if self.rndm_phaseshift and target_phase is not None:
                    o = cmul(fo, np.full(n, target_phase[k], c64))       # :360: the product is written to `out`
                else:
                    o = fo.copy()                         # :364 memcpy: every target OVERWRITES out, the last one leaves the block
            if self.self_coupling:                        # :372-378  out[i] += (gr_complex)pow(10, db / 20.0) * in[i]
                sc = c64(complex(f32(_libm.pow(10.0, float(self.self_coupling_db) / 20.0)), 0.0))
                o = cadd(o, cmul(np.full(n, sc, c64), x))
            out[l] = o
        return out


# ------------------------------------------------------------------------------------------------ lib/frame_sync_impl.cc
class FrameSyncRef:
    """frame_sync_impl::general_work (lib/frame_sync_impl.cc:89-229) and search_frame_start (:231-287), one call = work().
    gr::filter::kernel::fir_filter_ccc::filterN (gr-filter 3.8, not in the tree) is taken as out[i] = sum_k taps[k] in[i + ntaps - 1 - k]
    summed in index order (VOLK's dot product may order the sum differently: last bits of the correlation, which only ranks peaks);
    std::list::sort is a stable merge sort, as Python's sorted()."""

    def __init__(self, fft_len, cp_len, sync_length, ltf_seq_time):
        self.fft_len, self.cp_len, self.SYNC_LENGTH = int(fft_len), int(cp_len), int(sync_length)
        self.taps = np.asarray(ltf_seq_time, c64)
        self.state = "SYNC"
        self.sample_offset = 0
        self.cor = []
        self.frame_start = 0
        self.freq_offset = f32(0)                        # frame_sync_impl.h: float d_freq_offset
        self.cfo_coarse_est = 0.0                        # double
        self.total_out_count = 0
        self.nread = self.nwritten = 0

    def _search_frame_start(self):
        assert len(self.cor) == self.SYNC_LENGTH
        vec = sorted(self.cor, key=lambda p: -cabs(p[0]))                # compare_abs2: abs(first) > abs(second), stable
        self.cor = []
        self.frame_start = self.SYNC_LENGTH                             # :242
        N = self.fft_len
        for i in range(3):
            for k in range(i + 1, 4):
                if vec[i][1] > vec[k][1]:
                    first, second = vec[k][0], vec[i][0]
                else:
                    first, second = vec[i][0], vec[k][0]
                diff = abs(vec[i][1] - vec[k][1])
                if diff in (N, N - 1, N + 1):
                    self.frame_start = min(vec[i][1], vec[k][1])
                    self.freq_offset = f32(f32(carg(cmul(first, cconj(second))[()])) / f32(diff))      # float / int
                    if diff == N:
                        return                                          # :264 "nice match found, return immediately"

    def work(self, x, x_delayed, tags, noutput):
        x, xd = np.asarray(x, c64), np.asarray(x_delayed, c64)
        ninput = min(x.size, xd.size, 8192)                             # :111
        out, otags = [], []
        hits = sorted(t for t in tags if self.nread <= t[0] < self.nread + ninput)
        if hits:                                                        # :116-146
            off, val = hits[0]
            if off > self.nread:
                ninput = off - self.nread
            else:
                if self.sample_offset and self.state == "SYNC":
                    raise RuntimeError("[FRAME SYNC] Something is wrong!")
                if self.state == "COPY":
                    self.state = "RESET"
                self.cfo_coarse_est = float(val)
        n_in = n_out = 0
        if self.state == "SYNC":                                        # :153-181
            ncor = min(self.SYNC_LENGTH, max(ninput - self.fft_len - 1, 0))
            nt = self.taps.size
            corr = np.zeros(self.SYNC_LENGTH + 8192, c64)               # d_correlation keeps what earlier calls left; positions >= ncor are
            for i in range(ncor):                                       # read below only when ninput is within two samples of fft_len
                acc = c64(0)
                for k in range(nt):
                    acc = cadd(acc, cmul(self.taps[k], x[i + nt - 1 - k]))[()]
                corr[i] = acc
            while n_in + self.fft_len - 1 < ninput:
                self.cor.append((corr[n_in], self.sample_offset))
                n_in += 1
                self.sample_offset += 1
                if self.sample_offset == self.SYNC_LENGTH:
                    self._search_frame_start()
                    self.sample_offset = 0
                    self.total_out_count = 0
                    self.state = "COPY"
                    break
        elif self.state == "COPY":                                      # :183-204
            while n_in < ninput and n_out < noutput:
                rel = self.sample_offset - self.frame_start
                if rel == 0:
                    otags.append((self.nwritten, self.cfo_coarse_est - float(self.freq_offset)))
                if rel >= 0 and (rel < 2 * self.fft_len or ((rel - 2 * self.fft_len) % (self.fft_len + self.cp_len)) > self.cp_len - 1):
                    out.append(cmul(xd[n_in], cexp_j(f32(f32(self.sample_offset) * self.freq_offset)))[()])     # int * float -> float
                    n_out += 1
                n_in += 1
                self.sample_offset += 1
        else:                                                           # RESET :206-222
            while n_out < noutput:
                if (self.total_out_count + n_out) % self.fft_len == 0:
                    self.sample_offset = 0
                    self.state = "SYNC"
                    break
                out.append(c64(0))
                n_out += 1
        self.total_out_count += n_out
        self.nread += n_in
        self.nwritten += n_out
        return np.array(out, c64), n_in, otags
