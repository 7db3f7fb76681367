"""CPU tier: the bit-codec restatement (oracle/jrc_oracle_codec.c; reference lib/stream_encoder_impl.cc, lib/utils.cc,
lib/stream_decoder_impl.cc, lib/viterbi_decoder.cc) against independent restatements: zlib's CRC-32, a numpy GF(2)
encoder, and the full-traceback maximum-likelihood Viterbi of oracle/jrc_oracle_comm.c."""
import zlib

import numpy as np
import pytest

import oracle

N_DC = 48


def np_encode(mcs, n_dc, psdu, init):
    """numpy restatement of the encoder chain up to the punctured bit stream"""
    pp = oracle.packet_params(mcs, n_dc, len(psdu) + 4)
    pkt = bytes(psdu) + int(zlib.crc32(bytes(psdu))).to_bytes(4, "little")
    bits = np.zeros(pp["n_data_bits"], np.uint8)
    bits[16:16 + 8 * len(pkt)] = np.unpackbits(np.frombuffer(pkt, np.uint8), bitorder="little")
    seq = np.zeros(pp["n_data_bits"], np.uint8)          # x^7 + x^4 + 1 scrambler
    state = init
    for i in range(seq.size):
        fb = ((state >> 6) & 1) ^ ((state >> 3) & 1)
        seq[i] = fb
        state = ((state << 1) & 0x7e) | fb
    sc = bits ^ seq
    tail = pp["n_data_bits"] - pp["n_pad_bits"] - 6
    sc[tail:tail + 6] = 0
    padded = np.concatenate([np.zeros(6, np.uint8), sc])
    g0 = [1, 0, 1, 1, 0, 1, 1]        # 0155 = 1101101b: taps on delays 0,2,3,5,6
    g1 = [1, 1, 1, 1, 0, 0, 1]        # 0117 = 1001111b: taps on delays 0,1,2,3,6
    enc = np.zeros(2 * sc.size, np.uint8)
    for d in range(7):
        if g0[d]:
            enc[0::2] ^= padded[6 - d:6 - d + sc.size]
        if g1[d]:
            enc[1::2] ^= padded[6 - d:6 - d + sc.size]
    if mcs % 2 == 1:
        keep = np.ones(enc.size, bool)
        keep[3::6] = False
        keep[4::6] = False
        enc = enc[keep]
    return enc, pp


@pytest.mark.parametrize("n", [0, 1, 9, 1000])
def test_crc32_is_zlib(n):
    d = np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8).tobytes()
    assert oracle.crc32(d) == zlib.crc32(d)
    if n:
        assert oracle.crc32(d + zlib.crc32(d).to_bytes(4, "little")) == 558161692      # the residue the decoder tests (:246)


@pytest.mark.parametrize("mcs", range(6))
@pytest.mark.parametrize("nbytes,init", [(1, 1), (33, 93), (500, 127)])
def test_encoder_matches_numpy_restatement(mcs, nbytes, init):
    rng = np.random.default_rng(mcs * 100 + nbytes)
    psdu = bytes([2]) + rng.integers(0, 256, nbytes - 1, dtype=np.uint8).tobytes()
    sym, tags = oracle.stream_encode(mcs, N_DC, psdu, init)
    enc, pp = np_encode(mcs, N_DC, psdu, init)
    assert tags == dict(packet_len=pp["n_ofdm_sym"] * N_DC, packet_type=2, mcs=mcs, pdu_len=nbytes + 4)
    bpsc = pp["n_bpsc"]
    vals = (enc.reshape(-1, bpsc) << np.arange(bpsc)).sum(1)
    want = np.array([oracle.constellation_point(bpsc, int(v)) for v in vals], np.complex64)
    np.testing.assert_array_equal(sym, want)
    # every constellation maps back onto itself, and QPSK carries the encoder's 1/2 (:218-221)
    assert all(oracle.constellation_decide(bpsc, oracle.constellation_point(bpsc, v)) == v for v in range(1 << bpsc))
    if bpsc == 2:
        assert abs(abs(sym[0]) - 0.5) < 1e-6
    if bpsc == 4:
        assert abs(np.mean([abs(oracle.constellation_point(4, v)) ** 2 for v in range(16)]) - 1.0) < 1e-6


def test_encoder_refuses_oversized_pdu():
    assert oracle.stream_encode(2, N_DC, bytes(3097))[0] is None          # len + 4 > MAX_PAYLOAD_SIZE (:139-143)
    assert oracle.stream_encode(2, N_DC, bytes(3096))[0] is not None


@pytest.mark.parametrize("mcs", range(6))
@pytest.mark.parametrize("nbytes", [5, 64, 777])
def test_round_trip_clean_and_noisy(mcs, nbytes):
    rng = np.random.default_rng(mcs + 10 * nbytes)
    psdu = bytes([2]) + rng.integers(0, 256, nbytes - 1, dtype=np.uint8).tobytes()
    sym, tags = oracle.stream_encode(mcs, N_DC, psdu, 1 + (nbytes % 127))
    ok, payload = oracle.stream_decode(mcs, N_DC, tags["pdu_len"], sym)
    assert ok and payload == psdu
    sigma = {1: 0.25, 2: 0.12, 4: 0.05}[oracle.packet_params(mcs, N_DC, 8)["n_bpsc"]]
    noisy = sym + sigma * (rng.standard_normal(sym.size) + 1j * rng.standard_normal(sym.size)).astype(np.complex64)
    ok, payload = oracle.stream_decode(mcs, N_DC, tags["pdu_len"], noisy)
    assert ok and payload == psdu


@pytest.mark.parametrize("mcs", [0, 2])
def test_windowed_viterbi_equals_full_traceback_at_low_error_rates(mcs):
    """rate 1/2: the chunked traceback (5 x 8 bits) and the maximum-likelihood full traceback agree while errors are sparse"""
    rng = np.random.default_rng(mcs)
    psdu = bytes([1]) + rng.integers(0, 256, 299, dtype=np.uint8).tobytes()
    enc, pp = np_encode(mcs, N_DC, psdu, 77)
    rx = enc.copy()
    flips = rng.choice(rx.size - 200, 12, replace=False)
    rx[flips] ^= 1
    win = oracle.viterbi_windowed(mcs, pp["n_ofdm_sym"], pp["n_cbps"], pp["n_data_bits"], rx)
    ml = oracle.viterbi_k7(rx)
    n = pp["n_data_bits"] - pp["n_pad_bits"]
    np.testing.assert_array_equal(win[:n], ml[:n])


def test_bad_crc_and_refused_frames():
    rng = np.random.default_rng(5)
    psdu = bytes([2]) + rng.integers(0, 256, 99, dtype=np.uint8).tobytes()
    sym, tags = oracle.stream_encode(3, N_DC, psdu, 11)
    bad = sym.copy()
    bad[100:160] = -bad[100:160]                                           # a burst of errors the code cannot repair
    ok, payload = oracle.stream_decode(3, N_DC, tags["pdu_len"], bad)
    assert ok is False and len(payload) == len(psdu)
    assert oracle.stream_decode(0, N_DC, 3101, np.zeros(60000, np.complex64)) == (None, None)     # :133-146
