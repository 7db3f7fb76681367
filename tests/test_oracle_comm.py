"""CPU tier: comm-side oracle (SIG codec, equalizer, precoder, steering) against closed forms and round trips,
parameterised with the constant tables minted from the reference's ofdm_config module."""
import numpy as np
import pytest

import oracle
from conftest import crandn, rel_err

NDP, DATA = 1, 2
LS, STA = 0, 1


def conv_encode_np(bits):
    """independent restatement of the K=7 (0155, 0117) encoder as GF(2) polynomial products"""
    g0 = [(0o155 >> k) & 1 for k in range(7)]        # tap k applies to the input k steps ago
    g1 = [(0o117 >> k) & 1 for k in range(7)]
    b = np.concatenate([np.zeros(6, int), np.asarray(bits, int)])
    out = np.zeros(2 * len(bits), np.uint8)
    for i in range(len(bits)):
        w = b[i:i + 7][::-1]                          # w[k] = input k steps ago
        out[2 * i] = np.dot(g0, w) % 2
        out[2 * i + 1] = np.dot(g1, w) % 2
    return out


@pytest.mark.parametrize("mcs", range(6))
@pytest.mark.parametrize("ptype", [NDP, DATA])
@pytest.mark.parametrize("length", [0, 1, 100, 1500, 4095])
def test_sig_field_round_trip(mcs, ptype, length):
    nd = 48
    sym = oracle.sig_encode(nd, mcs, ptype, length)
    assert set(np.unique(sym)) <= {-1.0, 1.0}
    bits = (sym > 0).astype(np.uint8)
    dec = oracle.viterbi_k7(bits)
    ok, m, p, ln, ns = oracle.sig_parse(dec, nd)
    assert ok and (m, p, ln) == (mcs, ptype, length)
    assert ns == oracle.n_ofdm_sym(mcs, nd, length)
    hdr = dec[:24]                                            # header layout (lib/mimo_precoder_impl.cc:1003-1033)
    assert hdr[17] == hdr[:17].sum() % 2 and not hdr[18:].any()
    assert np.array_equal(conv_encode_np(hdr), bits)          # encoder is the (0155, 0117) code, not interleaved


def test_sig_field_survives_channel_errors_and_detects_parity():
    sym = oracle.sig_encode(48, 2, DATA, 777)
    bits = (sym > 0).astype(np.uint8)
    bad = bits.copy()
    bad[[5, 30]] ^= 1                                          # two isolated errors: free distance 10 corrects them
    ok, m, p, ln, _ = oracle.sig_parse(oracle.viterbi_k7(bad), 48)
    assert ok and (m, p, ln) == (2, DATA, 777)
    # the reference's check only fires when bits 17..22 are ALL zero, i.e. it also requires the parity bit
    # itself to be 0 (lib/mimo_ofdm_equalizer_impl.cc:695-702): a corrupted header whose parity bit is 1 passes
    seen = set()
    for length in range(700, 740):
        hdr = oracle.viterbi_k7((oracle.sig_encode(48, 2, DATA, length) > 0).astype(np.uint8))
        p17 = int(hdr[17])
        hdr[6] ^= 1                                            # wrong length bit -> parity mismatch
        assert oracle.sig_parse(hdr, 48)[0] == bool(p17)
        seen.add(p17)
    assert seen == {0, 1}


def test_n_ofdm_sym_formula():
    for mcs, dbps in [(0, 24), (1, 36), (2, 48), (3, 72), (4, 96), (5, 144)]:
        for nbytes in (0, 1, 50, 999):
            assert oracle.n_ofdm_sym(mcs, 48, nbytes) == int(np.ceil((16 + 8 * nbytes + 6) / dbps))


def test_steering_matrix_properties():
    rng = np.random.default_rng(0)
    for T in (2, 4):
        h = crandn(rng, T)
        Q = oracle.steering_from_channel(h)
        V = Q * np.linalg.norm(Q) / np.sqrt(T)
        assert np.allclose(Q.conj().T @ Q, np.eye(T), atol=2e-6)            # unitary (||V||_F = sqrt(T))
        row = h @ Q                                                           # h^T V = [beta, 0, ..., 0]
        assert abs(abs(row[0]) - np.linalg.norm(h)) < 1e-5 and np.abs(row[1:]).max() < 1e-5
        assert abs(abs(np.vdot(np.conj(h), Q[:, 0])) / np.linalg.norm(h) - 1) < 1e-5
        Qp = oracle.steering_from_channel(h, phased=True)
        assert np.allclose(Qp[:, 0], np.conj(h) * np.sqrt(T) / np.linalg.norm(h), atol=1e-6) and not Qp[:, 1:].any()
    F = oracle.dft_matrix(4)
    k = np.arange(4)
    assert np.allclose(F, np.exp(-2j * np.pi * np.outer(k, k) / 4) / 2, atol=1e-6)


def make_blocks(o, est=LS, T=4):
    dc, pc = o["data_subcarriers"], o["pilot_subcarriers"]
    ps, sw, ml, ltf = o["pilot_symbols"], o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], o["ltf_64"]
    pre = oracle.Precoder(64, T, 1, dc, pc, ps, sw, ml)
    eq = oracle.Equalizer(est, 24e9, 125e6, 64, 16, dc, pc, ps, ltf, ml, T)
    return pre, eq


def through_channel(tx, h, noise=0.0, rng=None):
    """flat MISO channel per subcarrier; equalizer input = [LTF, LTF, SIG, MIMO-LTFs, data] (frame_sync drops the STFs)"""
    y = np.tensordot(h, tx, axes=(0, 0))                      # [n_total, N]
    y = np.concatenate([y[3:4], y[3:]], axis=0)
    if noise:
        y = y + noise * (rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape))
    return y.astype(np.complex64)


def qpsk(rng, n):
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    return pts[rng.integers(0, 4, n)].astype(np.complex64)


def qam16(rng, n):
    return np.array([oracle.constellation_point(4, int(v)) for v in rng.integers(0, 16, n)], np.complex64)


@pytest.mark.parametrize("ptype", [NDP, DATA])
def test_sta_estimator_with_16qam_takes_decisions_from_the_16qam_table(ofdm64, ptype):
    """STA x 16-QAM (lib/mimo_ofdm_equalizer_impl.cc:505-521, :563-570): the decision-directed update divides by the decided 16-QAM
    point (unscaled: only QPSK is halved, :511).  On a noise-free channel the decisions are right, so the update leaves the estimate
    where it is: the STA output equals the LS output to rounding, and both return the transmitted points.  A QPSK-style decision
    (the former behaviour of this restatement) would pull the estimate away."""
    rng = np.random.default_rng(5)
    nbytes, mcs = 60, 4
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    s = qam16(rng, ns * 48)
    assert len(np.unique(np.round(np.abs(s) ** 2 * 10))) == 3               # 0.2, 1.0, 1.8: all three rings present
    out = {}
    for est in (LS, STA):
        pre, eq = make_blocks(ofdm64, est)
        tx = pre.work(s, mcs, ptype, nbytes)
        h = crandn(rng, 4) if est == LS else h
        out[est] = eq.general_work(through_channel(tx, h), [(0, 0.0)])["out"]
    assert rel_err(out[LS], s.reshape(ns, 48)) < 1e-5
    assert rel_err(out[STA], out[LS]) < 1e-5
    # decisions and points agree with the codec's table
    for z in (0.1 + 0.1j, 0.7 - 0.2j, -0.64 + 0.9j, -0.2 - 0.63j):
        v = oracle.constellation_decide(4, z)
        assert abs(oracle.constellation_point(4, v) - z) == min(abs(oracle.constellation_point(4, k) - z) for k in range(16))


@pytest.mark.parametrize("est", [LS, STA])
def test_ndp_frame_round_trip_and_channel_estimate(ofdm64, est):
    rng = np.random.default_rng(1)
    pre, eq = make_blocks(ofdm64, est)
    nbytes, mcs = 30, 2
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    s = qpsk(rng, ns * 48)
    tx = pre.work(s, mcs, NDP, nbytes)
    assert tx.shape == (4, 4 + 1 + 4 + ns, 64) and not tx[2:, :5].any()        # legacy preamble only on outputs 0, 1
    h = crandn(rng, 4)
    r = eq.general_work(through_channel(tx, h), [(0, 0.0)])
    assert r["consumed"] == ns + 7 and r["out"].shape == (ns, 48)
    assert rel_err(r["out"], s.reshape(ns, 48)) < 1e-5
    ev = r["events"]
    assert [e["kind"] for e in ev] == [1, 2] and ev[0]["offset"] == 0 and ev[1]["offset"] == ns - 1
    assert (ev[0]["data_bytes"], ev[0]["mcs"], ev[0]["packet_type"]) == (nbytes, mcs, NDP)
    ltf = ofdm64["ltf_64"].real
    exp = (4 * np.outer(ltf * ltf, h)).astype(np.complex64)                    # H = N_ltf * |ltf|^2 * h (no 1/N_ltf, :398)
    assert rel_err(r["chan_est"], exp) < 1e-5
    assert np.allclose(ev[1]["chan_mean"], 4 * h, atol=1e-4)


@pytest.mark.parametrize("steer", ["dft", "svd_mean", "svd_sc", "phased"])
def test_data_frame_precoded_round_trip(ofdm64, steer):
    rng = np.random.default_rng(2)
    pre, eq = make_blocks(ofdm64)
    nbytes, mcs = 61, 2
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    s = qpsk(rng, ns * 48)
    h = crandn(rng, 4)
    kw = {}
    if steer == "svd_mean":
        kw = dict(steer_mode=1, Q_mean=oracle.steering_from_channel(h))
    elif steer == "phased":
        kw = dict(steer_mode=1, Q_mean=oracle.steering_from_channel(h, phased=True))
    elif steer == "svd_sc":
        kw = dict(steer_mode=2, Q_sc=np.stack([oracle.steering_from_channel(h)] * 64))
    tx = pre.work(s, mcs, DATA, nbytes, **kw)
    if steer == "dft":
        assert np.allclose(tx[:, 9:, ofdm64["data_subcarriers"][0] + 32], np.outer(np.full(4, 0.5), s.reshape(ns, 48)[:, 0]), atol=1e-6)
    r = eq.general_work(through_channel(tx, h, 1e-4, rng), [(0, 0.0)])
    assert r["out"].shape == (ns, 48) and rel_err(r["out"], s.reshape(ns, 48)) < 5e-3
    assert r["events"][0]["packet_type"] == DATA and r["events"][1]["snr_data"] > 30
    if steer in ("svd_mean", "svd_sc"):                                         # beamforming gain: |h^T q0| = ||h||
        assert abs(abs(r["events"][1]["chan_mean"][0]) - np.linalg.norm(h)) < 2e-2


def test_radar_streams_are_orthogonal_to_the_user(ofdm64):
    rng = np.random.default_rng(3)
    pre, eq = make_blocks(ofdm64)
    nbytes, mcs = 20, 2
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    s = qpsk(rng, ns * 48)
    h = crandn(rng, 4)
    rs = qpsk(rng, 3 * ns * 64).reshape(3, ns, 64)
    tx = pre.work(s, mcs, DATA, nbytes, steer_mode=1, Q_mean=oracle.steering_from_channel(h), radar_streams=rs)
    r = eq.general_work(through_channel(tx, h), [(0, 0.0)])
    assert rel_err(r["out"], s.reshape(ns, 48)) < 1e-4       # null-space streams do not reach the user


def test_precoder_rejects_inconsistent_length(ofdm64):
    pre, _ = make_blocks(ofdm64)
    with pytest.raises(RuntimeError):
        pre.work(np.zeros(48 * 3, np.complex64), 2, DATA, 500)


def test_equalizer_skips_without_frame_start_and_after_frame_end(ofdm64):
    rng = np.random.default_rng(4)
    pre, eq = make_blocks(ofdm64)
    ns = oracle.n_ofdm_sym(2, 48, 10)
    s = qpsk(rng, ns * 48)
    y = through_channel(pre.work(s, 2, NDP, 10), crandn(rng, 4))
    r = eq.general_work(y)                                   # no tag yet: everything is consumed, nothing produced
    assert r["consumed"] == len(y) and len(r["out"]) == 0
    y2 = np.concatenate([y, crandn(rng, 5, 64)])             # trailing garbage after the frame is skipped (:250-255)
    r = eq.general_work(y2, [(0, 0.0)])
    assert r["consumed"] == len(y2) and r["out"].shape == (ns, 48)
    assert rel_err(r["out"], s.reshape(ns, 48)) < 1e-5


def test_windowed_decoder_returns_the_input_of_any_clean_codeword():
    """the claim behind the equalizer's SIG shortcut (comm.hip sig_viterbi_wave): when the 124 hard decisions the reference's windowed
    decoder reads (zeros beyond the symbol's cells) are the K=7 encoder's output for some input sequence, the decoder returns that
    sequence — and the sequence is u_t = c0[t-2] ^ c0[t-4] ^ c1[t] ^ c1[t-1] ^ c1[t-2] ^ c1[t-3] ^ c1[t-4]
    ((D^2 + D^4) g0 + (1 + D + D^2 + D^3 + D^4) g1 = 1 over GF(2)).  Clean fields, fields with a non-zero tail (not flushed: rejected at
    48 cells, a codeword at 224), and single flipped decisions (always rejected: the shortcut never fires on a word that is not a codeword)."""
    def encode(u, nsteps):
        st, out = 0, []
        for i in range(nsteps):
            st = ((st << 1) & 0x7e) | (int(u[i]) if i < len(u) else 0)
            out += [bin(st & 0o155).count("1") & 1, bin(st & 0o117).count("1") & 1]
        return np.array(out, np.uint8)

    def shortcut(word124):
        c0, c1 = word124[0::2].astype(int), word124[1::2].astype(int)
        g = lambda a, i: a[i] if i >= 0 else 0
        u = [g(c0, t - 2) ^ g(c0, t - 4) ^ g(c1, t) ^ g(c1, t - 1) ^ g(c1, t - 2) ^ g(c1, t - 3) ^ g(c1, t - 4) for t in range(62)]
        return u if np.array_equal(encode(u, 62), word124) else None

    rng = np.random.default_rng(0)
    fired = rejected = 0
    for ND in (48, 112, 224):
        for trial in range(400):
            u = rng.integers(0, 2, 24)
            if trial % 2 == 0:
                u[18:] = 0                                        # a proper header: tail zeros
            cells = encode(u, ND // 2)                            # generate_signal_field: the encoder runs over all ND cells (:1040-1054)
            word = np.zeros(124, np.uint8)
            word[:min(ND, 124)] = cells[:124]
            got = shortcut(word)
            dec = oracle.viterbi_windowed(0, 1, ND, 24, cells)[:24]
            if got is not None:
                fired += 1
                assert np.array_equal(dec, np.array(got[:24], np.uint8)) and np.array_equal(dec, u)
            else:
                rejected += 1
                assert ND == 48 and u[18:].any()                  # only an unflushed tail cut off by the symbol's end
            k = int(rng.integers(0, min(ND, 124)))
            word[k] ^= 1
            assert shortcut(word) is None                         # distance 1 from a codeword is not a codeword (d_free = 10)
    assert fired > 1000 and rejected > 50


def test_clean_rate_three_quarter_frames_with_a_short_pad_can_fail_their_crc():
    """a property of the reference's decoder that the restatement keeps (found by the codec blocks' scheduling fuzz): viterbi_decoder::decode
    runs until n_data_bits are out (lib/viterbi_decoder.cc:307-327), i.e. ntraceback calls past the end of the frame, on whatever its buffers hold
    there — zeros in a decoder that has not seen a longer frame.  Those zeros are taken as received bits (also where the puncturing pattern has
    erasures), the true path behind the scrambled pad does not agree with them, and at rate 3/4 (free distance 5) that can outweigh the frame's
    last data bits: a noise-free frame then fails its CRC.  Only at the rate-3/4 MCS and only when the pad is 6 or 10 bits long; never at rate
    1/2.  The HIP decoder reproduces each of these verdicts (tests/test_gpu_codec.py compares failing decodes bit for bit)."""
    rng = np.random.default_rng(1)
    failed = {m: [] for m in range(6)}
    for mcs in range(6):
        dbps = {0: 24, 1: 36, 2: 48, 3: 72, 4: 96, 5: 144}[mcs]
        for n in range(1, 360, 1 if mcs == 1 else 3):
            pdu = bytes([2]) + rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            sym, tags = oracle.stream_encode(mcs, 48, pdu, 1 + n % 127)
            ok, payload = oracle.stream_decode(mcs, 48, tags["pdu_len"], sym)
            if ok:
                assert payload == pdu
            else:
                bits = 16 + 8 * tags["pdu_len"] + 6
                failed[mcs].append(-(-bits // dbps) * dbps - bits)
    assert not failed[0] and not failed[2] and not failed[4]                       # rate 1/2: every clean frame decodes
    assert failed[1] and set(failed[1] + failed[3] + failed[5]) <= {6, 10}          # rate 3/4: some short-pad frames do not
