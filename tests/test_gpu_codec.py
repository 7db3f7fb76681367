"""GPU tier: stream_encoder / stream_decoder on the device through the C ABI against the oracle
restatement (lib/stream_encoder_impl.cc, lib/utils.cc, lib/stream_decoder_impl.cc, lib/viterbi_decoder.cc).  Integer
work: symbols, decoded bytes and CRC verdicts are compared exactly, also under channel errors heavy enough to make the
windowed Viterbi decoder fail."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
N_DC = 48


def pdu(rng, n, ptype=2):
    return bytes([ptype]) + rng.integers(0, 256, n - 1, dtype=np.uint8).tobytes() if n else b""


@pytest.mark.parametrize("mcs", range(6))
@pytest.mark.parametrize("nbytes", [1, 2, 47, 333, 3096])
def test_encoder_is_bit_exact(jrc, ctx, mcs, nbytes):
    rng = np.random.default_rng(mcs * 7 + nbytes)
    enc = jrc.stream_encoder(mcs, N_DC, ctx=ctx)
    for turn in range(2):                          # the scrambler seed advances from PDU to PDU (:171-175)
        p = pdu(rng, nbytes)
        got, tags = enc.work(p)
        want, wtags = oracle.stream_encode(mcs, N_DC, p, 1 + turn)
        np.testing.assert_array_equal(got, want)
        assert tags == wtags


def test_encoder_edges(jrc, ctx):
    enc = jrc.stream_encoder(2, N_DC, ctx=ctx)
    assert enc.work(bytes(3097)) == (None, None)                      # too large: dropped (:139-143)
    got, tags = enc.work(b"")
    want, wtags = oracle.stream_encode(2, N_DC, b"", 1)
    np.testing.assert_array_equal(got, want)
    enc.d_scrambler = 127
    g1, _ = enc.work(b"\x02abc")
    np.testing.assert_array_equal(g1, oracle.stream_encode(2, N_DC, b"\x02abc", 127)[0])
    assert enc.d_scrambler == 1                                      # wraps (:172-175)
    with pytest.raises(ValueError):
        jrc.stream_encoder(6, N_DC, ctx=ctx)
    wide = jrc.stream_encoder(5, 200, ctx=ctx)                       # config-C sized symbol: 600 data bits per OFDM symbol
    p = pdu(np.random.default_rng(0), 500)
    np.testing.assert_array_equal(wide.work(p)[0], oracle.stream_encode(5, 200, p, 1)[0])


@pytest.mark.parametrize("mcs", range(6))
@pytest.mark.parametrize("nbytes", [5, 100, 1500])
def test_decoder_matches_oracle_clean_noisy_and_broken(jrc, ctx, mcs, nbytes):
    rng = np.random.default_rng(100 + mcs * 13 + nbytes)
    p = pdu(rng, nbytes)
    sym, tags = oracle.stream_encode(mcs, N_DC, p, 1 + nbytes % 127)
    dec = jrc.stream_decoder(N_DC, ctx=ctx)
    start = dict(mcs=mcs, data_bytes=tags["pdu_len"], packet_type=2, snr=20.0)
    assert dec.work(sym, start) == (True, p)
    bpsc = oracle.packet_params(mcs, N_DC, 8)["n_bpsc"]
    for scale in (1.0, 2.5, 6.0):                  # from "always decodes" to "Viterbi fails": the verdicts must agree bit for bit
        sigma = scale * {1: 0.25, 2: 0.12, 4: 0.05}[bpsc]
        noisy = sym + sigma * (rng.standard_normal(sym.size) + 1j * rng.standard_normal(sym.size)).astype(np.complex64)
        got = dec.work(noisy, start)
        want = oracle.stream_decode(mcs, N_DC, tags["pdu_len"], noisy)
        assert got == want
    assert 0.0 <= dec.per <= 100.0


def test_decoder_refuses_and_checks_arguments(jrc, ctx):
    dec = jrc.stream_decoder(N_DC, ctx=ctx)
    assert dec.work(np.zeros(60000, np.complex64), dict(mcs=0, data_bytes=3101)) == (None, None)      # :133-146
    with pytest.raises(jrc.JrcError):
        dec.work(np.zeros(10, np.complex64), dict(mcs=2, data_bytes=100))                            # fewer symbols than the frame needs


@pytest.mark.parametrize("fpw", [0, 1, 2])
def test_batched_device_round_trip_with_mixed_mcs(jrc, ctx, fpw, monkeypatch):
    """fpw: frames per wave in the decoder — 0 = chosen by batch size (one per wave here), 1 / 2 forced (two frames share a lane
    register above 8192 frames); an odd batch leaves the last wave of the two-frame variant half empty"""
    import torch
    if fpw:
        monkeypatch.setenv("JRC_DEC_FPW", str(fpw))
        ctx = jrc.Context(0)                        # the switch is read when a context is created
    rng = np.random.default_rng(9)
    F, stride_b = 97, 512
    lens = rng.integers(1, 400, F).astype(np.int32)
    lens[5] = 3200                                  # one oversized PDU in the batch: dropped, the rest unaffected
    psdu = np.zeros((F, 3328), np.uint8)
    for f in range(F):
        psdu[f, :lens[f]] = np.frombuffer(pdu(rng, int(lens[f])), np.uint8)
    scr = (1 + np.arange(F) % 127).astype(np.uint8)
    d_psdu, d_len, d_scr = torch.from_numpy(psdu).cuda(), torch.from_numpy(lens).cuda(), torch.from_numpy(scr).cuda()
    for mcs in (1, 2, 5):
        enc = jrc.stream_encoder(mcs, N_DC, ctx=ctx)
        max_sym = max(oracle.packet_params(mcs, N_DC, int(l) + 4)["n_ofdm_sym"] for l in lens if l + 4 <= 3100) * N_DC
        d_sym = torch.zeros((F, max_sym), dtype=torch.complex64, device="cuda")
        d_ns = torch.zeros(F, dtype=torch.int32, device="cuda")
        enc.encode_dev(d_psdu, psdu.shape[1], d_len, d_scr, d_sym, max_sym, d_ns, F)
        ctx.sync()
        ns = d_ns.cpu().numpy()
        assert ns[5] == 0
        sym = d_sym.cpu().numpy()
        for f in (0, 17, 96):
            want, _ = oracle.stream_encode(mcs, N_DC, psdu[f, :lens[f]].tobytes(), int(scr[f]))
            np.testing.assert_array_equal(sym[f, :ns[f]], want)
        # decode the whole batch, every third frame through a noisy channel
        noise = 0.3 * {1: 0.25, 2: 0.12, 4: 0.05}[oracle.packet_params(mcs, N_DC, 8)["n_bpsc"]]
        noisy = sym.copy()
        noisy[::3] += (noise * (rng.standard_normal(sym[::3].shape) + 1j * rng.standard_normal(sym[::3].shape))).astype(np.complex64)
        d_rx = torch.from_numpy(noisy).cuda()
        d_mcs = torch.full((F,), mcs, dtype=torch.int32, device="cuda")
        nb = (lens + 4).astype(np.int32)
        nb[5] = 3200 + 4                             # refused by the decoder as well
        d_nb = torch.from_numpy(nb).cuda()
        d_pl = torch.zeros((F, 3328), dtype=torch.uint8, device="cuda")
        d_st = torch.full((F,), -7, dtype=torch.int32, device="cuda")
        dec = jrc.stream_decoder(N_DC, ctx=ctx)
        dec.decode_dev(d_rx, max_sym, d_mcs, d_nb, d_pl, 3328, d_st, F)
        ctx.sync()
        st, pl = d_st.cpu().numpy(), d_pl.cpu().numpy()
        assert st[5] == -1
        for f in range(F):
            if f == 5:
                continue
            ok, payload = oracle.stream_decode(mcs, N_DC, int(nb[f]), noisy[f])
            assert st[f] == int(ok) and pl[f, :lens[f]].tobytes() == payload
        assert st[np.arange(F) != 5].sum() >= 0.8 * F             # light noise: most frames decode
