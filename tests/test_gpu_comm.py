"""GPU tier: equalizer (C1), precoder (C2) and steering (C3) entry points against the oracle on the same inputs.
Tolerance: the north star's 1e-4 on ||a-b||_inf/||b||_inf; the kernels restate the reference's scalar complex
arithmetic (libgcc division, unfused products), so what remains is libm-vs-ocml sin/cos/atan2 rounding (1e-6)."""
import ctypes

import numpy as np
import pytest

import oracle
from conftest import crandn, rel_err
from test_oracle_comm import qam16, qpsk, through_channel

pytestmark = pytest.mark.gpu
NDP, DATA, LS, STA = 1, 2, 0, 1
TOL = 2e-5


def blocks(jrc, ctx, o, est=LS, T=4):
    dc, pc = o["data_subcarriers"], o["pilot_subcarriers"]
    ps, sw, ml, ltf = o["pilot_symbols"], o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], o["ltf_64"]
    gp = jrc.mimo_precoder(64, T, 1, dc, pc, ps, sw, ml, ctx=ctx)
    ge = jrc.mimo_ofdm_equalizer(est, 24e9, 125e6, 64, 16, dc, pc, ps, ltf, ml, T, ctx=ctx)
    op = oracle.Precoder(64, T, 1, dc, pc, ps, sw, ml)
    oe = oracle.Equalizer(est, 24e9, 125e6, 64, 16, dc, pc, ps, ltf, ml, T)
    return gp, ge, op, oe


def close(a, b, rel=1e-3):
    return a == b or abs(a - b) < rel * max(1.0, abs(b))        # a == b also covers +-inf (noise-free channels)


def same_events(ge, oe):
    assert len(ge) == len(oe)
    for a, b in zip(ge, oe):
        assert a["kind"] == b["kind"] and a["offset"] == b["offset"]
        if a["kind"] == 1:
            assert (a["data_bytes"], a["mcs"], a["packet_type"]) == (b["data_bytes"], b["mcs"], b["packet_type"])
            assert close(a["snr"], b["snr"]) and a["freq_offset"] == b["freq_offset"]
        else:
            assert close(a["snr_data"], b["snr_data"])
            assert rel_err(a["chan_mean"], b["chan_mean"]) < TOL


@pytest.mark.parametrize("T", [2, 4, 8])
def test_steering_matrices(jrc, ctx, T):
    rng = np.random.default_rng(T)
    h = crandn(rng, 64, T)
    h[3, 1:] = 0                                   # degenerate Householder case
    h[5, 0] = -abs(h[5, 0])                        # both signs of Re x0
    for phased in (False, True):
        Q = jrc.steering_from_channel(h, phased, ctx=ctx)
        ref = np.stack([oracle.steering_from_channel(h[i], phased) for i in range(64)])
        assert rel_err(Q, ref) < 1e-6
    Q = jrc.steering_from_channel(h, False, ctx=ctx)
    for i in range(64):
        assert np.allclose(Q[i].conj().T @ Q[i], np.eye(T), atol=5e-6)


@pytest.mark.parametrize("ptype,steer", [(NDP, "dft"), (DATA, "dft"), (DATA, "mean"), (DATA, "sc"), (DATA, "radar")])
def test_precoder_work(jrc, ctx, ofdm64, ptype, steer):
    rng = np.random.default_rng(7)
    gp, _, op, _ = blocks(jrc, ctx, ofdm64)
    nbytes, mcs = 77, 2
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    assert jrc.n_ofdm_sym(mcs, 48, nbytes) == ns and gp.calculate_output_stream_length(ns * 48) == ns + 9
    s = qpsk(rng, ns * 48)
    kw = {}
    if steer in ("mean", "radar"):
        kw = dict(steer_mode=1, Q_mean=oracle.steering_from_channel(crandn(rng, 4)))
    if steer == "sc":
        kw = dict(steer_mode=2, Q_sc=np.stack([oracle.steering_from_channel(crandn(rng, 4)) for _ in range(64)]))
    if steer == "radar":
        kw["radar_streams"] = qpsk(rng, 3 * ns * 64).reshape(3, ns, 64)
    got = gp.work(s, mcs, ptype, nbytes, **kw)
    ref = op.work(s, mcs, ptype, nbytes, **kw)
    assert got.shape == ref.shape and rel_err(got, ref) < 1e-6
    assert np.array_equal(got[:, :5], ref[:, :5])          # sync words + SIG field are exact
    with pytest.raises(RuntimeError, match="MIMO PRECODER"):
        gp.work(s, mcs, ptype, nbytes + 200, **kw)


@pytest.mark.parametrize("est", [LS, STA])
@pytest.mark.parametrize("ptype", [NDP, DATA])
@pytest.mark.parametrize("mcs", [0, 2, 3, 4, 5])
def test_equalizer_whole_frame(jrc, ctx, ofdm64, est, ptype, mcs):
    """every MCS under both estimators; with 16-QAM (mcs 4, 5) the STA estimator takes its decisions from gr-digital's
    constellation_16qam (lib/mimo_ofdm_equalizer_impl.cc:127, :505-521, :563-570; table recollected, parity unpinned)"""
    rng = np.random.default_rng(est * 10 + ptype + mcs)
    gp, ge, op, oe = blocks(jrc, ctx, ofdm64, est)
    nbytes = 45
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    s = qam16(rng, ns * 48) if mcs >= 4 else qpsk(rng, ns * 48) if mcs >= 2 else (rng.integers(0, 2, ns * 48) * 2 - 1).astype(np.complex64)
    y = through_channel(op.work(s, mcs, ptype, nbytes), crandn(rng, 4), 2e-3, rng)
    y = np.concatenate([crandn(rng, 2, 64), y, crandn(rng, 3, 64)])        # junk before the tag and after the frame
    g = ge.general_work(y, [(2, 0.013)])
    o = oe.general_work(y, [(2, 0.013)])
    assert g["consumed"] == o["consumed"] == len(y) and g["out"].shape == o["out"].shape == (ns, 48)
    assert rel_err(g["out"], o["out"]) < TOL
    same_events(g["events"], o["events"])
    if ptype == NDP:
        assert rel_err(g["chan_est"], o["chan_est"]) < TOL
    else:
        assert g["chan_est"] is None and o["chan_est"] is None


@pytest.mark.parametrize("est", [STA, LS])
@pytest.mark.parametrize("ptype", [DATA, NDP])
def test_equalizer_frame_cut_at_every_symbol(jrc, ctx, ofdm64, est, ptype):
    """a frame handed over in two calls, cut behind every one of its symbols in turn, and symbol by symbol: the same output as in one call, to the
    bit.  Round 5: the MIMO-LTF store lives in LDS (on arrays that are dead between the SIG field and the data symbols) and only travels through
    HBM when a call ends between two MIMO-LTF symbols — the cuts behind symbols 3 .. 5 are exactly that."""
    rng = np.random.default_rng(11)
    gp, ge, op, oe = blocks(jrc, ctx, ofdm64, est)
    nbytes, mcs = 60, 2
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    y = through_channel(op.work(qpsk(rng, ns * 48), mcs, ptype, nbytes), crandn(rng, 4), 1e-3, rng)
    whole = ge.general_work(y, [(0, 0.01)])
    assert rel_err(whole["out"], oe.general_work(y, [(0, 0.01)])["out"]) < TOL
    for cuts in [[k] for k in range(1, len(y))] + [list(range(1, len(y)))]:
        outs, pos, ce = [], 0, None
        for end in cuts + [len(y)]:
            r = ge.general_work(y[pos:end], [(0, 0.01)] if pos == 0 else [])
            assert r["consumed"] == end - pos
            outs.append(r["out"])
            ce = r["chan_est"] if r["chan_est"] is not None else ce
            pos = end
        assert np.array_equal(np.concatenate(outs), whole["out"]), cuts
        if ptype == NDP:
            assert np.array_equal(ce, whole["chan_est"]), cuts


def test_equalizer_split_calls_keep_state(jrc, ctx, ofdm64):
    """work() may see a frame in pieces; state (incl. the MIMO-LTF store the reference loses) persists"""
    rng = np.random.default_rng(5)
    gp, ge, op, oe = blocks(jrc, ctx, ofdm64, STA)
    nbytes, mcs = 120, 2
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    s = qpsk(rng, ns * 48)
    y = through_channel(op.work(s, mcs, DATA, nbytes), crandn(rng, 4), 1e-3, rng)
    whole = oe.general_work(y, [(0, -0.02)])
    outs, pos = [], 0
    for chunk in (1, 4, 3, 2, 100):
        part = y[pos:pos + chunk]
        if len(part) == 0:
            break
        r = ge.general_work(part, [(0, -0.02)] if pos == 0 else [])
        assert r["consumed"] == len(part)
        outs.append(r["out"])
        pos += chunk
    got = np.concatenate(outs)
    assert got.shape == whole["out"].shape and rel_err(got, whole["out"]) < TOL


def test_equalizer_limited_output_space_and_sig_failure(jrc, ctx, ofdm64):
    rng = np.random.default_rng(6)
    gp, ge, op, oe = blocks(jrc, ctx, ofdm64)
    ns = oracle.n_ofdm_sym(2, 48, 60)
    s = qpsk(rng, ns * 48)
    y = through_channel(op.work(s, 2, NDP, 60), crandn(rng, 4))
    g = ge.general_work(y, [(0, 0.0)], noutput_items=3)
    o = oe.general_work(y, [(0, 0.0)], noutput_items=3)
    assert g["consumed"] == o["consumed"] and g["out"].shape == o["out"].shape == (3, 48)
    g2 = ge.general_work(y[g["consumed"]:])
    o2 = oe.general_work(y[o["consumed"]:])
    assert rel_err(np.concatenate([g["out"], g2["out"]]), np.concatenate([o["out"], o2["out"]])) < TOL
    bad = y.copy()
    bad[2] = crandn(rng, 64)                                   # destroy the SIG symbol
    g = ge.general_work(bad, [(0, 0.0)])
    o = oe.general_work(bad, [(0, 0.0)])
    assert g["consumed"] == o["consumed"] and len(g["out"]) == len(o["out"])
    same_events(g["events"], o["events"])


def test_equalizer_batched_frames_on_device(jrc, ctx, ofdm64):
    """config-C style: several RX lanes x frames, one workgroup each, device resident"""
    import torch
    rng = np.random.default_rng(8)
    o = ofdm64
    S = 6
    gp, _, op, _ = blocks(jrc, ctx, o)
    ge = jrc.mimo_ofdm_equalizer(LS, 24e9, 125e6, 64, 16, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                                 o["ltf_64"], o["ltf_mapped_sc__ss_sym"], 4, n_streams=S, ctx=ctx)
    nbytes, mcs = 50, 2
    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
    frames, refs, phases = [], [], []
    for i in range(S):
        s = qpsk(rng, ns * 48)
        y = through_channel(op.work(s, mcs, DATA if i % 2 else NDP, nbytes), crandn(rng, 4), 1e-3, rng)
        ph = 0.01 * i
        oe = oracle.Equalizer(LS, 24e9, 125e6, 64, 16, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                              o["ltf_64"], o["ltf_mapped_sc__ss_sym"], 4)
        refs.append(oe.general_work(y, [(0, ph)]))
        frames.append(y)
        phases.append(ph)
    x = np.stack(frames)
    d_in = torch.from_numpy(x.view(np.float32).reshape(S, x.shape[1], 64, 2)).to("cuda:0")
    d_ph = torch.tensor(phases, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    out, n_out, ev = ge.frames_dev(d_in, d_ph, x.shape[1], ns)
    ctx.sync()
    out = out.cpu().numpy().view(np.complex64)[..., 0]
    assert n_out.cpu().tolist() == [ns] * S
    evb = ev.cpu().numpy()
    for i in range(S):
        assert rel_err(out[i], refs[i]["out"]) < TOL
        e0 = jrc.EqEvent.from_buffer_copy(evb[i, 0].tobytes())
        e1 = jrc.EqEvent.from_buffer_copy(evb[i, 1].tobytes())
        assert (e0.kind, e1.kind) == (1, 2) and e0.packet_type == refs[i]["events"][0]["packet_type"]
        assert abs(e1.snr_data - refs[i]["events"][1]["snr_data"]) < 1e-3 * abs(refs[i]["events"][1]["snr_data"])


def config_c_tables(N=256, T=4, seed=0):
    """documented generalisation of the 64-carrier tables to N = 256 (the reference has none): +-1 LTF with
    guard bands and DC nulled, the reference's P_ltf Hadamard mapping, pilots every 32 carriers with the 802.11 polarity row"""
    from jrc_amd import synth
    rng = np.random.default_rng(seed)
    guard = N // 16
    act = [c for c in range(-N // 2 + guard, N // 2 - guard + 1) if c != 0]
    pilots = [c for c in act if c % 32 == 16][:8]
    data = [c for c in act if c not in pilots]
    ltf = np.zeros(N, np.complex64)
    ltf[np.array(act) + N // 2] = rng.choice([-1.0, 1.0], len(act))
    mapped = np.stack([(synth.hadamard(T) * ltf[sc]).reshape(-1) for sc in range(N)]).astype(np.complex64)
    pil = np.array([[1, 1, 1, -1, 1, 1, 1, -1], [-1, -1, -1, 1, -1, -1, -1, 1], [1, 1, 1, -1, 1, 1, 1, -1]], np.complex64)[:, :len(pilots)]
    sync = np.stack([ltf, ltf, ltf, ltf])
    return data, pilots, pil, ltf, mapped, sync


@pytest.mark.parametrize("est,ptype,steer", [(LS, DATA, "dft"), (LS, DATA, "sc"), (LS, NDP, "dft"), (STA, DATA, "mean")])
def test_config_c_equalizer_and_precoder_256_subcarriers_64_symbols(jrc, ctx, est, ptype, steer):
    """BASELINE config C: 4 TX, 256 subcarriers, 64 data symbols, LS channel estimate + per-subcarrier solve; the north
    star's tolerance 1e-4 on ||a-b||_inf/||b||_inf against the oracle (the restated Eigen/libstdc++ arithmetic)"""
    N, cp, T, S = 256, 64, 4, 64
    rng = np.random.default_rng(11)
    data, pilots, pil, ltf, mapped, sync = config_c_tables(N, T)
    nd = len(data)
    mcs = 2
    nbytes = (S * nd - 22) // 8
    assert oracle.n_ofdm_sym(mcs, nd, nbytes) == S
    s = qpsk(rng, S * nd)
    h = crandn(rng, T)
    gp = jrc.mimo_precoder(N, T, 1, data, pilots, pil, sync, mapped, ctx=ctx)
    op = oracle.Precoder(N, T, 1, data, pilots, pil, sync, mapped)
    kw = {}
    if steer == "sc":
        hs = crandn(rng, N, T)
        kw = dict(steer_mode=2, Q_sc=jrc.steering_from_channel(hs, ctx=ctx))
        assert rel_err(kw["Q_sc"], np.stack([oracle.steering_from_channel(hs[i]) for i in range(N)])) < 1e-6
    elif steer == "mean":
        kw = dict(steer_mode=1, Q_mean=oracle.steering_from_channel(h))
    tx_g = gp.work(s, mcs, ptype, nbytes, **kw)
    tx_o = op.work(s, mcs, ptype, nbytes, **kw)
    assert tx_g.shape == (T, S + 9, N) and rel_err(tx_g, tx_o) < 1e-6
    y = np.tensordot(h, tx_o, axes=(0, 0))
    y = np.concatenate([y[3:4], y[3:]], axis=0)
    y = (y + 2e-3 * (rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape))).astype(np.complex64)
    ge = jrc.mimo_ofdm_equalizer(est, 24e9, 125e6, N, cp, data, pilots, pil, ltf, mapped, T, ctx=ctx)
    oe = oracle.Equalizer(est, 24e9, 125e6, N, cp, data, pilots, pil, ltf, mapped, T)
    g = ge.general_work(y, [(0, 0.004)])
    o = oe.general_work(y, [(0, 0.004)])
    assert g["out"].shape == o["out"].shape == (S, nd) and g["consumed"] == o["consumed"] == len(y)
    assert rel_err(g["out"], o["out"]) < 1e-4
    same_events(g["events"], o["events"])
    if ptype == NDP:
        assert rel_err(g["chan_est"], o["chan_est"]) < 1e-4
    if steer != "sc" and est == LS:
        ref = s.reshape(S, nd)                                      # and the frame really decodes: QPSK decisions match
        assert np.mean((np.sign(g["out"].real) == np.sign(ref.real)) & (np.sign(g["out"].imag) == np.sign(ref.imag))) > 0.999


@pytest.mark.parametrize("N,est,ptype", [(256, LS, DATA), (256, LS, NDP), (256, STA, DATA), (128, LS, DATA), (512, LS, DATA)])
def test_equalizer_batched_launch_with_a_workgroup_per_cu_or_more(jrc, ctx, N, est, ptype):
    """a batch of at least one stream per CU runs the equalizer in its narrow geometry (a quarter of the lanes, several subcarriers per
    lane — `launch_equalizer` in comm.hip): every stream against the oracle's general_work on the same frame"""
    import torch
    cp, T, S = N // 4, 4, 12
    rng = np.random.default_rng(21)
    data, pilots, pil, ltf, mapped, sync = config_c_tables(N, T)
    nd = len(data)
    mcs = 2
    nbytes = (S * nd - 22) // 8
    assert oracle.n_ofdm_sym(mcs, nd, nbytes) == S
    op = oracle.Precoder(N, T, 1, data, pilots, pil, sync, mapped)
    n_distinct, n_streams = 6, 320
    frames, refs, phases = [], [], []
    for i in range(n_distinct):
        tx = op.work(qpsk(rng, S * nd), mcs, ptype, nbytes)
        y = np.tensordot(crandn(rng, T), tx, axes=(0, 0))
        y = np.concatenate([y[3:4], y[3:]], axis=0)
        y = (y + 2e-3 * (rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape))).astype(np.complex64)
        ph = 0.002 * i
        oe = oracle.Equalizer(est, 24e9, 125e6, N, cp, data, pilots, pil, ltf, mapped, T)
        refs.append(oe.general_work(y, [(0, ph)]))
        frames.append(y)
        phases.append(ph)
    n_sym = frames[0].shape[0]
    x = np.stack([frames[i % n_distinct] for i in range(n_streams)])
    ge = jrc.mimo_ofdm_equalizer(est, 24e9, 125e6, N, cp, data, pilots, pil, ltf, mapped, T, n_streams=n_streams, ctx=ctx)
    d_in = torch.from_numpy(x.view(np.float32).reshape(n_streams, n_sym, N, 2)).to("cuda:0")
    d_ph = torch.tensor([phases[i % n_distinct] for i in range(n_streams)], dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    out, n_out, ev = ge.frames_dev(d_in, d_ph, n_sym, S)
    ctx.sync()
    out = out.cpu().numpy().view(np.complex64)[..., 0]
    assert n_out.cpu().tolist() == [S] * n_streams
    evb = ev.cpu().numpy()
    for i in range(n_streams):
        r = refs[i % n_distinct]
        assert r["out"].shape == (S, nd)
        assert rel_err(out[i], r["out"]) < 1e-4
        e0 = jrc.EqEvent.from_buffer_copy(evb[i, 0].tobytes())
        e1 = jrc.EqEvent.from_buffer_copy(evb[i, 1].tobytes())
        assert (e0.kind, e1.kind) == (1, 2)
        assert (e0.data_bytes, e0.mcs, e0.packet_type) == tuple(r["events"][0][k] for k in ("data_bytes", "mcs", "packet_type"))
        assert close(e1.snr_data, r["events"][1]["snr_data"])


@pytest.mark.parametrize("ptype,steer", [(NDP, "dft"), (DATA, "dft"), (DATA, "mean"), (DATA, "sc"), (DATA, "radar")])
def test_precoder_batched_device_resident(jrc, ctx, ptype, steer):
    """jrc_precoder_frames_dev: a batch of config-C packets (4 TX, 256 subcarriers, 64 data symbols) in one launch, symbols in and
    port buffers out in HBM, equals the per-packet work() bit for bit and the oracle within 1e-6; format change between calls"""
    import torch
    N, T, S, F = 256, 4, 64, 5
    rng = np.random.default_rng(21)
    data, pilots, pil, ltf, mapped, sync = config_c_tables(N, T)
    nd, mcs = len(data), 2
    nbytes = (S * nd - 22) // 8
    gp = jrc.mimo_precoder(N, T, 1, data, pilots, pil, sync, mapped, ctx=ctx)
    op = oracle.Precoder(N, T, 1, data, pilots, pil, sync, mapped)
    s = np.stack([qpsk(rng, S * nd) for _ in range(F)])
    kw, dkw = {}, {}
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.float32).reshape(a.shape + (2,))).cuda()
    if steer in ("mean", "radar"):
        Q = oracle.steering_from_channel(crandn(rng, T))
        kw = dict(steer_mode=1, Q_mean=Q)
        dkw = dict(steer_mode=1, d_Q_mean=up(np.ascontiguousarray(Q.T).reshape(-1)))                 # column-major
    if steer == "sc":
        Q = np.stack([oracle.steering_from_channel(crandn(rng, T)) for _ in range(N)])
        kw = dict(steer_mode=2, Q_sc=Q)
        dkw = dict(steer_mode=2, d_Q_sc=up(np.ascontiguousarray(np.transpose(Q, (0, 2, 1))).reshape(N, -1)))
    rs = None
    if steer == "radar":
        rs = np.stack([qpsk(rng, (T - 1) * S * N).reshape(T - 1, S, N) for _ in range(F)])
        dkw["d_radar_streams"] = up(rs)
    out = gp.frames_dev(up(s), mcs, ptype, nbytes, **dkw)
    ctx.sync()
    got = out.cpu().numpy().view(np.complex64)[..., 0]
    assert got.shape == (F, T, S + 9, N)
    for f in range(F):
        kwf = dict(kw)
        if rs is not None:
            kwf["radar_streams"] = rs[f]
        assert np.array_equal(got[f], gp.work(s[f], mcs, ptype, nbytes, **kwf))
        if f in (0, F - 1):
            assert rel_err(got[f], op.work(s[f], mcs, ptype, nbytes, **kwf)) < 1e-6
    # another format on the same block: the cached SIG field is rebuilt
    nb2 = nbytes - 3 * nd // 8 - 40
    S2 = oracle.n_ofdm_sym(mcs, nd, nb2)
    s2 = np.stack([qpsk(rng, S2 * nd) for _ in range(2)])
    dkw.pop("d_radar_streams", None)
    out2 = gp.frames_dev(up(s2), mcs, ptype, nb2, **dkw)
    ctx.sync()                                                       # the library's stream, not torch's
    got2 = out2.cpu().numpy().view(np.complex64)[..., 0]
    assert np.array_equal(got2[1], gp.work(s2[1], mcs, ptype, nb2, **kw))
    with pytest.raises(RuntimeError, match="MIMO PRECODER"):
        gp.frames_dev(up(s2), mcs, ptype, nbytes, **dkw)


@pytest.mark.parametrize("n_err", [1, 2, 3, 5, 8])
def test_sig_field_with_bit_errors_resolves_as_the_windowed_decoder_does(jrc, ctx, ofdm64, n_err):
    """SIG symbols with flipped BPSK decisions through sig_viterbi_wave (the reference's windowed decoder on one
    wavefront, comm.hip) and through the oracle (lane-by-lane restatement of lib/viterbi_decoder.cc, itself equal to the second source,
    tests/test_second_source.py): same header fields, same success / failure, same number of output symbols - for error counts the
    code corrects and for counts it cannot, where a maximum-likelihood decoder would answer differently"""
    from test_second_source import sig_frame
    rng = np.random.default_rng(50 + n_err)
    for trial in range(12):
        mcs, ptype, length = int(rng.integers(0, 6)), int(rng.integers(1, 3)), int(rng.integers(1, 300))
        flips = rng.choice(48, n_err, replace=False)
        y = sig_frame(ofdm64, mcs, ptype, length, flips, rng)
        _, ge, _, oe = blocks(jrc, ctx, ofdm64)
        rg, ro = ge.general_work(y, [(0, 0.0)]), oe.general_work(y, [(0, 0.0)])
        assert rg["consumed"] == ro["consumed"] and rg["out"].shape == ro["out"].shape
        same_events(rg["events"], ro["events"])
        if ro["out"].size:
            assert rel_err(rg["out"], ro["out"]) < TOL


@pytest.mark.parametrize("n_err", [0, 1, 4])
def test_sig_codeword_shortcut_agrees_with_the_trellis(jrc, ofdm64, n_err, monkeypatch):
    """round 4: a SIG word that is a codeword skips the trellis (comm.hip sig_viterbi_wave); JRC_EQ_SIG_FULL=1 always runs it.  Same frames,
    both ways: events, consumed counts and equalised symbols byte for byte — clean fields take the shortcut, corrupted ones cannot."""
    from test_second_source import sig_frame
    outs = []
    for full in (False, True):
        if full:
            monkeypatch.setenv("JRC_EQ_SIG_FULL", "1")
        c = jrc.Context(0)
        rng = np.random.default_rng(90 + n_err)
        res = []
        for trial in range(10):
            mcs, ptype, length = int(rng.integers(0, 6)), int(rng.integers(1, 3)), int(rng.integers(1, 300))
            flips = rng.choice(48, n_err, replace=False)
            y = sig_frame(ofdm64, mcs, ptype, length, flips, rng)
            o = ofdm64
            ge = jrc.mimo_ofdm_equalizer(LS, 24e9, 125e6, 64, 16, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"], o["ltf_64"],
                                         o["ltf_mapped_sc__ss_sym"], 4, ctx=c)
            r = ge.general_work(y, [(0, 0.0)])
            res.append((r["consumed"], r["out"].tobytes(), [(e["kind"], e["offset"], e.get("data_bytes"), e.get("mcs"), e.get("packet_type")) for e in r["events"]]))
            if n_err == 0:
                assert r["events"] and r["events"][0]["data_bytes"] == length and r["events"][0]["mcs"] == mcs
            ge.close()
        outs.append(res)
        c.close()
    assert outs[0] == outs[1]
