"""The measurement switches of INTEGRATION.md change HOW a launch is laid out (slices per frame, workgroups per CU, store pacing, frames per A1
launch, XCD count, the alternative kernels kept behind a switch), never WHAT it computes: channel estimate, map and records of the radar chain
must come out byte for byte as with the defaults - in map mode, detect-only mode and with the receive side in the time domain."""
import numpy as np
import pytest

SWITCHES = [{"JRC_WPF": "2"}, {"JRC_WPF": "8"}, {"JRC_WG_PER_CU": "1"}, {"JRC_RA_PACE": "0"}, {"JRC_RA_PACE": "60"}, {"JRC_RA_OFFERED_TBPS": "5.5"},
            {"JRC_NCUS": "96"}, {"JRC_XCDS": "4"}, {"JRC_XCDS": "1"}, {"JRC_CHANEST_CHUNK": "16"}, {"JRC_CHANEST_X1": "1"}, {"JRC_THREADS": "512"},
            {"JRC_DETECT_SLICES": "2"}, {"JRC_DETECT_SLICES": "5"}, {"JRC_DETECT_SLICES": "3", "JRC_CHANEST_U2": "1"}]


def _run(jrc, env, monkeypatch, cfg, with_map=True):
    import torch
    from jrc_amd import synth
    for k in list(env):
        monkeypatch.setenv(k, env[k])
    N, T, R, S, Ir, F = cfg
    sc = synth.Scenario(N, T, R, S, targets=[(12.0, 15.0, 0.0, 100.0), (31.0, -25.0, 0.0, 40.0)])
    frames = synth.make_frames(sc, F)
    P = T * R
    rb, ab = jrc.radar_axes(N, sc.fs, Ir, P, 16)
    c = jrc.Context(0)
    chain = jrc.RadarChain(N, T, R, S, sc.Npre, Ir, 16, rb, ab, 2.4, 14.4, 15.0, 0.0, max_frames=F, ctx=c)
    bufs = chain.alloc(F, "cuda:0")
    bufs["frames"].copy_(torch.from_numpy(frames.view(np.float32).reshape(bufs["frames"].shape)))
    torch.cuda.synchronize()
    chain.run(bufs, F)
    c.sync()
    out = [bufs["chanest"].cpu().numpy().tobytes(), bufs["map"].cpu().numpy().tobytes() if with_map else b"", bufs["results"].cpu().numpy().tobytes()]
    chain.set_write_map(False)
    chain.run(bufs, F)
    c.sync()
    out.append(bufs["results"].cpu().numpy().tobytes())
    chain.close()
    c.close()
    for k in env:
        monkeypatch.delenv(k)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(256, 4, 4, 16, 8, 40), (1024, 4, 4, 8, 4, 9), (128, 2, 2, 16, 4, 21)])
def test_radar_chain_does_not_depend_on_the_launch_switches(jrc, monkeypatch, cfg):
    want = _run(jrc, {}, monkeypatch, cfg)
    assert want[2] == want[3]                                    # detect-only records == map-mode records
    for env in SWITCHES:
        got = _run(jrc, env, monkeypatch, cfg)
        for name, a, b in zip(("chanest", "map", "records", "detect-only records"), want, got):
            assert a == b, (env, name)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"JRC_DETECT_SLICES": "2", "JRC_NCUS": "32"}, {"JRC_DETECT_SLICES": "3", "JRC_NCUS": "32"}, {"JRC_DETECT_SLICES": "2"}])
def test_detect_slices_beyond_one_resident_wave_with_an_uneven_last_slice(jrc, monkeypatch, env):
    """the sliced detect-only pipeline with more frames than resident workgroups and slices that pick DIFFERENT partial strides — with
    32 CUs x 2 workgroups, 136 frames in 2 slices are 72 frames (one full chunk of 64 + a tail of 8: stride 8) beside 64 frames (stride 1) on
    the other stream; the second slice's partial maxima used to start at 72 x 1, inside the first slice's 72 x 8 (corrupted peaks).  Slices now
    start at f0 x C.  Records byte for byte as with the defaults; the third case is the ADVICE's own (1032 frames on all CUs: 520 + 512)."""
    small = "JRC_NCUS" in env
    cfg = (256, 4, 4, 4, 8, 136 if small else 1032)
    want = _run(jrc, {}, monkeypatch, cfg, with_map=small)
    got = _run(jrc, env, monkeypatch, cfg, with_map=small)
    assert want[2] == want[3]
    for name, a, b in zip(("chanest", "map", "records", "detect-only records"), want, got):
        assert a == b, (env, name)


@pytest.mark.gpu
@pytest.mark.parametrize("spr", ["1", "2", "4"])
def test_time_domain_front_kernels_agree(jrc, monkeypatch, spr):
    """JRC_DEMOD_SPR selects the radix-4 A6+A7+A1 kernel (1 / 2 / 4 symbols per round) where the radix-16 one is the default: another
    factorisation of the same transform, so the estimate agrees to rounding, and the accumulation order over the symbols is the same"""
    import torch
    from jrc_amd import synth
    sc = synth.Scenario(256, 4, 2, 16, targets=[(12.0, 15.0, 0.0, 100.0)])
    F = 6
    frames = synth.make_frames(sc, F)
    n_items, N, cp = sc.Npre + sc.S, sc.N, sc.cp
    x = np.fft.ifft(np.fft.ifftshift(frames[:, sc.T:], axes=-1), axis=-1).astype(np.complex64)     # time-domain RX streams whose demodulation gives the RX ports
    td = np.concatenate([x[..., N - cp:], x], axis=-1).reshape(F, sc.R, n_items * (N + cp))
    outs = []
    for env in (None, spr):
        if env:
            monkeypatch.setenv("JRC_DEMOD_SPR", env)
        c = jrc.Context(0)
        tx = torch.from_numpy(np.ascontiguousarray(frames[:, :sc.T]).view(np.float32).reshape(F, sc.T, n_items, sc.N, 2)).to("cuda:0")
        rx = torch.from_numpy(np.ascontiguousarray(td).view(np.float32).reshape(F, sc.R, -1, 2)).to("cuda:0")
        H = torch.empty((F, sc.T * sc.R, sc.N, 2), device="cuda:0")
        c.check(c.lib.jrc_radar_chanest_td_dev(c.h, sc.N, sc.cp, sc.T, sc.R, sc.S, sc.Npre, n_items, rx.shape[2], 0, F, tx.data_ptr(), rx.data_ptr(), H.data_ptr(), None))
        c.sync()
        outs.append(H.cpu().numpy().view(np.complex64)[..., 0])
        c.close()
    err = np.abs(outs[0] - outs[1]).max() / np.abs(outs[0]).max()
    assert err < 2e-6, err
