"""GPU tier: empty, ragged and boundary inputs of the per-block entry points, checked against the oracle's behaviour."""
import numpy as np
import pytest

import oracle
from conftest import crandn, rel_err

pytestmark = pytest.mark.gpu


def test_empty_and_sub_symbol_inputs(jrc, ctx):
    rng = np.random.default_rng(0)
    assert jrc.ofdm_cyclic_prefix_remover(64, 16, ctx=ctx).work(crandn(rng, 79)).shape == (0, 64)      # < one symbol
    assert jrc.ofdm_cyclic_prefix_remover(64, 16, ctx=ctx).work(np.zeros(0, np.complex64)).shape == (0, 64)
    assert jrc.fft_vcc(64, True, ctx=ctx).work(np.zeros((0, 64), np.complex64)).shape == (0, 64)          # batch 0
    assert jrc.ofdm_mod(np.zeros((0, 64), np.complex64), 64, 16, ctx=ctx).shape == (0, 80)
    t = jrc.matrix_transpose(8, 4, 2, ctx=ctx).work(np.zeros((0, 8), np.complex64))                        # empty packet
    assert t.shape == (8, 8) and not t.any()
    k, f, p, m = jrc.fft_peak_detect(1000, 1.0, -50.0, 10, ctx=ctx).work(crandn(rng, 20))                  # all samples protected
    assert k == -1 and np.isnan(f)
    k, f, p, m = jrc.fft_peak_detect(1000, 1.0, -50.0, 0, ctx=ctx).work(np.zeros(0, np.complex64))
    assert k == -1


def test_radar_zero_symbols_and_single_subcarrier(jrc, ctx):
    rng = np.random.default_rng(1)
    tx = [crandn(rng, 3, 8) for _ in range(2)]
    rx = [crandn(rng, 3, 8) for _ in range(1)]
    out = jrc.mimo_ofdm_radar(8, 2, 1, 0, 3, interp_factor=2, ctx=ctx).general_work(tx, rx)               # N_sym = 0: all zero rows
    assert out.shape == (2, 16) and not out.any()
    tx = [crandn(rng, 4, 1)]
    rx = [crandn(rng, 4, 1)]
    got = jrc.mimo_ofdm_radar(1, 1, 1, 3, 1, ctx=ctx).general_work(tx, rx)
    assert np.array_equal(got, oracle.Radar(1, 1, 1, 3, 1).work(tx, rx))


def test_estimator_partial_packet_and_tiny_maps(jrc, ctx):
    """ninput_items may be smaller than the bin axis (short packet): n_inputs, not len(range_bins), wraps the noise window"""
    rng = np.random.default_rng(2)
    rb, ab = jrc.radar_axes(64, 125e6, 8, 8, 16)
    m = crandn(rng, 300, 128, scale=0.05)
    m[250, 31] = 4.0
    est = jrc.range_angle_estimator(128, rb, ab, 2.4, 28.96, 15.0, 0.0, ctx=ctx)
    g, o = est.work(m), oracle.ra_estimate(m, rb, ab, 2.4, 28.96, 15.0, 0.0)
    for k in ("peak_range_idx", "peak_angle_idx", "angle_null_idx", "n_noise_samples", "published"):
        assert getattr(g, k) == getattr(o, k)
    assert g.noise_power == o.noise_power and g.snr_est == o.snr_est
    rb2, ab2 = np.linspace(0, 10, 4).astype(np.float32), np.linspace(-60, 60, 4).astype(np.float32)
    m2 = crandn(rng, 4, 4)
    g = jrc.range_angle_estimator(4, rb2, ab2, 5.0, 50.0, 0.0, 0.0, ctx=ctx).work(m2)
    o = oracle.ra_estimate(m2, rb2, ab2, 5.0, 50.0, 0.0, 0.0)
    assert (g.peak_range_idx, g.peak_angle_idx, g.n_noise_samples) == (o.peak_range_idx, o.peak_angle_idx, o.n_noise_samples)
    assert g.noise_power == o.noise_power
    with pytest.raises(ValueError):
        jrc.range_angle_estimator(128, rb[:100], ab, 2.4, 28.96, 15.0, 0.0, ctx=ctx).work(m)                  # more rows than range bins


def test_equalizer_and_precoder_degenerate_calls(jrc, ctx, ofdm64):
    o = ofdm64
    eq = jrc.mimo_ofdm_equalizer(0, 24e9, 125e6, 64, 16, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                                 o["ltf_64"], o["ltf_mapped_sc__ss_sym"], 4, ctx=ctx)
    r = eq.general_work(np.zeros((0, 64), np.complex64))
    assert r["consumed"] == 0 and len(r["out"]) == 0 and r["events"] == []
    r = eq.general_work(np.ones((3, 64), np.complex64), [(0, 0.0)], noutput_items=0)
    assert r["consumed"] == 0 and len(r["out"]) == 0
    pre = jrc.mimo_precoder(64, 4, 1, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"], o["l_stf_ltf_64"],
                            o["ltf_mapped_sc__ss_sym"], ctx=ctx)
    with pytest.raises(ValueError, match="packet type"):
        pre.work(np.zeros(48, np.complex64), 0, 7, 0)
    with pytest.raises(ValueError):
        jrc.mimo_precoder(64, 4, 1, [], o["pilot_subcarriers"], o["pilot_symbols"], o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], ctx=ctx)
    with pytest.raises(ValueError, match="Estimator"):
        jrc.mimo_ofdm_equalizer(5, 24e9, 125e6, 64, 16, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"],
                                o["ltf_64"], o["ltf_mapped_sc__ss_sym"], 4, ctx=ctx)


def test_largest_supported_transform_and_map(jrc, ctx):
    rng = np.random.default_rng(3)
    x = crandn(rng, 1, 16384)
    assert rel_err(jrc.fft_vcc(16384, False, None, True, ctx=ctx).work(x), oracle.fft_vcc(x, False, True)) < 2e-6
    m = crandn(rng, 8192, 256, scale=0.01)                                   # config-D sized map, peak in the last cell
    m[8191, 255] = 1.0
    rb, ab = jrc.radar_axes(1024, 125e6, 8, 16, 16)
    g = jrc.range_angle_estimator(256, rb, ab, 2.4, 14.36, 15.0, 0.0, ctx=ctx).work(m)
    assert (g.peak_range_idx, g.peak_angle_idx) == (8191, 255)


def test_entry_points_bind_their_context_from_any_thread(jrc, ctx):
    """GNU Radio calls every block from its own thread, and hipSetDevice is per thread: the device-resident entry points bind the
    context's GPU themselves.  A chain created in this thread runs from a fresh thread, and after a second context was created,
    with the same results bit for bit."""
    import threading
    import torch
    from jrc_amd import synth
    sc = synth.Scenario(64, 2, 2, 4, targets=[(9.0, 15.0, 0.0, 60.0)])
    F, Ir, Ia, P = 5, 4, 8, 4
    rb, ab = jrc.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    chain = jrc.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 29.0, max_frames=F, ctx=ctx)
    bufs = chain.alloc(F, "cuda:0")
    fr = synth.make_frames(sc, F)
    bufs["frames"].copy_(torch.from_numpy(fr.view(np.float32).reshape(bufs["frames"].shape)))
    torch.cuda.synchronize()
    chain.run(bufs, F)
    ctx.sync()
    want = (bufs["chanest"].clone(), bufs["map"].clone(), bufs["results"].clone())
    other = jrc.Context(jrc._device_count() - 1)           # the last GPU of the box (GPU 0 again on a one-GPU box)
    other.sync()
    err = []

    def worker():
        try:
            bufs["map"].zero_()
            torch.cuda.synchronize()
            chain.run(bufs, F)
            ctx.sync()
        except Exception as e:              # pragma: no cover
            err.append(e)
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert not err
    assert torch.equal(bufs["chanest"], want[0]) and torch.equal(bufs["map"], want[1]) and torch.equal(bufs["results"], want[2])
    other.close()


def test_zero_pad_with_row_strides(jrc, ctx):
    """jrc_zero_pad_strided_dev: one port of a [frame][port][samples] batch padded into a contiguous [frame][n_out] array — the burst copied untouched,
    the pads those of the contiguous entry point for the same seed (the generator is keyed by seed, burst and sample), strides shorter than a row
    and null buffers refused, no bursts / nothing to write = nothing done"""
    import ctypes as C
    import torch
    L = ctx.lib
    L.jrc_zero_pad_strided_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
    L.jrc_zero_pad_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    F, T, n, front, tail = 5, 3, 100, 7, 23
    x = torch.randn((F, T, n, 2), device="cuda:0")
    n_out = n + front + tail
    for t in range(T):
        out = torch.zeros((F, n_out + 9, 2), device="cuda:0")                               # rows further apart than they are long
        assert L.jrc_zero_pad_strided_dev(ctx.h, F, n, front, tail, 11, x.data_ptr() + 8 * t * n, T * n, out.data_ptr(), n_out + 9, None) == n_out
        ref = torch.zeros((F, n_out, 2), device="cuda:0")
        port = x[:, t].contiguous()
        assert L.jrc_zero_pad_dev(ctx.h, F, n, front, tail, 11, port.data_ptr(), ref.data_ptr(), None) == n_out
        ctx.sync()
        assert torch.equal(out[:, :n_out], ref) and not out[:, n_out:].any()
        assert torch.equal(out[:, front:front + n], x[:, t])
    with pytest.raises(ValueError, match="stride"):
        ctx.check(L.jrc_zero_pad_strided_dev(ctx.h, F, n, front, tail, 11, x.data_ptr(), n - 1, out.data_ptr(), n_out, None))
    with pytest.raises(ValueError, match="stride"):
        ctx.check(L.jrc_zero_pad_strided_dev(ctx.h, F, n, front, tail, 11, x.data_ptr(), n, out.data_ptr(), n_out - 1, None))
    with pytest.raises(ValueError, match="null"):
        ctx.check(L.jrc_zero_pad_strided_dev(ctx.h, F, n, front, tail, 11, None, n, out.data_ptr(), n_out, None))
    assert L.jrc_zero_pad_strided_dev(ctx.h, 0, n, front, tail, 11, None, n, None, n_out, None) == n_out
    assert L.jrc_zero_pad_strided_dev(ctx.h, F, 0, 0, 0, 11, None, 0, None, 0, None) == 0


def test_a_context_closed_first_destroys_the_objects_created_on_it(jrc):
    """jrc_destroy frees only the context's own twiddles / scratch / stream, so a context that goes first must take its chains, feeds
    and blocks with it (their device buffers used to leak silently): Context.close() closes its children, their own close() afterwards is a
    no-op, and the device memory they held is back."""
    import torch
    from jrc_amd import synth
    sc = synth.Scenario(256, 4, 4, 8, targets=[(10.0, 20.0, 0.0, 100.0)])
    P = sc.T * sc.R
    rb, ab = jrc.radar_axes(sc.N, sc.fs, 8, P, 16)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    c = jrc.Context(0)
    chain = jrc.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, 8, 16, rb, ab, 2.4, 14.4, 15.0, 0.0, max_frames=64, ctx=c)
    feed = jrc.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, 8, 16, rb, ab, 2.4, 14.4, 15.0, 0.0, ctx=c, n_slots=2, frames_per_slot=32, maps_per_slot=1)
    radar = jrc.mimo_ofdm_radar(sc.N, sc.T, sc.R, sc.S, sc.Npre, ctx=c)
    det = jrc.frame_detector(64, 16, 0.6, 10, 640, ctx=c)
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held > 32 << 20                                            # the feed's slots alone are 2 x 32 frames x 9 symbols x 8 streams x 2 KiB
    c.close()
    assert chain.h is None and feed.h is None and radar.h is None and det.h is None
    for o in (chain, feed, radar, det):
        o.close()                                                     # no-ops now
    assert free0 - torch.cuda.mem_get_info()[0] < held // 8, (free0, torch.cuda.mem_get_info()[0], held)
