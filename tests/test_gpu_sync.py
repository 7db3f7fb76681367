"""GPU tier: the sync front-end on the device (moving_avg, frame_detector, frame_sync and the stock
metric blocks in front of them) through the C ABI against the oracle restatement, call for call: items consumed /
produced, tags and decisions exact, samples within the north star's 1e-4."""
import numpy as np
import pytest

import oracle
from conftest import rel_err
from metric_truth import assert_metric_parity
from test_oracle_sync import CP, N, make_stream

pytestmark = pytest.mark.gpu
TOL = 1e-4
SYNC_LEN = 4 * (N + CP)


def capture(ofdm64, seed, cfo=0.01, n_frames=1, gap=2500):
    rng = np.random.default_rng(seed)
    parts, meta = [], []
    for k in range(n_frames):
        payload = bytes([2]) + rng.integers(0, 256, 60 + 30 * k, dtype=np.uint8).tobytes()
        x, tags, flen = make_stream(ofdm64, payload, 2, rng, lead=600 + 37 * k, tail=gap, cfo=cfo)
        parts.append(x)
        meta.append((payload, tags))
    return np.concatenate(parts), meta


@pytest.mark.parametrize("length,scale,n,max_iter", [(32, 1.0, 3000, 16000), (48, 1 / 1.5, 500, 16000), (1, 2.0, 10, 4), (5, 1.0, 100, 60)])
def test_moving_avg(jrc, ctx, length, scale, n, max_iter):
    rng = np.random.default_rng(length)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    blk = jrc.moving_avg(length, scale, max_iter, ctx=ctx)
    got = blk.work(x)
    want = oracle.moving_avg(x, length, scale, max_iter)
    assert got.shape == want.shape and rel_err(got, want) < TOL
    got2 = blk.work(x[len(got):])                                   # the next call continues with the history of the first
    want2 = oracle.moving_avg(x[len(got):], length, scale, max_iter, history=np.concatenate([np.zeros(length - 1), x])[len(got):len(got) + length - 1])
    assert got2.shape == want2.shape and (got2.size == 0 or rel_err(got2, want2) < TOL)


def test_metrics_match_the_stock_blocks(jrc, ctx, ofdm64):
    x, _ = capture(ofdm64, 1)
    gxd, gia, gic = jrc.sync_metrics(x, 16, 32, 48, 1 / 1.5, ctx=ctx)
    oxd, oia, oic = oracle.sync_metrics(x, 16, 32, 48, 1 / 1.5)
    assert np.array_equal(gxd, oxd)
    assert rel_err(gia, oia) < TOL
    # the normalised metric against its float64 definition (tests/metric_truth.py): device within 1e-4 and no further from it than the
    # oracle's running sums; device against oracle within 1e-4 + the running sums' own distance from the definition
    assert_metric_parity(gic, oic, x, x, 16, 32, 48, 1 / 1.5, live=48, tol=TOL)


def long_stream(ofdm64, seed, n, noise, loud_every=0, loud=100.0):
    rng = np.random.default_rng(seed)
    parts, tot, k = [], 0, 0
    while tot < n:
        payload = bytes([2]) + rng.integers(0, 256, int(rng.integers(40, 400)), dtype=np.uint8).tobytes()
        x, _, _ = make_stream(ofdm64, payload, 2, rng, lead=int(rng.integers(500, 3000)), tail=int(rng.integers(2500, 20000)),
                              cfo=float(rng.uniform(-0.01, 0.01)), noise=noise)
        if loud_every and k % loud_every == 0:
            x = (x * np.float32(loud)).astype(np.complex64)
        parts.append(x)
        tot += x.size
        k += 1
    return np.concatenate(parts)[:n]


def stock_metrics_at_max_iter(x, delay, window, pwindow, pscale, max_iter=16000):
    """the oracle's metric blocks with their running sums started afresh every max_iter outputs — blocks_moving_average_ff at the .grc's own
    max_iter 16000 (mimo_ofdm_jrc_comm_sim.grc:605) when the scheduler always hands it at least that many items"""
    ia, ic = np.empty(x.size, np.complex64), np.empty(x.size, np.float32)
    hist = delay + window + pwindow
    for i0 in range(0, x.size, max_iter):
        h = min(i0, hist)
        a = oracle.sync_metrics(x[i0 - h:i0 + max_iter], delay, window, pwindow, pscale)
        ia[i0:i0 + max_iter], ic[i0:i0 + max_iter] = a[1][h:], a[2][h:]
    return ia, ic


def decisions(det, xd, ia, ic):
    seg, tags = det.run(xd, np.ascontiguousarray(ia, np.complex64), np.ascontiguousarray(ic, np.float32))
    return seg.size, [t[0] for t in tags]


@pytest.mark.parametrize("scene", ["burst_40dB_over_the_noise", "frames_40dB_apart"])
def test_metric_and_decisions_on_a_million_samples_behind_a_40dB_power_step(jrc, ctx, ofdm64, scene):
    """10^6 samples, ~57 frames.  The stock moving averages carry running float sums; behind a power step they keep the louder part's round-off.
    Scene 1 — every burst 40 dB over the noise that follows it: the oracle's one-sum-per-capture metric is 1e-1 away from the float64 definition
    by the end and the .grc's max_iter 16000 version 4e-2, the device (each window summed afresh) stays within 1e-4 — and the detector's
    decisions (copied samples, frame_start tag offsets) are the same from all four metrics.
    Scene 2 — every 7th frame 40 dB louder than the others (70 dB over the noise, more than a float running sum holds): the running sums fake
    peaks in the noise behind a loud frame (false frame_start tags, a different count with one sum than with max_iter 16000: the reference's
    decisions there hang on where the scheduler cut the stream); the device's decisions are those of the float64 definition."""
    from metric_truth import metric_errors, metric_truth
    d, w, pw, ps = N // 4, N // 2, int(1.5 * (N // 2)), 1 / 1.5
    x = long_stream(ofdm64, 5, 1_000_000, noise=0.004 if scene.startswith("burst") else 0.02, loud_every=0 if scene.startswith("burst") else 7)
    gxd, gia, gic = jrc.sync_metrics(x, d, w, pw, ps, ctx=ctx)
    oxd, oia, oic = oracle.sync_metrics(x, d, w, pw, ps)
    cia, cic = stock_metrics_at_max_iter(x, d, w, pw, ps)
    corr, _, truth = metric_truth(x, d, w, pw, ps)
    assert np.array_equal(gxd, oxd)
    e1 = metric_errors(gic, oic, x, x, d, w, pw, ps, live=pw)
    e2 = metric_errors(gic, cic, x, x, d, w, pw, ps, live=pw)
    assert e1["dev"] <= TOL and e1["dev"] <= e1["ora"] and e1["dev"] <= e2["ora"], (e1, e2)
    ign = 8 * (N + CP)
    mk = lambda: oracle.FrameDetector(N, CP, 0.6, 10, ign)
    want = decisions(mk(), oxd, corr, truth)                              # the definition's decisions
    got = decisions(jrc.frame_detector(N, CP, 0.6, 10, ign, ctx=ctx), gxd, gia, gic)
    assert got == want and len(want[1]) >= 50
    assert decisions(mk(), gxd, gia, gic) == want                       # (and the oracle's detector on the device's metric)
    one_sum, at_max_iter = decisions(mk(), oxd, oia, oic), decisions(mk(), oxd, cia, cic)
    if scene.startswith("burst"):
        assert e1["ora"] > 100 * TOL and e2["ora"] > 10 * TOL            # the drift is there ...
        assert one_sum == want and at_max_iter == want                   # ... and the decisions do not feel it
    else:
        assert len(one_sum[1]) > len(want[1]) and len(at_max_iter[1]) > len(want[1]) and one_sum != at_max_iter


@pytest.mark.parametrize("chunk", [1 << 30, 4096, 1000, 333])
def test_detector_call_for_call(jrc, ctx, ofdm64, chunk):
    x, meta = capture(ofdm64, 2, cfo=-0.015, n_frames=3)
    xd, ia, ic = oracle.sync_metrics(x, 16, 32, 48, 1 / 1.5)
    ignore_gap = (4 + 4) * (N + CP)
    g = jrc.frame_detector(N, CP, 0.6, 10, ignore_gap, ctx=ctx)
    o = oracle.FrameDetector(N, CP, 0.6, 10, ignore_gap)
    pos = 0
    n_tags = 0
    while pos < x.size:
        n = min(chunk, x.size - pos)
        nout = n if chunk > 1000 else max(1, n - 7)                  # a smaller output buffer than input now and then
        go, gc, gt = g.work(xd[pos:pos + n], ia[pos:pos + n], ic[pos:pos + n], nout)
        oo, oc, ot = o.work(xd[pos:pos + n], ia[pos:pos + n], ic[pos:pos + n], nout)
        assert (gc, go.size, len(gt)) == (oc, oo.size, len(ot))
        for a, b in zip(gt, ot):
            assert a[0] == b[0] and abs(a[1] - b[1]) < 1e-6
        if go.size:
            assert rel_err(go, oo) < TOL
        n_tags += len(gt)
        if gc == 0 and go.size == 0:
            break
        pos += gc
    assert n_tags == 3                                               # SEARCH -> COPY, then two re-detections inside COPY (:153-165)


@pytest.mark.parametrize("chunk", [8192, 1500, 200])
def test_frame_sync_call_for_call(jrc, ctx, ofdm64, chunk):
    x, meta = capture(ofdm64, 3, cfo=0.02, n_frames=2)
    xd, ia, ic = oracle.sync_metrics(x, 16, 32, 48, 1 / 1.5)
    seg, dtags = oracle.FrameDetector(N, CP, 0.6, 10, 8 * (N + CP)).run(xd, ia, ic)
    assert len(dtags) == 2
    delayed = np.concatenate([np.zeros(SYNC_LEN, np.complex64), seg])[:seg.size]
    g = jrc.frame_sync(N, CP, SYNC_LEN, ofdm64["l_ltf_fir"], ctx=ctx)
    o = oracle.FrameSync(N, CP, SYNC_LEN, ofdm64["l_ltf_fir"])
    pos, idle, tags_seen = 0, 0, []
    while pos < seg.size and idle < 3:
        m = min(chunk, seg.size - pos)
        nout = m if chunk != 1500 else m // 2 + 1
        go, gc, gt = g.work(seg[pos:pos + m], delayed[pos:pos + m], dtags, nout)
        oo, oc, ot = o.work(seg[pos:pos + m], delayed[pos:pos + m], dtags, nout)
        assert (gc, go.size, len(gt)) == (oc, oo.size, len(ot)), (pos, g.state)
        for a, b in zip(gt, ot):
            assert a[0] == b[0] and abs(a[1] - b[1]) < 1e-5
        if go.size:
            assert rel_err(go, oo) < TOL
        assert g.frame_start == o.frame_start and abs(g.freq_offset - o.freq_offset) < 1e-6
        tags_seen += gt
        idle = idle + 1 if (gc == 0 and go.size == 0) else 0
        pos += gc
    assert len(tags_seen) == 2


def test_front_end_delivers_decodable_frames(jrc, ctx, ofdm64):
    """capture -> metrics -> frame_detector -> frame_sync -> fft_vcc -> mimo_ofdm_equalizer -> stream_decoder, all on the device blocks"""
    x, meta = capture(ofdm64, 4, cfo=0.01, n_frames=2)
    xd, ia, ic = jrc.sync_metrics(x, 16, 32, 48, 1 / 1.5, ctx=ctx)
    seg, dtags = jrc.frame_detector(N, CP, 0.6, 10, 8 * (N + CP), ctx=ctx).run(xd, ia, ic)
    assert len(dtags) == 2
    delayed = np.concatenate([np.zeros(SYNC_LEN, np.complex64), seg])[:seg.size]
    out, otags = jrc.frame_sync(N, CP, SYNC_LEN, ofdm64["l_ltf_fir"], ctx=ctx).run(seg, delayed, dtags)
    assert len(otags) == 2
    sym_t = out[:(out.size // N) * N].reshape(-1, N)
    sym_f = jrc.fft_vcc(N, True, None, True, ctx=ctx).work(sym_t) / np.float32(np.sqrt(N))
    o = ofdm64
    eq = jrc.mimo_ofdm_equalizer(0, 24e9, 125e6, N, CP, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"], o["ltf_64"],
                                 o["ltf_mapped_sc__ss_sym"], 4, ctx=ctx)
    dec = jrc.stream_decoder(48, ctx=ctx)
    got = []
    tag_sym = [(t[0] // N, t[1]) for t in otags] + [(len(sym_f), 0.0)]
    for k in range(len(otags)):                                      # one equalizer call per tagged frame
        frame = sym_f[tag_sym[k][0]:tag_sym[k + 1][0]]
        r = eq.general_work(frame, [(0, tag_sym[k][1])])
        starts = [e for e in r["events"] if e["kind"] == 1]
        assert starts
        got.append(dec.work(r["out"], starts[0]))
    assert [g for g in got if g[0]] == [(True, m[0]) for m in meta]


def test_batched_front_end_equals_the_block_chain(jrc, ctx, ofdm64):
    """jrc_sync_frontend_dev on a capture with three frames against frame_detector + frame_sync run block by block"""
    import torch
    x, meta = capture(ofdm64, 6, cfo=-0.012, n_frames=3)
    fe = jrc.SyncFrontEnd(N, CP, 0.6, 10, 8 * (N + CP), SYNC_LEN, ofdm64["l_ltf_fir"], max_frames=8, max_symbols=80, ctx=ctx)
    d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()).cuda()
    fe.run(d_x, x.size)
    n, info = fe.results()
    assert n == 3
    rows = fe.frames.cpu().numpy().view(np.complex64)[..., 0].reshape(8, -1)
    # block chain on the same capture
    xd, ia, ic = oracle.sync_metrics(x, 16, 32, 48, 1 / 1.5)
    seg, dtags = oracle.FrameDetector(N, CP, 0.6, 10, 8 * (N + CP)).run(xd, ia, ic)
    delayed = np.concatenate([np.zeros(SYNC_LEN, np.complex64), seg])[:seg.size]
    out, otags = oracle.FrameSync(N, CP, SYNC_LEN, ofdm64["l_ltf_fir"]).run(seg, delayed, dtags)
    assert [t[0] for t in dtags] == list(np.cumsum([0] + [f.len for f in info[:-1]]))
    for k in range(3):
        assert abs(info[k].coarse_cfo - dtags[k][1]) < 1e-6 and abs(info[k].tag_value - otags[k][1]) < 1e-5
        lo = otags[k][0]
        hi = otags[k + 1][0] if k + 1 < 3 else out.size
        want = out[lo:hi]
        m = min(want.size, info[k].n_out)
        if k < 2:
            assert want.size == info[k].n_out                   # the block chain's RESET zero-fill included
        assert m > 20 * N and rel_err(rows[k][:m], want[:m]) < TOL
    # and the frames decode: FFT -> equalizer -> decoder
    o = ofdm64
    eq = jrc.mimo_ofdm_equalizer(0, 24e9, 125e6, N, CP, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"], o["ltf_64"],
                                 o["ltf_mapped_sc__ss_sym"], 4, ctx=ctx)
    dec = jrc.stream_decoder(48, ctx=ctx)
    for k in range(3):
        sym_f = jrc.fft_vcc(N, True, None, True, ctx=ctx).work(rows[k][:info[k].n_out].reshape(-1, N)) / np.float32(np.sqrt(N))
        r = eq.general_work(sym_f, [(0, info[k].tag_value)])
        st = [e for e in r["events"] if e["kind"] == 1]
        assert st and dec.work(r["out"], st[0]) == (True, meta[k][0])


@pytest.mark.parametrize("seed,density,min_peaks,gap", [(0, 0.02, 10, 640), (1, 0.3, 10, 640), (2, 0.9, 10, 100), (3, 0.5, 1, 0), (4, 0.97, 3, 50000),
                                                          (5, 0.6, 0, 200), (6, 0.999, 10, 640)])
def test_run_to_completion_scan_equals_the_call_by_call_detector(jrc, ctx, seed, density, min_peaks, gap):
    """the segment-parallel, wave-cooperative scan (word skipping, run skipping, one wave per 4096-sample segment starting at its first
    history-free sample) against the oracle's sample-by-sample state machine on random peak patterns, incl. bursts longer than
    MAX_SAMPLES, peak values at and above MAX_PEAK_VALUE, an ignore_gap longer than several segments, and captures without quiet stretches"""
    _scan_case(jrc, ctx, seed, density, min_peaks, gap, 4096, 3000)


@pytest.mark.parametrize("seed,density,min_peaks,gap,max_frames,max_quiet", [(10, 0.5, 10, 640, 7, 3000), (11, 0.9, 10, 640, 1, 9000), (12, 0.8, 5, 300, 23, 20000),
                                                                             (13, 0.7, 10, 640, 4096, 12), (14, 0.9, 10, 640, 4096, 60000)])
def test_scan_with_a_short_frame_list_and_other_spacings(jrc, ctx, seed, density, min_peaks, gap, max_frames, max_quiet):
    """the caller's list ends before the capture does (the last listed frame's length runs to the detection that did not fit); bursts
    12 samples apart (no quiet stretch anywhere: one wave walks the capture) and up to 60000 apart (most segments empty)"""
    _scan_case(jrc, ctx, seed, density, min_peaks, gap, max_frames, max_quiet)


def _scan_case(jrc, ctx, seed, density, min_peaks, gap, max_frames, max_quiet):
    import ctypes as C
    import torch
    rng = np.random.default_rng(seed)
    n = 200_000
    ic = np.zeros(n, np.float32)
    pos = 0
    while pos < n:                                                       # alternating quiet stretches and peak bursts
        pos += int(rng.integers(1, max_quiet))
        blen = int(rng.integers(1, 400))
        burst = (rng.random(blen) < density).astype(np.float32) * rng.choice([0.7, 0.7, 0.7, 2.0, 3.0], blen).astype(np.float32)
        ic[pos:pos + blen] = burst[:max(0, min(blen, n - pos))]
        pos += blen
    ia = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ramp = (np.arange(n) + 1j * 0).astype(np.complex64)                  # out[tag] = in[start] * exp(0): the ramp reveals `start`
    det = oracle.FrameDetector(N, CP, 0.6, min_peaks, gap)
    out, tags = det.run(ramp, ia, ic)
    L = ctx.lib
    L.jrc_frame_detector_scan_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_uint, C.c_uint, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3
    d_ia = torch.from_numpy(ia.view(np.float32).copy()).cuda()
    d_ic = torch.from_numpy(ic).cuda()
    d_marks = torch.zeros(n // 64 + 2, dtype=torch.int64, device="cuda")
    d_info = torch.zeros((max_frames, C.sizeof(jrc.SyncFrame)), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.check(L.jrc_frame_detector_scan_dev(ctx.h, N, CP, 0.6, min_peaks, gap, n, d_ia.data_ptr(), d_ic.data_ptr(), d_marks.data_ptr(), max_frames,
                                            d_info.data_ptr(), d_n.data_ptr(), None))
    ctx.sync()
    nf = int(d_n.cpu().item())
    raw = d_info[:nf].cpu().numpy().tobytes()
    info = [jrc.SyncFrame.from_buffer_copy(raw[i * C.sizeof(jrc.SyncFrame):(i + 1) * C.sizeof(jrc.SyncFrame)]) for i in range(nf)]
    assert nf == min(len(tags), max_frames)
    offs = [t[0] for t in tags] + [out.size]
    for k, f in enumerate(info):
        assert f.len == offs[k + 1] - offs[k], k
        assert f.len == 0 or f.start == int(round(out[offs[k]].real)), k
        assert abs(f.coarse_cfo - tags[k][1]) < 1e-6


def _front_end_outputs(jrc, env, monkeypatch, x, max_frames, n_list=None, fft_len=N, cp_len=CP, taps=None, windows=None):
    """mask words, frame list and frame rows of one front-end run in a context of its own (the switches are read when a context is made)"""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = jrc.Context(0)
    for k in env:
        monkeypatch.delenv(k)
    from conftest import GOLDEN  # noqa: F401
    import os
    o = np.load(os.path.join(GOLDEN, "ofdm_config_64.npz"))
    fe = jrc.SyncFrontEnd(fft_len, cp_len, 0.6, 10, 8 * (fft_len + cp_len), min(4 * (fft_len + cp_len), 4096), o["l_ltf_fir"] if taps is None else taps,
                          max_frames=max_frames, max_symbols=40, ctx=c)
    if windows:                                                  # (delay, window, power_window) other than the flowgraph's fft_len / 4, / 2, 3/4
        fe.cfg.delay, fe.cfg.window, fe.cfg.power_window = windows
    d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()).cuda()
    out = []
    for n in (n_list or [x.size]):
        fe._work = None
        fe.frames.zero_()
        torch.cuda.synchronize()
        fe.run(d_x, n)
        nf, info = fe.results()
        off = 2 * n * 8 + ((n + 1) & ~1) * 4
        words = fe._work[off:off + 8 * ((n + 63) // 64)].cpu().numpy().view(np.uint64).copy()
        out.append((words, nf, [(f.start, f.len, f.coarse_cfo, f.frame_start, f.fine_cfo, f.tag_value, f.n_out) for f in info],
                    fe.frames[:nf].cpu().numpy().tobytes()))
    c.close()
    return out


def test_front_end_forms_agree_bit_for_bit(jrc, monkeypatch, ofdm64):
    """the three forms of the front end's first stage — four samples per lane with the float pre-test (default), one sample per lane
    (JRC_SYNC_TILE), the three metric streams through HBM + fd_marks_kernel (JRC_SYNC_STREAMS) — leave the same peak mask, the same frame
    list (incl. the coarse CFO re-formed at the detections) and the same frame rows; capture lengths around the tile and word boundaries,
    noise at the level where the correlation sits at the threshold for many samples"""
    rng = np.random.default_rng(77)
    x, _ = capture(ofdm64, 9, cfo=0.008, n_frames=7, gap=1800)
    x = x + (0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    x[5000:5400] = 0                                                 # an all-zero stretch: 0 / 0 in the metric
    n_list = [x.size, x.size - 1, 1792 * 3, 1792 * 3 + 1, 1792 * 3 - 63, 2000, 64, 63, 1]
    want = _front_end_outputs(jrc, {"JRC_SYNC_STREAMS": "1"}, monkeypatch, x, 16, n_list)
    assert want[0][1] == 7 and int(sum(bin(int(w)).count("1") for w in want[0][0])) > 200
    for env in ({}, {"JRC_SYNC_TILE": "1"}):
        got = _front_end_outputs(jrc, env, monkeypatch, x, 16, n_list)
        for n, w, g in zip(n_list, want, got):
            assert np.array_equal(w[0], g[0]), (env, n, "peak mask")
            assert w[1] == g[1] and w[2] == g[2], (env, n, "frame list")
            assert w[3] == g[3], (env, n, "frame rows")


@pytest.mark.parametrize("fft_len", [256, 1024, 128])
def test_front_end_forms_agree_at_other_carrier_counts(jrc, monkeypatch, fft_len):
    """the same three forms at fft_len 256 and 1024 (windows of 8 / 12 and 32 / 48 runs of 16: 7 and 4 output waves per workgroup of the
    four-samples-per-lane kernel) and 128; bursts = ten repeats of a random quarter symbol (the short training field's periodicity, which is all
    the metric sees) in front of random symbols, in noise"""
    rng = np.random.default_rng(fft_len)
    cp = fft_len // 4
    q = fft_len // 4
    parts = []
    for k in range(5):
        stf = np.tile(rng.standard_normal(q) + 1j * rng.standard_normal(q), 10)
        body = rng.standard_normal(14 * (fft_len + cp)) + 1j * rng.standard_normal(14 * (fft_len + cp))
        parts += [np.zeros(3 * fft_len + 17 * k), stf, body, np.zeros(2 * fft_len)]
    x = np.concatenate(parts)
    x = (x + 0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    taps = (rng.standard_normal(64) + 1j * rng.standard_normal(64)).astype(np.complex64)
    n_list = [x.size, x.size - 777, 5 * 1024 + 3]
    want = _front_end_outputs(jrc, {"JRC_SYNC_STREAMS": "1"}, monkeypatch, x, 16, n_list, fft_len, cp, taps)
    assert want[0][1] >= 4
    for env in ({}, {"JRC_SYNC_TILE": "1"}):
        got = _front_end_outputs(jrc, env, monkeypatch, x, 16, n_list, fft_len, cp, taps)
        for n, w, g in zip(n_list, want, got):
            assert np.array_equal(w[0], g[0]), (env, n, "peak mask")
            assert w[1] == g[1] and w[2] == g[2], (env, n, "frame list")
            assert w[3] == g[3], (env, n, "frame rows")


def test_front_end_forms_agree_for_drawn_windows(jrc, monkeypatch, ofdm64):
    """delay / window / power_window drawn at random: whole runs of 16 take the four-samples-per-lane kernel, everything else the tile kernel's
    marks form; both against the streams form, bit for bit"""
    rng = np.random.default_rng(5)
    x, _ = capture(ofdm64, 12, cfo=0.004, n_frames=4, gap=1500)
    x = x + (0.08 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    cases = [(16, 32, 48), (16, 48, 64), (4, 16, 16), (32, 64, 128), (64, 112, 48), (16, 31, 48), (16, 32, 47), (15, 32, 48), (1, 1, 1), (7, 45, 90), (20, 100, 3)]
    cases += [tuple(int(v) for v in (rng.integers(0, 40), rng.integers(1, 120), rng.integers(1, 120))) for _ in range(8)]
    n_list = [x.size, 4000, 1793]
    for w in cases:
        want = _front_end_outputs(jrc, {"JRC_SYNC_STREAMS": "1"}, monkeypatch, x, 16, n_list, windows=w)
        got = _front_end_outputs(jrc, {}, monkeypatch, x, 16, n_list, windows=w)
        for n, a, b in zip(n_list, want, got):
            assert np.array_equal(a[0], b[0]), (w, n, "peak mask")
            assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3], (w, n)


@pytest.mark.parametrize("seed,cfo,n_frames,gap", [(21, 0.006, 24, 1200), (22, -0.009, 17, 3100)])
def test_batched_front_end_equals_the_block_chain_on_longer_captures(jrc, ctx, ofdm64, seed, cfo, n_frames, gap):
    """the run-to-completion front end (peak mask from the registers, correlation re-formed at the detections, segment-parallel scan with kept
    detections, output-indexed copy) against the oracle's frame_detector + frame_sync run sample by sample, on captures of 17 / 24 frames of
    growing length: the frame list exact (tag offsets = cumulated copy lengths), coarse CFO and tag value to 1e-6 / 1e-5, every row to 1e-4"""
    import torch
    x, meta = capture(ofdm64, seed, cfo=cfo, n_frames=n_frames, gap=gap)
    fe = jrc.SyncFrontEnd(N, CP, 0.6, 10, 8 * (N + CP), SYNC_LEN, ofdm64["l_ltf_fir"], max_frames=n_frames + 4, max_symbols=200, ctx=ctx)
    d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()).cuda()
    fe.run(d_x, x.size)
    n, info = fe.results()
    rows = fe.frames.cpu().numpy().view(np.complex64)[..., 0].reshape(n_frames + 4, -1)
    xd, ia, ic = oracle.sync_metrics(x, 16, 32, 48, 1 / 1.5)
    seg, dtags = oracle.FrameDetector(N, CP, 0.6, 10, 8 * (N + CP)).run(xd, ia, ic)
    delayed = np.concatenate([np.zeros(SYNC_LEN, np.complex64), seg])[:seg.size]
    out, otags = oracle.FrameSync(N, CP, SYNC_LEN, ofdm64["l_ltf_fir"]).run(seg, delayed, dtags)
    assert n == n_frames == len(dtags) == len(otags)
    assert [t[0] for t in dtags] == list(np.cumsum([0] + [f.len for f in info[:-1]]))
    for k in range(n):
        assert abs(info[k].coarse_cfo - dtags[k][1]) < 1e-6 and abs(info[k].tag_value - otags[k][1]) < 1e-5, k
        lo = otags[k][0]
        hi = otags[k + 1][0] if k + 1 < n else out.size
        want = out[lo:hi]
        m = min(want.size, info[k].n_out)
        if k + 1 < n:
            assert want.size == info[k].n_out, k
        assert m > 20 * N and rel_err(rows[k][:m], want[:m]) < TOL, k


def test_front_end_on_empty_and_frameless_captures(jrc, ctx, ofdm64):
    """nothing to find: an empty capture, a capture shorter than a window, noise only, silence — zero frames, no row touched, in all three forms'
    default one; and a list shorter than the capture's frames keeps the first ones"""
    import torch
    rng = np.random.default_rng(8)
    fe = jrc.SyncFrontEnd(N, CP, 0.6, 10, 8 * (N + CP), SYNC_LEN, ofdm64["l_ltf_fir"], max_frames=2, max_symbols=40, ctx=ctx)
    noise = (rng.standard_normal(50000) + 1j * rng.standard_normal(50000)).astype(np.complex64)
    for x, n in ((noise, 0), (noise, 1), (noise, 47), (noise, noise.size), (np.zeros(9000, np.complex64), 9000)):
        d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()).cuda()
        fe.frames.fill_(7.0)
        torch.cuda.synchronize()
        fe.run(d_x, n)
        nf, info = fe.results()
        assert nf == 0 and info == [], n
        assert float(fe.frames.min()) == 7.0 and float(fe.frames.max()) == 7.0
    x, meta = capture(ofdm64, 30, cfo=0.01, n_frames=5)
    d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()).cuda()
    fe.run(d_x, x.size)
    nf, info = fe.results()
    big = jrc.SyncFrontEnd(N, CP, 0.6, 10, 8 * (N + CP), SYNC_LEN, ofdm64["l_ltf_fir"], max_frames=8, max_symbols=40, ctx=ctx)
    big.run(d_x, x.size)
    nb, binfo = big.results()
    xd, ia, ic = oracle.sync_metrics(x, 16, 32, 48, 1 / 1.5)
    _, dtags = oracle.FrameDetector(N, CP, 0.6, 10, 8 * (N + CP)).run(xd, ia, ic)
    assert nb == len(dtags) >= 5 and nf == 2                   # (a payload can hold a plateau of its own: the detector's count is the oracle's)
    for a, b in zip(info, binfo[:2]):
        assert (a.start, a.len, a.coarse_cfo, a.frame_start, a.fine_cfo, a.n_out) == (b.start, b.len, b.coarse_cfo, b.frame_start, b.fine_cfo, b.n_out)
    assert torch.equal(fe.frames[:2], big.frames[:2])
