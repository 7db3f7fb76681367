"""GPU tier: the two chain modes added on top of the plain batched chain —
  * background recording / removal (lib/mimo_ofdm_radar_impl.cc:276-300) for batches of consecutive frames of one radar stream,
  * detect-only mode (no map stored; the estimator's message is the only consumer, lib/range_angle_estimator_impl.cc:234-253).
Both must reproduce, bit for bit, what the per-block path and the oracle produce."""
import ctypes

import numpy as np
import pytest

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _chain(jrc, ctx, sc, Ir, Ia, F, **kw):
    P = sc.T * sc.R
    rb, ab = jrc.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    ndr, nda = 2 * 3e8 / (2 * sc.fs), 2 * float(np.rad2deg(np.arcsin(2 / P))) if P > 2 else 30.0
    ch = jrc.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, ndr, nda, 15.0, 0.0, max_frames=F, ctx=ctx, **kw)
    return ch, (rb, ab, ndr, nda)


def _load(bufs, frames, n):
    import torch
    bufs["frames"][:n].copy_(torch.from_numpy(frames[:n].view(np.float32).reshape((n,) + tuple(bufs["frames"].shape[1:]))))
    torch.cuda.synchronize()


def _rec(r):
    return ctypes.string_at(ctypes.byref(r), ctypes.sizeof(r))


# ---- background removal ------------------------------------------------------------------------------------------------------
def _oracle_stream(sc, frames, record_len, recording_at=None, removal=True):
    """the oracle block over the frames of one stream, one general_work per frame; recording_at: {frame: bool} switches"""
    rad = oracle.Radar(sc.N, sc.T, sc.R, sc.S, sc.Npre, background_removal=removal, background_recording=True, record_len=record_len)
    out = []
    for f, fr in enumerate(frames):
        if recording_at and f in recording_at:
            rad.set_background_record(recording_at[f])
        out.append(rad.work([fr[t] for t in range(sc.T)], [fr[sc.T + r] for r in range(sc.R)])[:, :sc.N].copy())
    return np.stack(out), rad.ring_size()


@pytest.mark.parametrize("T,R,N,S,L", [(4, 2, 64, 4, 3), (2, 2, 128, 3, 8), (4, 4, 256, 8, 5), (1, 1, 64, 2, 1)])
def test_batched_background_removal_equals_sequential_blocks(jrc, ctx, T, R, N, S, L):
    """record_len + 3 frames: one batch through jrc_chain_run_dev == that many sequential mimo_ofdm_radar calls == the oracle block"""
    from jrc_amd import synth
    sc = synth.Scenario(N, T, R, S, targets=[(9.0, 12.0, 0.0, 80.0)])
    F = L + 3
    frames = synth.make_frames(sc, F)
    want, ring = _oracle_stream(sc, frames, L)
    # per-block path (jrc_radar_work keeps its own ring)
    blk = jrc.mimo_ofdm_radar(N, T, R, S, sc.Npre, background_removal=True, background_recording=True, record_len=L, ctx=ctx)
    seq = np.stack([blk.general_work([fr[t] for t in range(T)], [fr[T + r] for r in range(R)]) for fr in frames])
    assert np.array_equal(seq, want)
    ch, axes = _chain(jrc, ctx, sc, 4, 8, F)
    ch.set_background(True, True, L)
    bufs = ch.alloc(F, "cuda:0")
    _load(bufs, frames, F)
    ch.run(bufs, F)
    res = ch.results(bufs, F)
    got = bufs["chanest"].cpu().numpy().view(np.complex64)[..., 0]
    assert np.array_equal(got, want)
    assert ch.background_size() == ring == min(L, F)
    # the map and the estimate follow from the background-free channel estimate
    rb, ab, ndr, nda = axes
    gm = bufs["map"].cpu().numpy().view(np.complex64)[..., 0]
    for f in (0, F - 1):
        Hp = np.zeros((T * R, N * 4), np.complex64)
        Hp[:, :N] = want[f]
        m = oracle.fft_vcc(oracle.matrix_transpose(oracle.fft_vcc(Hp, False, False), N * 4, T * R, 8), True, True)
        d = np.abs(m).max()
        assert d == 0 or rel_err(gm[f], m) < 5e-6
        if d > 0:
            assert _rec(res[f]) == _rec(oracle.ra_estimate(gm[f], rb, ab, ndr, nda, 15.0, 0.0))


def test_background_state_carries_across_batches_and_recording_switch(jrc, ctx):
    """three batches (5 + 1 + 6 frames), recording switched off before the second and back on before the third: while it is off the
    reference keeps pushing the last recorded estimate (radar_chan_est_temp, :276-279, :297-300)"""
    from jrc_amd import synth
    sc = synth.Scenario(64, 2, 2, 3, targets=[(15.0, -20.0, 0.0, 50.0)])
    L, cuts = 4, [(0, 5, True), (5, 6, False), (6, 12, True)]
    frames = synth.make_frames(sc, 12)
    want, ring = _oracle_stream(sc, frames, L, recording_at={5: False, 6: True})
    ch, _ = _chain(jrc, ctx, sc, 2, 4, 6)
    ch.set_background(True, True, L)
    bufs = ch.alloc(6, "cuda:0")
    got = []
    for lo, hi, rec in cuts:
        ch.set_background_record(rec)
        _load(bufs, frames[lo:hi], hi - lo)
        ch.run(bufs, hi - lo)
        ctx.sync()
        got.append(bufs["chanest"][:hi - lo].cpu().numpy().view(np.complex64)[..., 0].copy())
    assert np.array_equal(np.concatenate(got), want)
    assert ch.background_size() == ring == L


def test_background_removal_off_recording_on_changes_nothing(jrc, ctx):
    from jrc_amd import synth
    sc = synth.Scenario(64, 2, 1, 3, targets=[(15.0, -20.0, 0.0, 50.0)])
    frames = synth.make_frames(sc, 4)
    want, ring = _oracle_stream(sc, frames, 3, removal=False)
    ch, _ = _chain(jrc, ctx, sc, 2, 4, 4)
    ch.set_background(False, True, 3)
    bufs = ch.alloc(4, "cuda:0")
    _load(bufs, frames, 4)
    ch.run(bufs, 4)
    ctx.sync()
    assert np.array_equal(bufs["chanest"].cpu().numpy().view(np.complex64)[..., 0], want)
    assert ch.background_size() == ring == 0


def test_sharded_stream_with_background_equals_one_gpu(jrc, ctx):
    """contiguous blocks of one stream on several GPUs; each replays the <= record_len frames in front of its block
    (jrc_chain_prime_background_dev) and then produces exactly the estimates a single GPU would"""
    import torch
    from jrc_amd import shard, synth
    sc = synth.Scenario(128, 2, 2, 4, targets=[(12.0, 25.0, 0.0, 70.0)])
    n, world, L = 23, 3, 4
    frames = synth.make_frames(sc, n)
    want, _ = _oracle_stream(sc, frames, L)
    got = []
    for rank in range(world):                    # the ranks of a 3-GPU job, one after the other on this GPU, each with its own chain
        first, lo, hi = shard.ring_warmup_block(n, rank, world, L)
        ch, _ = _chain(jrc, ctx, sc, 2, 4, max(hi - lo, lo - first, 1))
        ch.set_background(True, True, L)
        bufs = ch.alloc(max(hi - lo, lo - first, 1), "cuda:0")
        if lo > first:
            _load(bufs, frames[first:lo], lo - first)
            ch.prime_background(bufs["frames"], lo - first)
            ctx.sync()                                # the library's stream reads the frame buffer torch is about to overwrite
        _load(bufs, frames[lo:hi], hi - lo)
        ch.run(bufs, hi - lo)
        ctx.sync()
        got.append(bufs["chanest"][:hi - lo].cpu().numpy().view(np.complex64)[..., 0].copy())
    assert np.array_equal(np.concatenate(got), want)


def test_feed_with_background_shares_one_history(jrc, ctx):
    """the host-fed pipeline keeps several batches in flight on their own streams; with background removal they advance one history in
    submission order"""
    from jrc_amd import synth
    sc = synth.Scenario(64, 2, 2, 3, targets=[(15.0, -20.0, 0.0, 50.0)])
    L, fps, nb = 3, 2, 5
    frames = synth.make_frames(sc, fps * nb)
    want, _ = _oracle_stream(sc, frames, L)
    P = sc.T * sc.R
    rb, ab = jrc.radar_axes(sc.N, sc.fs, 2, P, 4)
    feed = jrc.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, 2, 4, rb, ab, 2.4, 29.0, 15.0, 0.0, ctx=ctx, n_slots=3, frames_per_slot=fps, maps_per_slot=fps)
    feed.set_background(True, True, L)
    maps = []
    for b in range(nb):
        if feed.pending() == 3:
            maps.append(feed.collect(want_maps=True)[1])
        feed.submit(frames[b * fps:(b + 1) * fps])
    while feed.pending():
        maps.append(feed.collect(want_maps=True)[1])
    maps = np.concatenate(maps)
    for f in range(fps * nb):
        Hp = np.zeros((P, sc.N * 2), np.complex64)
        Hp[:, :sc.N] = want[f]
        m = oracle.fft_vcc(oracle.matrix_transpose(oracle.fft_vcc(Hp, False, False), sc.N * 2, P, 4), True, True)
        assert rel_err(maps[f], m) < 5e-6, f


def test_background_removal_switched_on_through_a_sharing_chain(jrc, ctx):
    """the removal switch belongs to the shared state: a chain created recording-only (no raw-estimate buffer of its own) must be able to run
    after ANOTHER chain of the group turned removal on (it wrote its raw estimates to a null pointer).  Two chains alternating
    over one stream must produce what ONE chain produces going through the same switches (recording only, then removal on)."""
    from jrc_amd import synth
    sc = synth.Scenario(64, 2, 2, 3, targets=[(15.0, -20.0, 0.0, 50.0)])
    L, n = 3, 4
    frames = synth.make_frames(sc, 3 * n)
    raw, _ = _oracle_stream(sc, frames[:n], L, removal=False)     # recording only: the outputs are the raw estimates

    def chanest(b):
        return b["chanest"].cpu().numpy().view(np.complex64)[..., 0].copy()

    single, _ = _chain(jrc, ctx, sc, 2, 4, n)
    single.set_background(False, True, L)
    b1 = single.alloc(n, "cuda:0")
    want = []
    for k in range(3):
        if k == 1:
            single.set_background(True, True, L)
        _load(b1, frames[k * n:(k + 1) * n], n)
        single.run(b1, n)
        ctx.sync()
        want.append(chanest(b1))
    assert np.array_equal(want[0], raw)

    owner, _ = _chain(jrc, ctx, sc, 2, 4, n)
    other, _ = _chain(jrc, ctx, sc, 2, 4, n)
    owner.set_background(False, True, L)                          # recording only: allocates no raw buffer
    other.share_background(owner)
    bo, bs = owner.alloc(n, "cuda:0"), other.alloc(n, "cuda:0")
    _load(bo, frames[:n], n)
    owner.run(bo, n)
    ctx.sync()
    assert np.array_equal(chanest(bo), want[0])
    other.set_background(True, True, L)                           # the sharer switches removal on for the group
    _load(bs, frames[n:2 * n], n)
    other.run(bs, n)
    ctx.sync()
    assert np.array_equal(chanest(bs), want[1])
    _load(bo, frames[2 * n:], n)
    owner.run(bo, n)                                              # the owner now subtracts too: needs its own raw buffer
    ctx.sync()
    assert np.array_equal(chanest(bo), want[2])
    assert not np.array_equal(want[2], _oracle_stream(sc, frames[2 * n:], L, removal=False)[0])    # something was subtracted


# ---- detect-only mode -----------------------------------------------------------------------------------------------------------
def _both_modes(jrc, ctx, sc, Ir, Ia, F, frames=None, interleave=False):
    import torch
    from jrc_amd import synth
    if frames is None:
        frames = synth.make_frames(sc, F)
    ch, axes = _chain(jrc, ctx, sc, Ir, Ia, F, enable_tx_interleave=interleave)
    bufs = ch.alloc(F, "cuda:0")
    _load(bufs, frames, F)
    ch.run(bufs, F)
    full = [_rec(r) for r in ch.results(bufs, F)]
    H = bufs["chanest"].clone()
    ch.set_write_map(False)
    b2 = ch.alloc(F, "cuda:0", with_map=False)
    b2["frames"].copy_(bufs["frames"])
    torch.cuda.synchronize()
    ch.run(b2, F)
    det = [_rec(r) for r in ch.results(b2, F)]
    assert torch.equal(b2["chanest"], H)
    ch.set_write_map(True)                       # and back
    bufs["map"].zero_()
    torch.cuda.synchronize()                     # zero_ runs on torch's stream, the chain on the context's
    ch.run(bufs, F)
    again = [_rec(r) for r in ch.results(bufs, F)]
    return full, det, again, ch.results(bufs, F)


@pytest.mark.parametrize("cfg,F", [("A", 5), ("A", 300), ("B", 3), ("B", 40), ("B", 512), ("B", 1100), ("D", 2), ("D", 9), ("D", 256)])
def test_detect_only_results_equal_map_mode(jrc, ctx, cfg, F):
    """B x 1100: three launches of the fused kernel (512 + 512 + a ragged 76-frame tail with more slices per frame)"""
    from jrc_amd import synth
    sc = {"A": synth.config_A, "B": synth.config_B, "D": synth.config_D}[cfg]()
    nd = min(F, 4)
    base = synth.make_frames(sc, nd)
    frames = np.concatenate([base] * (F // nd + 1))[:F]
    frames = frames * (1.0 + 0.25 * np.arange(F, dtype=np.float32))[:, None, None, None]      # distinct peaks per frame (exact scaling)
    full, det, again, res = _both_modes(jrc, ctx, sc, 8, 16, F, frames=frames.astype(np.complex64))
    assert det == full and again == full
    assert all(r.n_noise_samples > 0 for r in res)


@pytest.mark.parametrize("N,S,Ir,F", [(256, 3, 16, 5), (512, 2, 8, 4), (256, 2, 32, 3)])
def test_detect_only_and_power_map_on_the_wide_kernel_at_smaller_fft_len(jrc, ctx, N, S, Ir, F):
    """range_angle_wide_kernel at fft_len 256 / 512 (long range axis): the three MODEs agree bit for bit as everywhere else"""
    import torch
    from jrc_amd import synth
    sc = synth.Scenario(N, 4, 4, S, targets=[(9.0, 25.0, 0.0, 40.0)])
    frames = synth.make_frames(sc, F)
    frames = (frames * (1.0 + 0.5 * np.arange(F, dtype=np.float32))[:, None, None, None]).astype(np.complex64)
    full, det, again, res = _both_modes(jrc, ctx, sc, Ir, 16, F, frames=frames)
    assert det == full and again == full
    ch, axes = _chain(jrc, ctx, sc, Ir, 16, F)
    bufs = ch.alloc(F, "cuda:0")
    _load(bufs, frames, F)
    ch.run(bufs, F)
    want = [_rec(r) for r in ch.results(bufs, F)]
    cmap = bufs["map"].cpu().numpy().view(np.complex64)[..., 0]
    ch.set_map_format(True)
    bp = ch.alloc(F, "cuda:0", power_map=True)
    _load(bp, frames, F)
    ch.run(bp, F)
    assert [_rec(r) for r in ch.results(bp, F)] == want
    pw = bp["map"].cpu().numpy()
    ref = (cmap.real.astype(np.float32) * cmap.real.astype(np.float32)) + (cmap.imag.astype(np.float32) * cmap.imag.astype(np.float32))
    assert np.array_equal(pw.reshape(ref.shape), ref)


@pytest.mark.parametrize("case", ["noise_only", "two_equal_targets", "exact_duplicate_rows", "all_zero", "weak_target_in_noise", "late_peak"])
@pytest.mark.parametrize("cfg,F", [("B", 6), ("B", 300), ("B", 700), ("D", 3)])
def test_detect_only_bound_pruning_keeps_the_records_exact(jrc, ctx, cfg, F, case):
    """detect-only mode skips the angle transforms of range bins whose bound (sum_p |R[p][k]|)^2 lies below the running maximum
    (range_angle_wide_kernel, MODE 1).  Inputs chosen against that shortcut: nothing to prune with (noise only), maxima that tie to the
    last bit in different range bins (two equal targets; frames whose RX ports are exact copies, so map cells repeat bit for bit), an
    all-zero frame (every bound equals the maximum, 0), a target a few dB over the noise, and a peak in the last class / last bins of
    the scan order.  The records must stay byte-identical to map mode (lib/range_angle_estimator_impl.cc:137-151: first maximum in
    scan order)."""
    from jrc_amd import synth
    sc0 = {"B": synth.config_B, "D": synth.config_D}[cfg]()
    rng = np.random.default_rng(sum(map(ord, cfg + case)) * 1000 + F)
    R_max = 3e8 * sc0.N / (2 * sc0.fs)
    if case == "two_equal_targets":
        sc = synth.Scenario(sc0.N, sc0.T, sc0.R, sc0.S, targets=[(0.2 * R_max, 20.0, 0.0, 100.0), (0.55 * R_max, -20.0, 0.0, 100.0)])
    elif case == "late_peak":
        sc = synth.Scenario(sc0.N, sc0.T, sc0.R, sc0.S, targets=[(0.9995 * R_max, 58.0, 0.0, 100.0)])
    else:
        sc = synth.Scenario(sc0.N, sc0.T, sc0.R, sc0.S, targets=[(0.31 * R_max, -12.0, 0.0, 100.0)])
    nd = min(F, 3)
    base = synth.make_frames(sc, nd)
    if case == "noise_only":
        base[:, sc.T:] = (rng.standard_normal(base[:, sc.T:].shape) + 1j * rng.standard_normal(base[:, sc.T:].shape)).astype(np.complex64) * 1e-3
    elif case == "weak_target_in_noise":
        sig = np.abs(base[:, sc.T:]).mean()
        base[:, sc.T:] += ((rng.standard_normal(base[:, sc.T:].shape) + 1j * rng.standard_normal(base[:, sc.T:].shape)) * 6.0 * sig).astype(np.complex64)
    elif case == "exact_duplicate_rows":
        base[:, sc.T + 1:] = base[:, sc.T:sc.T + 1]            # every RX port carries the same samples: the angle axis has exact repeats
    elif case == "all_zero":
        base[:] = 0
    frames = np.concatenate([base] * (F // nd + 1))[:F]
    if case != "all_zero":
        frames = frames * (1.0 + 0.5 * (np.arange(F, dtype=np.float32) % 7))[:, None, None, None]
    full, det, again, res = _both_modes(jrc, ctx, sc, 8, 16, F, frames=frames.astype(np.complex64))
    assert det == full and again == full


@pytest.mark.parametrize("T,R,N,F", [(4, 2, 256, 7), (4, 2, 1024, 3), (2, 4, 512, 5), (4, 2, 256, 600)])
@pytest.mark.parametrize("case", ["target", "noise_only"])
def test_detect_only_pruning_with_eight_pairs(jrc, ctx, T, R, N, F, case):
    """the wide kernel's 8-pair geometry (two pairs per wave; only half of a range bin's 16 lanes hold a pair when the bound is summed)
    in detect-only mode against map mode, with and without something to prune"""
    from jrc_amd import synth
    sc = synth.Scenario(N, T, R, 4, targets=[(0.4 * 3e8 * N / (2 * 125e6), 17.0, 0.0, 100.0)])
    base = synth.make_frames(sc, min(F, 3))
    if case == "noise_only":
        rng = np.random.default_rng(N + F)
        base[:, T:] = ((rng.standard_normal(base[:, T:].shape) + 1j * rng.standard_normal(base[:, T:].shape)) * 1e-3).astype(np.complex64)
    frames = np.concatenate([base] * (F // len(base) + 1))[:F]
    frames = (frames * (1.0 + 0.5 * (np.arange(F, dtype=np.float32) % 5))[:, None, None, None]).astype(np.complex64)
    full, det, again, res = _both_modes(jrc, ctx, sc, 8, 16, F, frames=frames)
    assert det == full and again == full


def _shapes(n, seed=77):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        T, R = int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 4]))
        N = int(rng.choice([64, 128, 256, 512, 1024]))
        S = int(rng.integers(1, 6))
        Ir, Ia = int(rng.choice([1, 2, 4, 8])), int(rng.choice([2, 4, 8, 16, 32]))
        if T * R * Ia >= 4 and N * Ir * T * R * Ia <= 1 << 21:
            out.append((T, R, N, S, Ir, Ia, bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("T,R,N,S,Ir,Ia,interleave", _shapes(16))
def test_detect_only_random_shapes(jrc, ctx, T, R, N, S, Ir, Ia, interleave):
    from jrc_amd import synth
    sc = synth.Scenario(N, T, R, S, targets=[(0.6 * 3e8 * N / (2 * 125e6) * 0.5, -35.0, 0.0, 90.0)])
    full, det, again, res = _both_modes(jrc, ctx, sc, Ir, Ia, 3, interleave=interleave)
    assert det == full and again == full


def test_detect_only_window_wraps_around_the_range_axis(jrc, ctx):
    """a peak in the upper half of the range axis puts the noise window across bin NR-1 -> 0 (:211): the window rows wrap"""
    from jrc_amd import synth
    sc = synth.Scenario(64, 2, 2, 4, targets=[(0.97 * 3e8 * 64 / (2 * 125e6) / 2, 10.0, 0.0, 100.0)])
    full, det, again, res = _both_modes(jrc, ctx, sc, 8, 16, 4)
    assert det == full
    NR = 64 * 8
    assert any(r.peak_range_idx + NR // 2 + r.discard_range_idx > NR for r in res) or any(r.peak_range_idx + NR // 2 - r.discard_range_idx < NR <= r.peak_range_idx + NR // 2 + r.discard_range_idx for r in res)


def test_detect_only_refuses_shapes_outside_the_fused_kernel(jrc, ctx):
    from jrc_amd import synth
    sc = synth.Scenario(32, 2, 2, 3, targets=[(6.0, -15.0, 0.0, 80.0)])
    ch, _ = _chain(jrc, ctx, sc, 4, 4, 2)
    with pytest.raises(jrc.JrcError) as e:
        ch.set_write_map(False)
    assert e.value.status == jrc.JRC_ERR_UNSUPPORTED


def test_feed_in_detect_only_mode(jrc, ctx):
    from jrc_amd import synth
    sc = synth.Scenario(64, 2, 2, 4, targets=[(12.0, -25.0, 0.0, 100.0)])
    P = sc.T * sc.R
    rb, ab = jrc.radar_axes(sc.N, sc.fs, 4, P, 8)
    frames = synth.make_frames(sc, 8)
    out = {}
    for mode in (True, False):
        feed = jrc.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, 4, 8, rb, ab, 2.4, 29.0, 15.0, 0.0, ctx=ctx, n_slots=2, frames_per_slot=4)
        feed.set_write_map(mode)
        got = []
        for b in range(2):
            feed.submit(frames[4 * b:4 * b + 4])
        while feed.pending():
            got += feed.collect()[0]
        out[mode] = [_rec(r) for r in got]
        feed.close()
    assert out[True] == out[False] and len(out[True]) == 8


# ---- one host process, several GPUs ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_multi_device_feed_equals_single_device_feed(jrc, ctx, devices):
    """jrc_chain_feed_create_multi deals the batches round-robin over per-device contexts (here the one GPU of the box, listed several
    times) and returns results in submission order: result for result what the single-device feed returns, for single submits, in-place
    (acquire) submits and submit_many (the per-device host threads stage and enqueue in parallel)"""
    from jrc_amd import synth
    sc = synth.Scenario(128, 2, 2, 4, targets=[(12.0, -25.0, 0.0, 100.0)])
    P = sc.T * sc.R
    rb, ab = jrc.radar_axes(sc.N, sc.fs, 4, P, 8)
    fps, nb = 3, 11
    frames = synth.make_frames(sc, fps * nb)
    frames[:, sc.T:] *= (1.0 + 0.03 * np.arange(fps * nb, dtype=np.float32))[:, None, None, None]
    batches = [frames[b * fps:(b + 1) * fps] for b in range(nb)]
    batches[-1] = batches[-1][:2]                                   # a short last batch

    def drain(feed, out, maps):
        r, m = feed.collect(want_maps=True)
        out += [_rec(x) for x in r]
        maps.append(m)

    def run(feed, mode):
        out, maps = [], []
        if mode == "many":
            k = 0
            while k < nb:
                free = feed.n_slots - feed.pending()
                if free == 0:
                    drain(feed, out, maps)
                    continue
                n = min(free, nb - k, len(devices))
                feed.submit_many(batches[k:k + n])
                k += n
        else:
            for b in range(nb):
                if feed.pending() == feed.n_slots:
                    drain(feed, out, maps)
                if mode == "inplace" and len(batches[b]) == fps:
                    feed.acquire()[:] = batches[b]
                    feed.submit(None, fps)
                else:
                    feed.submit(batches[b])
        while feed.pending():
            drain(feed, out, maps)
        return out, np.concatenate(maps)

    single = jrc.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, 4, 8, rb, ab, 2.4, 29.0, 15.0, 0.0, ctx=ctx, n_slots=2, frames_per_slot=fps, maps_per_slot=1)
    want, want_maps = run(single, "single")
    assert len(want) == fps * (nb - 1) + 2
    for mode in ("single", "inplace", "many"):
        multi = jrc.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, 4, 8, rb, ab, 2.4, 29.0, 15.0, 0.0, ctx=ctx, n_slots=2, frames_per_slot=fps, maps_per_slot=1,
                              devices=devices, graph=(mode == "inplace"))
        assert multi.n_devices() == len(devices) and multi.n_slots == 2 * len(devices)
        got, got_maps = run(multi, mode)
        assert got == want, mode
        assert np.array_equal(got_maps, want_maps), mode
        multi.close()
    single.close()


def test_multi_device_feed_with_a_device_that_does_not_exist_is_refused_and_leaks_nothing(jrc, ctx):
    """the device list of jrc_chain_feed_create_multi (the radar_chain block's JRC_DEVICES) names a device the process cannot
    open — first, in the middle, last: creation fails loudly (jrc_create refuses the index), the contexts and slots made for the
    devices before it are destroyed again (device memory in use afterwards = before, over twenty refused creations), and a list of eight entries
    that do exist (the box's one GPU eight times: BASELINE config 5's device count) works and reports eight devices"""
    import torch
    from jrc_amd import synth
    sc = synth.Scenario(256, 2, 2, 3, targets=[(15.0, -20.0, 0.0, 50.0)])
    P = sc.T * sc.R
    rb, ab = jrc.radar_axes(sc.N, sc.fs, 8, P, 16)                  # 1 MiB of map per frame: a slot of 8 frames holds ~9 MB on the device
    mk = lambda devices: jrc.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, 8, 16, rb, ab, 2.4, 29.0, 15.0, 0.0, ctx=ctx, n_slots=2, frames_per_slot=8,
                                       maps_per_slot=1, devices=devices)
    ctx.sync()
    torch.cuda.synchronize()
    mk([0, 0]).close()                                              # whatever the first creation caches (twiddle tables of the context) exists now
    free0 = torch.cuda.mem_get_info()[0]
    whole = mk([0, 0, 0])
    held = free0 - torch.cuda.mem_get_info()[0]                     # what one feed over three contexts holds: the unit a leak would come in
    whole.close()
    assert held > (32 << 20)
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 1
    missing = max(n_dev, 1) + 57                                    # no such device on any box
    for rep in range(7):
        for devices in ([missing, 0, 0], [0, missing, 0], [0, 0, 0, missing]):
            with pytest.raises(jrc.JrcError) as e:
                mk(devices)
            assert e.value.status in (jrc.JRC_ERR_INVALID_ARG, jrc.JRC_ERR_NO_DEVICE), e.value      # jrc_create: a device index outside [0, count)
    ctx.sync()
    # 21 refused creations, each after one or two contexts with their slots had been made: leaked, they would hold several times `held`
    assert free0 - torch.cuda.mem_get_info()[0] < held // 2, (free0, torch.cuda.mem_get_info()[0], held)
    eight = mk([0] * 8)
    assert eight.n_devices() == 8 and eight.n_slots == 16
    frames = synth.make_frames(sc, 2)
    for _ in range(16):
        eight.submit(frames)
    got = []
    while eight.pending():
        got += [_rec(x) for x in eight.collect()[0]]
    assert len(got) == 32 and all(g == got[i % 2] for i, g in enumerate(got))          # the same two frames through every slot of every context
    eight.close()


def test_multi_device_feed_refuses_background_and_one_feed_per_stream_carries_it(jrc, ctx):
    """background removal orders the frames of ONE radar stream (lib/mimo_ofdm_radar_impl.cc:281-300): a feed that deals batches over devices
    refuses it loudly (no silent per-device histories), and the supported layout — one feed per stream, each on its own device context, all
    driven from one host thread with their batches interleaved and in flight together — gives every stream exactly the estimates it gets
    alone.  The devices are the one GPU of the box under two contexts."""
    from jrc_amd import synth
    sc = synth.Scenario(64, 2, 2, 3, targets=[(15.0, -20.0, 0.0, 50.0)])
    P, L, fps, nb = sc.T * sc.R, 3, 2, 6
    rb, ab = jrc.radar_axes(sc.N, sc.fs, 2, P, 4)
    mk = lambda c, **kw: jrc.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, 2, 4, rb, ab, 2.4, 29.0, 15.0, 0.0, ctx=c, n_slots=3, frames_per_slot=fps,
                                       maps_per_slot=fps, **kw)
    multi = mk(ctx, devices=[0, 0])
    with pytest.raises(jrc.JrcError) as e:
        multi.set_background(True, True, L)
    assert e.value.status == jrc.JRC_ERR_UNSUPPORTED and "one feed per stream" in str(e.value)
    multi.set_background(False, False, L)                           # (off is accepted)
    multi.close()
    streams = [synth.make_frames(sc, fps * nb, first_frame=1000 * k) for k in range(2)]
    for k in range(2):
        streams[k][:, sc.T:] *= (1.0 + 0.05 * (k + 1) * np.arange(fps * nb, dtype=np.float32))[:, None, None, None]

    def drive(feeds, frames_of):
        maps = [[] for _ in feeds]
        for b in range(nb):
            for k, fd in enumerate(feeds):                          # interleaved: both feeds have batches in flight at once
                if fd.pending() == 3:
                    maps[k].append(fd.collect(want_maps=True)[1])
                fd.submit(frames_of[k][b * fps:(b + 1) * fps])
        for k, fd in enumerate(feeds):
            while fd.pending():
                maps[k].append(fd.collect(want_maps=True)[1])
        return [np.concatenate(m) for m in maps]

    ctx2 = jrc.Context(0)
    feeds = [mk(ctx), mk(ctx2)]
    for fd in feeds:
        fd.set_background(True, True, L)
    together = drive(feeds, streams)
    for fd in feeds:
        fd.close()
    for k in range(2):
        alone = mk(ctx)
        alone.set_background(True, True, L)
        want = drive([alone], [streams[k]])[0]
        alone.close()
        assert np.array_equal(together[k], want), k
        ref, _ = _oracle_stream(sc, streams[k], L)                    # and the history is the reference's: map of frame f from the oracle's estimate
        for f in (0, L, fps * nb - 1):
            Hp = np.zeros((P, sc.N * 2), np.complex64)
            Hp[:, :sc.N] = ref[f]
            m = oracle.fft_vcc(oracle.matrix_transpose(oracle.fft_vcc(Hp, False, False), sc.N * 2, P, 4), True, True)
            assert rel_err(together[k][f], m) < 5e-6, (k, f)
    ctx2.close()


def test_time_domain_entry_with_background_and_detect_only(jrc, ctx):
    """jrc_chain_run_td_dev (A6 + A7 + A1 as one kernel) goes through the same background step and the same detect-only modes as the
    frequency-domain entry: its detect-only records equal its own map-mode records byte for byte, and its background-free estimates
    agree with the frequency-domain entry's to the rounding of the two A1 kernels (relative to the raw estimate's magnitude)"""
    import torch
    from jrc_amd import synth
    sc = synth.Scenario(64, 4, 2, 4, targets=[(9.0, 15.0, 0.0, 80.0)])
    Ir, Ia, F, L = 8, 16, 6, 2
    n_items, N, cp = sc.Npre + sc.S, sc.N, sc.N // 4
    fr = synth.make_frames(sc, F)
    fr[:, sc.T:] *= (1.0 + 0.1 * np.arange(F, dtype=np.float32))[:, None, None, None]
    x = np.fft.ifft(np.fft.ifftshift(fr[:, sc.T:], axes=-1), axis=-1).astype(np.complex64)
    td = np.concatenate([x[..., N - cp:], x], axis=-1).reshape(F, sc.R, n_items * (N + cp))
    d_tx = torch.from_numpy(np.ascontiguousarray(fr[:, :sc.T]).view(np.float32).reshape(F, sc.T, n_items, N, 2)).cuda()
    d_td = torch.from_numpy(td.view(np.float32).reshape(F, sc.R, -1, 2)).cuda()
    out = {}
    for mode in ("td_map", "td_detect", "fd_map"):
        ch, _ = _chain(jrc, ctx, sc, Ir, Ia, F)
        ch.set_background(True, True, L)
        ch.set_write_map(mode != "td_detect")
        bufs = ch.alloc(F, "cuda:0", with_map=(mode != "td_detect"))
        if mode == "fd_map":
            rx_tmp = torch.empty((F, sc.R, n_items, N, 2), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            ctx.check(ctx.lib.jrc_cp_remove_fft_dev(ctx.h, N, cp, F * sc.R * n_items, d_td.data_ptr(), rx_tmp.data_ptr(), None))
            ctx.sync()
            bufs["frames"][:, :sc.T] = d_tx
            bufs["frames"][:, sc.T:] = rx_tmp
            torch.cuda.synchronize()
            ch.run(bufs, F)
        else:
            torch.cuda.synchronize()
            ch.run_td(bufs, d_tx, d_td, F, cp)
        res = ch.results(bufs, F)
        out[mode] = ([_rec(r) for r in res], bufs["chanest"].cpu().numpy().view(np.complex64)[..., 0].copy())
        assert ch.background_size() == L
    assert out["td_detect"][0] == out["td_map"][0]
    assert np.array_equal(out["td_detect"][1], out["td_map"][1])
    raw = np.abs(np.einsum("frsn,ftsn->frtn", fr[:, sc.T:, sc.Npre:], np.conj(fr[:, :sc.T, sc.Npre:]))).max()
    assert np.abs(out["td_map"][1] - out["fd_map"][1]).max() < 2e-6 * raw


# ---- power-map format (the heat-map branch's stream) -----------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,F", [("A", 7), ("B", 5), ("B", 512), ("D", 3), ("D", 256)])
def test_power_map_equals_magnitude_squared_of_the_complex_map(jrc, ctx, cfg, F):
    """JRC_MAP_POWER: the map as float |z|^2 = re*re + im*im (blocks_complex_to_mag_squared, ...radar_sim.grc:2192) — every cell equal to
    that expression on the complex map's cell bit for bit, result records byte-identical to the complex format; config D takes the
    one-row-at-a-time store tile (LDS), the others the full tile"""
    import torch
    from jrc_amd import synth
    sc = {"A": synth.config_A, "B": synth.config_B, "D": synth.config_D}[cfg]()
    nd = min(F, 4)
    base = synth.make_frames(sc, nd)
    frames = np.concatenate([base] * (F // nd + 1))[:F]
    frames = (frames * (1.0 + 0.25 * np.arange(F, dtype=np.float32))[:, None, None, None]).astype(np.complex64)
    ch, _ = _chain(jrc, ctx, sc, 8, 16, F)
    bufs = ch.alloc(F, "cuda:0")
    _load(bufs, frames, F)
    ch.run(bufs, F)
    want = [_rec(r) for r in ch.results(bufs, F)]
    ch.set_map_format(True)
    assert ch.map_bytes * 2 == ch.NR * ch.NA * 8
    pb = ch.alloc(F, "cuda:0", power_map=True)
    pb["frames"].copy_(bufs["frames"])
    pb["map"].fill_(float("nan"))
    torch.cuda.synchronize()
    ch.run(pb, F)
    got = [_rec(r) for r in ch.results(pb, F)]
    assert got == want
    re, im = bufs["map"][..., 0], bufs["map"][..., 1]
    assert torch.equal(pb["map"], re * re + im * im)          # two roundings of the products, one of the sum: no FMA
    assert torch.equal(pb["chanest"], bufs["chanest"])
    ch.set_map_format(False)                                   # back to the complex format
    bufs["map"].zero_()
    torch.cuda.synchronize()
    ch.run(bufs, F)
    assert [_rec(r) for r in ch.results(bufs, F)] == want


@pytest.mark.parametrize("T,R,N,S,Ir,Ia,interleave", _shapes(10, seed=5))
def test_power_map_random_shapes(jrc, ctx, T, R, N, S, Ir, Ia, interleave):
    import torch
    from jrc_amd import synth
    sc = synth.Scenario(N, T, R, S, targets=[(0.3 * 3e8 * N / (2 * 125e6) * 0.5, 25.0, 0.0, 90.0)])
    F = 3
    ch, _ = _chain(jrc, ctx, sc, Ir, Ia, F, enable_tx_interleave=interleave)
    bufs = ch.alloc(F, "cuda:0")
    _load(bufs, synth.make_frames(sc, F), F)
    ch.run(bufs, F)
    want = [_rec(r) for r in ch.results(bufs, F)]
    ch.set_map_format(True)
    pb = ch.alloc(F, "cuda:0", power_map=True)
    pb["frames"].copy_(bufs["frames"])
    pb["map"].fill_(float("nan"))
    torch.cuda.synchronize()
    ch.run(pb, F)
    assert [_rec(r) for r in ch.results(pb, F)] == want
    re, im = bufs["map"][..., 0], bufs["map"][..., 1]
    assert torch.equal(pb["map"], re * re + im * im)


@pytest.mark.gpu
@pytest.mark.spawns
def test_work_skipping_experiment_switches_do_nothing_in_this_library():
    """JRC_DETECT_EXP / JRC_RD_EXP bits that leave work out of a kernel (tools/detect_exp.sh, tools/rd_exp.sh; the equalizer's JRC_EQ_EXP) exist
    only in a library built with -DJRC_TIMING_EXPERIMENTS.  A process that happens to inherit them with the shipped library is told they are
    ignored and gets the same records as without them; the switches that keep the results (8, 16) stay silent."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, hashlib; sys.path.insert(0, %r)
import numpy as np, torch, jrc_amd
from jrc_amd import synth
c = jrc_amd.Context(0)
sc = synth.Scenario(256, 4, 4, 16, targets=[(12.0, 15.0, 0.0, 100.0)])
rb, ab = jrc_amd.radar_axes(256, sc.fs, 8, 16, 16)
chain = jrc_amd.RadarChain(256, 4, 4, 16, sc.Npre, 8, 16, rb, ab, 2.4, 14.4, 15.0, 0.0, max_frames=6, ctx=c)
bufs = chain.alloc(6, "cuda:0")
fr = synth.make_frames(sc, 6)
bufs["frames"].copy_(torch.from_numpy(fr.view(np.float32).reshape(bufs["frames"].shape))); torch.cuda.synchronize()
chain.set_write_map(False); chain.run(bufs, 6); c.sync()
print("records", hashlib.sha256(bufs["results"].cpu().numpy().tobytes()).hexdigest())
""" % root
    seen = {}
    for env, told in (({}, False), ({"JRC_RD_EXP": "1"}, True), ({"JRC_DETECT_EXP": "32"}, True), ({"JRC_DETECT_EXP": "3"}, True), ({"JRC_DETECT_EXP": "8"}, False),
                      ({"JRC_EQ_EXP": "2"}, False)):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-1000:]
        assert ("are ignored" in r.stderr) == told and "RESULTS ARE WRONG" not in r.stderr, (env, r.stderr[-500:])
        seen[str(env)] = [l for l in r.stdout.splitlines() if l.startswith("records")][-1]
    assert len(set(seen.values())) == 1, seen
