"""CPU tier: the sync front-end restatement (oracle/jrc_oracle_sync.c; reference lib/moving_avg_impl.cc,
lib/frame_detector_impl.cc, lib/frame_sync_impl.cc) — closed forms for the moving averages, and an end-to-end check that
the restated detector + synchroniser hand the (restated) equalizer a frame it decodes: precoder -> OFDM modulator -> delay,
carrier offset, noise -> autocorrelation metrics -> frame_detector -> frame_sync -> FFT -> equalizer -> stream_decoder."""
import numpy as np
import pytest

import oracle
from test_oracle_comm import make_blocks

N, CP = 64, 16


def ofdm_mod_np(sym_f):
    """fft_vxx(reverse, shift, window 1/sqrt(N)) + cyclic prefixer, numpy"""
    t = np.fft.ifft(np.fft.ifftshift(sym_f, axes=-1), axis=-1) * N / np.sqrt(N)
    return np.concatenate([t[..., -CP:], t], axis=-1).reshape(sym_f.shape[:-2] + (-1,)).astype(np.complex64)


def make_stream(o, payload, mcs, rng, lead=700, tail=3000, cfo=0.0, noise=0.02, h=None):
    pre, _ = make_blocks(o)
    sym, tags = oracle.stream_encode(mcs, 48, payload, 1)
    tx_f = pre.work(sym, mcs, 2, tags["pdu_len"])                      # [T][n_total][N]
    tx_t = ofdm_mod_np(tx_f)
    h = np.array([1.0, 0.5j, -0.3, 0.2 + 0.1j], np.complex64) if h is None else h
    frame = np.tensordot(h, tx_t, axes=(0, 0))
    x = np.concatenate([np.zeros(lead, np.complex64), frame, np.zeros(tail, np.complex64)])
    x = x * np.exp(1j * cfo * np.arange(x.size))
    # noise floor ~30 dB under the frame: far above the float drift of the reference's running-sum moving averages, which at a
    # 60 dB step fakes correlation peaks after a strong frame (why the flowgraph caps max_iter and takes abs())
    x = x + noise * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))
    return x.astype(np.complex64), tags, frame.size


def test_moving_avg_is_a_window_sum_and_honours_max_iter():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(500) + 1j * rng.standard_normal(500)).astype(np.complex64)
    y = oracle.moving_avg(x, 32, scale=0.5)
    want = 0.5 * np.convolve(np.concatenate([np.zeros(31), x]), np.ones(32), "valid")
    assert y.shape == (500,) and np.abs(y - want).max() < 1e-4
    assert oracle.moving_avg(x, 32, max_iter=100).shape == (100,)        # :78


def test_metrics_plateau_on_the_short_training_field(ofdm64):
    rng = np.random.default_rng(1)
    x, _, _ = make_stream(ofdm64, bytes([2]) + bytes(60), 2, rng)
    xd, ia, ic = oracle.sync_metrics(x, 16, 32, 48, 1 / 1.5)
    assert np.array_equal(xd[16:], x[:-16]) and not xd[:16].any()
    plateau = ic[700 + 60:700 + 150]                                     # inside the two STF symbols (period 16)
    assert plateau.min() > 0.8 and ic[100:600].max() < 0.6


@pytest.mark.parametrize("cfo", [0.0, 0.01, -0.02])
def test_detector_and_sync_deliver_a_decodable_frame(ofdm64, cfo):
    rng = np.random.default_rng(5)
    payload = bytes([2]) + rng.integers(0, 256, 99, dtype=np.uint8).tobytes()
    mcs = 2
    x, tags, flen = make_stream(ofdm64, payload, mcs, rng, cfo=cfo)
    xd, ia, ic = oracle.sync_metrics(x, 16, 32, 48, 1 / 1.5)
    det = oracle.FrameDetector(N, CP, 0.6, 10, (4 + 4) * (N + CP))
    seg, dtags = det.run(xd, ia, ic)
    assert len(dtags) == 1 and dtags[0][0] == 0                          # one frame_start tag at the first copied sample
    assert abs(dtags[0][1] - cfo) < 2e-3                                 # coarse CFO from the lag-16 autocorrelation (rad/sample)
    sync_length = 4 * (N + CP)
    fs = oracle.FrameSync(N, CP, sync_length, ofdm64["l_ltf_fir"])
    delayed = np.concatenate([np.zeros(sync_length, np.complex64), seg])[:seg.size]     # blocks_delay(sync_length)
    out, otags = fs.run(seg, delayed, dtags)
    assert len(otags) == 1 and otags[0][0] == 0 and fs.frame_start < sync_length
    residual = otags[0][1]                                               # coarse - fine (:186): the detector already de-rotated
    assert abs(residual - cfo) < 3e-3 and abs(fs.freq_offset) < 3e-3     # by the coarse estimate, so the fine one is ~0
    sym_f = np.fft.fftshift(np.fft.fft(out[:(out.size // N) * N].reshape(-1, N), axis=1), axes=1) / np.sqrt(N)
    _, eq = make_blocks(ofdm64)
    r = eq.general_work(sym_f.astype(np.complex64), [(0, residual)])
    starts = [e for e in r["events"] if e["kind"] == 1]
    assert starts and starts[0]["mcs"] == mcs and starts[0]["data_bytes"] == tags["pdu_len"]
    ok, got = oracle.stream_decode(mcs, 48, tags["pdu_len"], r["out"])
    assert ok and got == payload


def test_detector_needs_min_peaks_and_ignores_isolated_spikes():
    n = 4000
    ic = np.zeros(n, np.float32)
    ic[100] = 0.9                                                        # one spike: n_peaks = 1, then the distance rule resets it
    ic[1000:1009] = 0.9                                                  # 9 < min_n_peaks + 1 samples above threshold
    ic[2000:2030] = 5.0                                                  # above MAX_PEAK_VALUE: not a peak (:95)
    det = oracle.FrameDetector(N, CP, 0.6, 10, 640)
    x = np.ones(n, np.complex64)
    out, tags = det.run(x, x, ic)
    assert out.size == 0 and tags == []
    ic[3000:3011] = 0.9                                                  # the 11th sample in a row triggers (:106-118)
    det = oracle.FrameDetector(N, CP, 0.6, 10, 640)
    out, tags = det.run(x, x, ic)
    assert tags == [(0, 0.0)] and out.size == n - 3010
