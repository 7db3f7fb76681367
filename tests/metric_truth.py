"""The comm graph's normalised detection metric from its DEFINITION, in float64 (test infrastructure).

The flowgraph forms it from stock blocks (examples/simulation/communication/mimo_ofdm_jrc_comm_sim.grc: blocks_delay(fft_len/4) ->
blocks_conjugate_cc -> blocks_multiply_xx -> moving_avg(window) -> complex_to_mag, over complex_to_mag_squared ->
blocks_moving_average_ff(int(1.5 window), 1/1.5, max_iter 16000 :598-608) -> abs -> blocks_divide_ff):

    corr[i]  = sum_{k = i-window+1 .. i}  x[k] conj(x[k - delay])          (samples before the stream are zero)
    power[i] = pscale * sum_{k = i-pwindow+1 .. i} |x[k]|^2
    metric[i] = |corr[i]| / |power[i]|

The stock moving averages carry a RUNNING float sum (add the entering sample, subtract the leaving one) that is started afresh at
every scheduler call, at most max_iter outputs apart: their round-off depends on where the scheduler happened to cut the stream, so
the reference's output on this edge is not a function of the input alone.  The oracle restates the worst case (one call, one running
sum over the whole capture); the device adds each window afresh.  Both are held against this definition: every window summed on its
own in float64 from the float32 samples, no history carried from one output to the next."""
import numpy as np


def window_sums(v, window, chunk=1 << 16):
    """s[i] = sum of v[i-window+1 .. i] with zeros in front, each window added on its own (no running sum), float64 / complex128"""
    v = np.asarray(v)
    acc = np.complex128 if np.iscomplexobj(v) else np.float64
    p = np.concatenate([np.zeros(window - 1, acc), v.astype(acc)])
    out = np.empty(v.size, acc)
    for i0 in range(0, v.size, chunk):
        i1 = min(v.size, i0 + chunk)
        out[i0:i1] = np.lib.stride_tricks.sliding_window_view(p[i0:i1 + window - 1], window).sum(axis=-1)
    return out


def metric_truth(x, delay, window, pwindow, pscale):
    """(corr, power, metric) of the definition above, float64"""
    x = np.asarray(x, np.complex64).astype(np.complex128)
    xd = np.concatenate([np.zeros(delay, np.complex128), x])[:x.size]
    corr = window_sums(x * np.conj(xd), window)
    power = float(pscale) * window_sums(x.real ** 2 + x.imag ** 2, pwindow)
    with np.errstate(divide="ignore", invalid="ignore"):
        metric = np.abs(corr) / np.abs(power)
    return corr, power, metric


def metric_errors(dev, ora, x_dev, x_ora, delay, window, pwindow, pscale, live):
    """max |metric - truth| / max(1, |truth|_inf) behind the first `live` samples (which divide by a near-empty power window) for the device's
    metric and for the oracle's running-sum metric, each against the definition on ITS OWN input samples; and device against oracle"""
    tg = metric_truth(x_dev, delay, window, pwindow, pscale)[2][live:]
    to = tg if x_ora is x_dev else metric_truth(x_ora, delay, window, pwindow, pscale)[2][live:]
    a, b = np.asarray(dev, np.float64)[live:], np.asarray(ora, np.float64)[live:]
    sg, so = max(1.0, float(tg.max())), max(1.0, float(to.max()))
    return dict(dev=float(np.abs(a - tg).max() / sg), ora=float(np.abs(b - to).max() / so), dev_vs_ora=float(np.abs(a - b).max() / so))


FLOAT_FLOOR = 2e-6      # a few float32 ulps of a metric of order 1: below this "closer to the truth" is the toss of a rounding


def assert_metric_parity(dev, ora, x_dev, x_ora, delay, window, pwindow, pscale, live, tol=1e-4):
    """north_star's 1e-4 for the device against the definition; the device no further from the definition than the oracle's running sum; the
    device against the oracle within 1e-4 plus the running sum's own measured distance from the definition (the drift, named: it is `ora`)"""
    e = metric_errors(dev, ora, x_dev, x_ora, delay, window, pwindow, pscale, live)
    assert e["dev"] <= tol, e
    assert e["dev"] <= max(e["ora"], FLOAT_FLOOR), e
    assert e["dev_vs_ora"] <= tol + e["ora"], e
    return e
