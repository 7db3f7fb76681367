"""jrc_sincosf_fast (csrc/jrc_internal.h: the sine / cosine of the sync front end's per-sample de-rotations) modelled in numpy from the constants
in the header itself: Cody-Waite reduction by pi/2 in two fused steps, Cephes' single-precision polynomials, quadrant from the multiple.  A fused
multiply-add of floats is exact in float64 before the final rounding (24 x 24 bit product, one rounding to 53 bits, one to 24: the double
rounding moves a result by at most one float ulp in ~2^-29 of the cases), so the model follows the device arithmetic to that.  Bound claimed in
the header: 1.5 ulp / 9.3e-8 absolute for |angle| < 2^15."""
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gr-mimo-ofdm-jrc_amd", "csrc", "jrc_internal.h")
f32 = np.float32


def constants():
    src = open(HDR).read()
    body = src[src.index("void jrc_sincosf_fast"):]
    body = body[:body.index("\n}\n")]
    nums = [float(x.rstrip("f")) for x in re.findall(r"-?\d+\.\d*(?:e[-+]?\d+)?f", body)]
    return body, nums


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def model(a, k):
    limit, two_over_pi, c1, c2, s3, s2, s1, q3, q2, q1, half, one = [np.full_like(a, v) for v in k]
    n = np.rint(a * two_over_pi).astype(f32)
    r = fma(n, c1, a)
    r = fma(n, c2, r)
    z = r * r
    s = fma(r * z, fma(z, fma(z, s3, s2), s1), r)
    c = fma(z * z, fma(z, fma(z, q3, q2), q1), fma(z, half, one))
    q = n.astype(np.int64)
    ss, cc = np.where(q & 1, c, s), np.where(q & 1, s, c)
    return np.where(q & 2, -ss, ss), np.where((q + 1) & 2, -cc, cc)


def test_fast_sincos_model_meets_the_bound_stated_in_the_header():
    body, k = constants()
    assert len(k) == 12 and k[0] == 32768.0 and abs(k[1] - 2 / np.pi) < 1e-7, k          # the order model() reads them in
    assert abs((-k[2]) + (-k[3]) - np.pi / 2) < 1e-14                                     # the two parts of pi/2
    assert "1.5 ulp / 9.3e-8" in open(HDR).read()
    rng = np.random.default_rng(0)
    for scale in (1.0, 10.0, 300.0, 3000.0, 32767.0):
        a = rng.uniform(-scale, scale, 1_000_000).astype(f32)
        s, c = model(a, k)
        rs, rc = np.sin(a.astype(np.float64)), np.cos(a.astype(np.float64))
        assert np.abs(s - rs).max() < 9.3e-8 and np.abs(c - rc).max() < 9.4e-8, scale
        big = np.abs(rs) > 0.1
        assert (np.abs(s[big] - rs[big]) / np.spacing(np.abs(rs[big]).astype(f32))).max() < 1.6, scale
    edge = np.array([0.0, -0.0, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi, 1e-30, 32767.99], f32)
    s, c = model(edge, k)
    assert np.abs(s - np.sin(edge.astype(np.float64))).max() < 1e-7 and np.abs(c - np.cos(edge.astype(np.float64))).max() < 1e-7
