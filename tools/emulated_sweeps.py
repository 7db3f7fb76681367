#!/usr/bin/env python3
"""Exhaustive small-case sweeps that only make sense without launch latency: run with JRC_EMULATE=1 (the kernels on the CPU emulation, tests/hipcpu).
Every fft_vcc size 1..700 (+ the sizes around powers of two up to 4096) in all four direction / shift combinations against numpy; every
matrix_transpose shape up to 24 x 24 x interp 1..3 against the oracle; every (fft_len, cp_len) of the prefix remover up to 40 / 12; the estimator's
peak in every angle bin x 7 range rows x 3 discard settings (records byte for byte); fft_peak_detect's peak at every bin x 3 thresholds.
Prints one line per sweep; exit status 1 if anything disagrees."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert os.environ.get("JRC_EMULATE"), "run with JRC_EMULATE=1"
import conftest  # noqa: F401,E402  (points the package at the emulated library)
import numpy as np  # noqa: E402

import jrc_amd as jrc  # noqa: E402
import oracle  # noqa: E402


def main():
    ctx = jrc.Context(0)
    rng = np.random.default_rng(2026)
    bad = 0
    sizes = list(range(1, 701)) + [n + d for n in (1024, 2048, 4096) for d in (-3, -1, 0, 1) if 1 <= n + d <= 4096] + [1536, 3000, 3072, 4095]
    worst = 0.0
    for n in sizes:
        for fwd in (True, False):
            for shift in (False, True):
                b = 1 + (n % 3)
                x = (rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))).astype(np.complex64)
                w = None if n % 5 else (0.5 + rng.random(n)).astype(np.float32)
                got = jrc.fft_vcc(n, fwd, window=w, shift=shift, ctx=ctx).work(x)
                xin = x.astype(np.complex128)
                if w is not None:
                    xin = xin * w
                if fwd:
                    ref = np.fft.fft(xin, axis=-1)
                    if shift:
                        ref = np.fft.fftshift(ref, axes=-1)
                else:
                    if shift:
                        xin = np.fft.ifftshift(xin, axes=-1)
                    ref = np.fft.ifft(xin, axis=-1) * n
                err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
                worst = max(worst, err)
                if not err < 1e-4:                      # north_star tolerance; the tests hold the stages to 5e-6 on their own shapes
                    bad += 1
                    print("fft_vcc n=%d forward=%s shift=%s: %g" % (n, fwd, shift, err))
    print("fft_vcc: %d sizes x 4 forms, worst relative error %.3g, %d outside 1e-4" % (len(sizes), worst, bad))
    nbad = 0
    cases = 0
    for P in range(1, 25):
        for L in list(range(1, 25)) + [64, 100]:
            for interp in (1, 2, 3):
                x = (rng.standard_normal((P, L)) + 1j * rng.standard_normal((P, L))).astype(np.complex64)
                got = jrc.matrix_transpose(L, P, interp, ctx=ctx).work(x)
                want = oracle.matrix_transpose(x, L, P, interp)
                cases += 1
                if got.shape != want.shape or not np.array_equal(got, want):
                    nbad += 1
                    print("matrix_transpose P=%d L=%d interp=%d differs" % (P, L, interp))
    print("matrix_transpose: %d shapes, %d differ" % (cases, nbad))
    bad += nbad
    nbad = cases = 0
    for N in range(1, 41):
        for cp in range(0, min(N, 12) + 1):
            for k in (1, 2, 5):
                x = (rng.standard_normal(k * (N + cp) + (N % 3)) + 0j).astype(np.complex64)     # a ragged tail the block must not read as a symbol
                got = jrc.ofdm_cyclic_prefix_remover(N, cp, ctx=ctx).work(x)
                want = oracle.cp_remove(x, N, cp)
                cases += 1
                if got.shape != want.shape or not np.array_equal(got, want):
                    nbad += 1
                    print("cp_remove N=%d cp=%d k=%d differs" % (N, cp, k))
    print("ofdm_cyclic_prefix_remover: %d shapes, %d differ" % (cases, nbad))
    bad += nbad
    # range_angle_estimator: the peak in every angle bin x a set of range rows (window wrap on both axes), three discard settings, records byte for byte
    import ctypes
    rb, ab = jrc.radar_axes(64, 125e6, 8, 8, 16)
    nbad = cases = 0
    base = (0.05 * (rng.standard_normal((512, 128)) + 1j * rng.standard_normal((512, 128)))).astype(np.complex64)
    for ndr, nda in ((2.4, 28.96), (9.0, 14.0), (0.3, 60.0)):
        est = jrc.range_angle_estimator(128, rb, ab, ndr, nda, 15.0, 0.0, ctx=ctx)
        for row in (0, 1, 255, 256, 257, 510, 511):
            for col in range(128):
                m = base.copy()
                m[row, col] = 3.0 + 0.5j
                g = est.work(m)
                o = oracle.ra_estimate(m, rb, ab, ndr, nda, 15.0, 0.0)
                cases += 1
                if ctypes.string_at(ctypes.byref(g), ctypes.sizeof(g)) != ctypes.string_at(ctypes.byref(o), ctypes.sizeof(o)):
                    nbad += 1
                    if nbad < 10:
                        print("range_angle_estimator row=%d col=%d discard=(%g, %g) differs" % (row, col, ndr, nda))
    print("range_angle_estimator: %d maps (every angle bin x 7 range rows x 3 discard settings), %d records differ" % (cases, nbad))
    bad += nbad
    # fft_peak_detect: the peak at every position of a 257-bin spectrum (both halves of the frequency formula, the protected edges), three thresholds
    nbad = cases = 0
    for thr in (-20.0, 3.0, 40.0):
        det = jrc.fft_peak_detect(1000000, 4.0, thr, 5, ctx=ctx)
        for pk in range(257):
            x = (0.01 * (rng.standard_normal(257) + 1j * rng.standard_normal(257))).astype(np.complex64)
            x[pk] = 2 * np.exp(0.3j * pk)
            cases += 1
            if not np.array_equal(np.array(det.work(x), np.float64), np.array(oracle.fft_peak_detect(x, 1000000, 4.0, thr, 5), np.float64), equal_nan=True):     # nothing over the threshold: (-1, nan, nan, nan) on both sides
                nbad += 1
                print("fft_peak_detect peak at %d thr %g differs" % (pk, thr))
    print("fft_peak_detect: %d spectra, %d differ" % (cases, nbad))
    bad += nbad
    # stream_encoder / stream_decoder: every PDU length 1..160 at every MCS — symbols bit-exact against the oracle, decoded bytes and CRC verdict back
    nbad = cases = lost = 0
    for mcs in range(6):
        enc = jrc.stream_encoder(mcs, 48, ctx=ctx)
        dec = jrc.stream_decoder(48, ctx=ctx)
        for nbytes in range(1, 161):
            pdu = bytes([2]) + rng.integers(0, 256, nbytes - 1, dtype=np.uint8).tobytes()
            seed = enc.d_scrambler
            got, tags = enc.work(pdu)
            want, wtags = oracle.stream_encode(mcs, 48, pdu, seed)
            ok = np.array_equal(got, want) and tags == wtags
            # the decoder against the ORACLE's decoder on the same clean symbols (not against "must decode": the reference's windowed Viterbi loses
            # some clean frames — BPSK 3/4 at 6 and 42 bytes, about one payload in seven — and the device must lose exactly those)
            d = dec.work(got, dict(mcs=mcs, data_bytes=tags["pdu_len"], packet_type=2, snr=20.0))
            o = oracle.stream_decode(mcs, 48, tags["pdu_len"], want)
            ok = ok and d == o
            lost += d != (True, pdu)
            cases += 1
            if not ok:
                nbad += 1
                print("codec mcs=%d nbytes=%d differs" % (mcs, nbytes))
    print("stream_encoder -> stream_decoder: %d (MCS, PDU length) pairs, %d differ from the oracle (%d clean frames lost by both decoders alike)" % (cases, nbad, lost))
    bad += nbad
    # precoder -> flat channel -> equalizer at the .grc's carriers: every PDU length 1..40 x every MCS x LS / STA x NDP / DATA, device blocks against the
    # oracle's blocks on the same symbols (precoder: sync words + SIG exact, the rest 1e-6; equalizer: 2e-5, events equal)
    from test_oracle_comm import qam16, qpsk, through_channel
    from conftest import crandn, rel_err
    o64 = np.load(os.path.join(ROOT, "tests", "golden", "ofdm_config_64.npz"))
    dc, pc, ps, sw, ml, ltf = (o64[k] for k in ("data_subcarriers", "pilot_subcarriers", "pilot_symbols", "l_stf_ltf_64", "ltf_mapped_sc__ss_sym", "ltf_64"))
    nbad = cases = 0
    worst_p = worst_e = 0.0
    for est in (0, 1):
        gp = jrc.mimo_precoder(64, 4, 1, dc, pc, ps, sw, ml, ctx=ctx)
        op = oracle.Precoder(64, 4, 1, dc, pc, ps, sw, ml)
        for ptype in (1, 2):
            for mcs in range(6):
                if est == 1 and mcs in (4, 5):
                    continue                      # STA with 16-QAM decides through gr-digital's table (recollected: covered by its own test, not swept)
                for nbytes in range(1, 41):
                    ge = jrc.mimo_ofdm_equalizer(est, 24e9, 125e6, 64, 16, dc, pc, ps, ltf, ml, 4, ctx=ctx)
                    oe = oracle.Equalizer(est, 24e9, 125e6, 64, 16, dc, pc, ps, ltf, ml, 4)
                    ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
                    sy = qam16(rng, ns * 48) if mcs >= 4 else qpsk(rng, ns * 48) if mcs >= 2 else (rng.integers(0, 2, ns * 48) * 2 - 1).astype(np.complex64)
                    tg, tr = gp.work(sy, mcs, ptype, nbytes), op.work(sy, mcs, ptype, nbytes)
                    ep = rel_err(tg, tr)
                    ok = tg.shape == tr.shape and ep < 1e-6 and np.array_equal(tg[:, :5], tr[:, :5])
                    y = through_channel(tr, crandn(rng, 4), 2e-3, rng)
                    g, o = ge.general_work(y, [(0, 0.013)]), oe.general_work(y, [(0, 0.013)])
                    ee = rel_err(g["out"], o["out"]) if g["out"].shape == o["out"].shape and g["out"].size else (0.0 if g["out"].shape == o["out"].shape else 1.0)
                    ok = ok and g["consumed"] == o["consumed"] and ee < 2e-5 and len(g["events"]) == len(o["events"]) and \
                        all(a["kind"] == b["kind"] and a["offset"] == b["offset"] for a, b in zip(g["events"], o["events"]))
                    worst_p, worst_e = max(worst_p, ep), max(worst_e, ee)
                    cases += 1
                    if not ok:
                        nbad += 1
                        if nbad < 12:
                            print("comm est=%d ptype=%d mcs=%d nbytes=%d: precoder %g equalizer %g" % (est, ptype, mcs, nbytes, ep, ee))
    print("precoder -> channel -> equalizer: %d (estimator, packet type, MCS, PDU length) cases, %d outside tolerance (worst precoder %.2g, equalizer %.2g)" % (cases, nbad, worst_p, worst_e))
    bad += nbad
    # target_simulator: every burst length n1 x 2^a with n1 = 1..80 and 2^a = 16..256 (the direct route's column lengths incl. every prime to 79), and the
    # lengths just off them (chirp-z route), two RX antennas, two summed targets: default route, and the burst-on-chip kernel where it takes the burst
    nbad = cases = 0
    worst = {"default": 0.0, "onchip": 0.0}
    args = ([12.0, 33.0], [3.0, -20.0], [50.0, 20.0], [15.0, -40.0], [0.0, 0.00625], 125_000_000, 24e9)
    ref = oracle.TargetSimulator(*args)
    lengths = sorted(set([n1 * p2 for n1 in range(1, 81) for p2 in (16, 64, 256)] + [n1 * 32 + 1 for n1 in range(1, 40, 3)]))
    for n in lengths:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        want = ref.work(x, sum_targets=True)
        for route in ("default", "onchip"):
            os.environ["JRC_TSIM_ONCHIP"] = "1" if route == "onchip" else "0"
            got = jrc.target_simulator(*args, sum_targets=True, ctx=ctx).work(x)
            err = rel_err(got, want)
            worst[route] = max(worst[route], err)
            cases += 1
            if not err < 1e-4:
                nbad += 1
                if nbad < 12:
                    print("target_simulator n=%d route=%s: %g" % (n, route, err))
    os.environ.pop("JRC_TSIM_ONCHIP", None)
    print("target_simulator: %d burst lengths x 2 routes, %d outside 1e-4 (worst: default route %.2g, on-chip switch on %.2g)" % (len(lengths), nbad, worst["default"], worst["onchip"]))
    bad += nbad
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
