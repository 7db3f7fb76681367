"""CPU tier: the oracle (test infrastructure) against closed forms, numpy and the golden constant tables.
The reference ships no tests or vectors and cannot be built here, so these closed-form checks plus the
tables minted from its embedded Python module are the only pins available ("oracle")."""
import ctypes

import numpy as np
import pytest

import oracle
from conftest import crandn, rel_err

f32 = np.float32


def radar_py(tx, rx, N, T, R, S, Npre, interleave=False, discard=0):
    """independent float32 restatement of lib/mimo_ofdm_radar_impl.cc:250-274 (pure Python loops, tiny sizes)"""
    H = np.zeros((T * R, N), np.complex64)
    for sc in range(N):
        for r in range(R):
            for t in range(T):
                ar, ai = f32(0), f32(0)
                for s in range(S):
                    a, b = f32(rx[r][Npre + s, sc].real), f32(rx[r][Npre + s, sc].imag)
                    c, d = f32(tx[t][Npre + discard + s, sc].real), f32(tx[t][Npre + discard + s, sc].imag)
                    ar = f32(ar + f32(f32(a * c) + f32(b * d)))
                    ai = f32(ai + f32(f32(b * c) - f32(a * d)))
                p = t * R + r if interleave else r * T + t
                H[p, sc] = complex(ar, ai)
    return H


@pytest.mark.parametrize("T,R,interleave,discard", [(2, 2, False, 0), (4, 2, True, 0), (1, 1, False, 3), (3, 2, False, 1)])
def test_radar_matches_python_restatement_bit_exact(T, R, interleave, discard):
    rng = np.random.default_rng(1)
    N, S, Npre, Ir = 16, 5, 2, 4
    tx = [crandn(rng, Npre + discard + S, N) for _ in range(T)]
    rx = [crandn(rng, Npre + S, N) for _ in range(R)]
    rad = oracle.Radar(N, T, R, S, Npre, interp_factor=Ir, enable_tx_interleave=interleave)
    out = rad.work(tx, rx, tx_discard=discard)
    H = radar_py(tx, rx, N, T, R, S, Npre, interleave, discard)
    assert out.shape == (T * R, N * Ir)
    assert np.array_equal(out[:, :N], H)
    assert np.all(out[:, N:] == 0)             # zero padded rows (:243, :312-315)


def test_radar_closed_form_f64():
    rng = np.random.default_rng(2)
    N, T, R, S, Npre = 64, 4, 2, 4, 5
    tx = [crandn(rng, Npre + S, N) for _ in range(T)]
    rx = [crandn(rng, Npre + S, N) for _ in range(R)]
    out = oracle.Radar(N, T, R, S, Npre, interp_factor=1).work(tx, rx)
    txa = np.array(tx, np.complex128)[:, Npre:]
    rxa = np.array(rx, np.complex128)[:, Npre:]
    H = np.einsum("rsn,tsn->rtn", rxa, np.conj(txa)).reshape(T * R, N)
    assert rel_err(out, H) < 1e-6


def test_radar_orthogonal_ltf_separates_tx(ofdm64):
    """with the reference's P_ltf the matched filter over N_sym = N_tx LTF symbols returns N_ltf*|ltf|^2*h per pair"""
    P, ltf = ofdm64["P_ltf"].real, ofdm64["ltf_64"].real
    T, R, N = 4, 2, 64
    h = np.random.default_rng(3).standard_normal((R, T)) + 0j
    tx = [np.array([P[t, l] * ltf for l in range(4)], np.complex64) for t in range(T)]
    rx = [sum(h[r, t] * tx[t] for t in range(T)).astype(np.complex64) for r in range(R)]
    out = oracle.Radar(N, T, R, 4, 0).work(tx, rx)
    for r in range(R):
        for t in range(T):
            exp = 4 * h[r, t] * ltf * ltf
            assert np.allclose(out[r * T + t], exp, atol=1e-5)


def ring_py(frames_H, record_len, recording_flags):
    """background ring semantics (lib/mimo_ofdm_radar_impl.cc:276-300) on precomputed raw estimates"""
    ring, temp, outs = [], np.zeros_like(frames_H[0]), []
    for H, rec in zip(frames_H, recording_flags):
        if rec:
            temp = H.copy()
        n = len(ring)
        mr = np.zeros(H.shape, f32)
        mi = np.zeros(H.shape, f32)
        for e in ring:
            mr = (mr + (e.real / f32(n)).astype(f32)).astype(f32)
            mi = (mi + (e.imag / f32(n)).astype(f32)).astype(f32)
        outs.append(((H.real - mr).astype(f32) + 1j * (H.imag - mi).astype(f32)).astype(np.complex64))
        ring.append(temp.copy())
        if len(ring) > record_len:
            ring.pop(0)
    return outs


def test_radar_background_ring_over_frames():
    rng = np.random.default_rng(4)
    N, T, R, S, Npre, L = 16, 2, 2, 3, 1, 3
    rad = oracle.Radar(N, T, R, S, Npre, background_removal=True, background_recording=True, record_len=L)
    raw = oracle.Radar(N, T, R, S, Npre)
    flags = [True, True, False, True, True, True]
    Hs, outs = [], []
    for rec in flags:
        tx = [crandn(rng, Npre + S, N) for _ in range(T)]
        rx = [crandn(rng, Npre + S, N) for _ in range(R)]
        rad.set_background_record(rec)
        outs.append(rad.work(tx, rx))
        Hs.append(raw.work(tx, rx))
    exp = ring_py(Hs, L, flags)
    for o, e in zip(outs, exp):
        assert np.array_equal(o, e)
    assert rad.ring_size() == L


@pytest.mark.parametrize("n", [2, 8, 64, 256, 2048])
@pytest.mark.parametrize("forward,shift", [(True, False), (True, True), (False, False), (False, True)])
def test_fft_vcc_semantics_vs_numpy(n, forward, shift):
    rng = np.random.default_rng(n)
    x = crandn(rng, 3, n)
    y = oracle.fft_vcc(x, forward, shift)
    xd = x.astype(np.complex128)
    if forward:
        ref = np.fft.fft(xd, axis=-1)
        if shift:
            ref = np.fft.fftshift(ref, axes=-1)
    else:
        ref = n * np.fft.ifft(np.fft.ifftshift(xd, axes=-1) if shift else xd, axis=-1)
    assert rel_err(y, ref) < 1e-6
    assert rel_err(oracle.fft_vcc(x, forward, shift, f32=True), ref) < 2e-5


def test_fft_vcc_window_and_non_pow2():
    rng = np.random.default_rng(5)
    n = 64
    x = crandn(rng, 2, n)
    w = np.full(n, 1 / np.sqrt(64), f32)      # the TX IFFT window of the flowgraphs
    y = oracle.fft_vcc(x, False, True, window=w)
    ref = n * np.fft.ifft(np.fft.ifftshift(x.astype(np.complex128) * w, axes=-1), axis=-1)
    assert rel_err(y, ref) < 1e-6
    x = crandn(rng, 1, 40)
    assert rel_err(oracle.fft_vcc(x, True, True), np.fft.fftshift(np.fft.fft(x.astype(np.complex128)))) < 1e-6


def test_matrix_transpose():
    rng = np.random.default_rng(6)
    P, L, Ia = 8, 48, 4
    x = crandn(rng, P, L)
    y = oracle.matrix_transpose(x, L, P, Ia)
    assert y.shape == (L, P * Ia)
    assert np.array_equal(y[:, :P], x.T) and np.all(y[:, P:] == 0)
    with pytest.raises(RuntimeError):      # lib/matrix_transpose_impl.cc:82-83
        oracle.matrix_transpose(crandn(rng, 3, 10), 10, 4, 1)


def test_cp_remove():
    rng = np.random.default_rng(7)
    N, cp, k = 16, 4, 5
    x = crandn(rng, k * (N + cp) + 7)      # ragged tail is ignored (:86)
    y = oracle.cp_remove(x, N, cp)
    assert y.shape == (k, N)
    for i in range(k):
        assert np.array_equal(y[i], x[i * (N + cp) + cp:(i + 1) * (N + cp)])


def test_hypotf_is_double_sqrt_rounded_once():
    """the HIP kernels restate glibc's hypotf as (float)sqrt((double)x*x+(double)y*y): pin that on this image"""
    libm = ctypes.CDLL("libm.so.6")
    libm.hypotf.restype = ctypes.c_float
    libm.hypotf.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(8)
    xs = (rng.standard_normal(20000) * 10.0 ** rng.integers(-6, 3, 20000)).astype(f32)
    ys = (rng.standard_normal(20000) * 10.0 ** rng.integers(-6, 3, 20000)).astype(f32)
    exp = np.sqrt(xs.astype(np.float64) ** 2 + ys.astype(np.float64) ** 2).astype(f32)
    got = np.array([libm.hypotf(float(a), float(b)) for a, b in zip(xs, ys)], f32)
    assert np.array_equal(got, exp)


def test_golden_tables_shape_the_generator(ofdm64):
    import jrc_amd
    from jrc_amd import synth
    assert np.array_equal(synth.hadamard(4), ofdm64["P_ltf"].real)
    P, ltf, m = ofdm64["P_ltf"], ofdm64["ltf_64"], ofdm64["ltf_mapped_sc__ss_sym"]
    for sc in range(64):
        assert np.array_equal(m[sc].reshape(4, 4), P * ltf[sc])     # [sc][t*N_ltf + l] = P[t][l]*ltf[sc]
    assert np.allclose(ofdm64["P_ltf"] @ ofdm64["P_ltf"].conj().T, 4 * np.eye(4))
    assert list(ofdm64["pilot_subcarriers"]) == [-21, -7, 7, 21] and len(ofdm64["data_subcarriers"]) == 48


@pytest.mark.parametrize("P", [8, 16])
def test_sampling_bound_of_the_detect_only_angle_stage(P):
    """chain.hip (range_angle_wide_kernel, MODE 1, `refine`): the angle axis of one range bin is f(theta) = sum_p R[p] e^{-j p theta}; its
    cells r = 0, Ia/4, Ia/2, 3 Ia/4 (mod Ia) are 4 P equispaced samples, and the kernel skips a row when
    max_samples |f|^2 / cos^2(pi (P-1) / (8 P)) is below the running maximum (Ehlich-Zeller inequality on e^{j(P-1)phi} f(2 phi)).
    Numerical check of the constant it compiles in: random, sparse and optimised coefficient vectors never exceed it, and the
    margin to the worst case found is small enough that a looser constant would be noticed."""
    from scipy.optimize import minimize
    const = {16: 1.14880, 8: 1.12803}[P]
    exact = 1.0 / np.cos(np.pi * (P - 1) / (8.0 * P)) ** 2
    assert exact <= const < exact * (1 + 1e-4)
    rng = np.random.default_rng(P)

    def ratio(a):
        if not np.any(a):
            return 0.0
        return float((np.abs(np.fft.fft(a, 16384)) ** 2).max() / (np.abs(np.fft.fft(a, 4 * P)) ** 2).max())

    worst = 0.0
    for trial in range(1500):
        a = rng.standard_normal(P) + 1j * rng.standard_normal(P)
        if trial % 3 == 0:
            a = a * (rng.random(P) < 0.3)
        worst = max(worst, ratio(a))
    for start in range(12):
        r = minimize(lambda x: -ratio(x[:P] + 1j * x[P:]), rng.standard_normal(2 * P), method="Nelder-Mead",
                     options=dict(maxiter=3000, xatol=1e-6, fatol=1e-9))
        worst = max(worst, -r.fun)
    assert 1.05 < worst <= const
