"""CPU tier: the target_simulator restatement (oracle/jrc_oracle_tsim.c, reference lib/target_simulator_impl.cc:132-385)
against an independent numpy restatement and against closed forms (integer-sample delay, Doppler tone, radar equation)."""
import numpy as np
import pytest

import oracle
from conftest import crandn as _crandn, rel_err


def crandn(n, seed=0):
    return _crandn(np.random.default_rng(seed), n)

C0 = np.float32(3e8)


def np_tsim(x, rng, vel, rcs, az, pos, fs, fc, sum_targets=False, phase=None, self_coupling_db=None):
    """numpy restatement, float64 FFTs; same float32 roundings of the scalar parameters as the reference"""
    x = np.asarray(x, np.complex64)
    n = x.size
    rng, vel, rcs, az, pos = (np.atleast_1d(np.asarray(v, np.float32)) for v in (rng, vel, rcs, az, pos))
    fc = np.float32(fc)
    dop = (np.float32(2) * vel * fc / C0).astype(np.float32)
    amp = (C0 * np.sqrt(rcs)).astype(np.float64) / 44.54662397465366 / (rng * rng).astype(np.float64) / np.float64(fc)
    amp = amp.astype(np.float32)
    i = np.arange(n)
    freq = (i.astype(np.float32) * np.float32(fs) / np.float32(n)).astype(np.float32)
    freq[n // 2:] = freq[n // 2:] - np.float32(fs)
    out = np.zeros((pos.size, n), np.complex128)
    for l in range(pos.size):
        for k in range(rng.size):
            # doppler phase integrates in float32 (gr_complex) one sample at a time
            ph = np.zeros(n, np.float32)
            step = 2 * np.pi * np.float64(dop[k]) / np.float64(np.float32(fs))
            cur = np.float32(0)
            for j in range(n):
                ph[j] = cur
                cur = np.float32(np.fmod(np.float64(cur) + step, 2 * np.pi))
            fd = np.exp(1j * ph.astype(np.float64)) * np.float64(amp[k])
            ts = np.float32((2.0 * np.float64(rng[k]) - np.float64(pos[l]) * np.sin(np.float64(az[k]) * np.pi / 180.0)) / np.float64(C0))
            fsum = (freq + fc).astype(np.float32)
            pt = np.fmod(2 * np.pi * np.float64(ts) * fsum.astype(np.float64), 2 * np.pi).astype(np.float32)
            ft = np.exp(-1j * pt.astype(np.float64)) / n
            y = np.fft.ifft(np.fft.fft(x.astype(np.complex128) * fd) * ft) * n
            if phase is not None:
                y = y * phase[k]
            out[l] = out[l] + y if (sum_targets and k > 0) else y
        if self_coupling_db is not None:
            out[l] += np.float32(10 ** (self_coupling_db / 20.0)) * x
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 16, 45, 96, 345, 1000, 2310])
def test_dft_any_matches_numpy(n):
    x = crandn(n, seed=n)
    assert rel_err(oracle.dft_any(x, True), np.fft.fft(x.astype(np.complex128))) < 3e-7
    assert rel_err(oracle.dft_any(x, False), np.fft.ifft(x.astype(np.complex128)) * n) < 3e-7


@pytest.mark.parametrize("n", [240, 1920])
@pytest.mark.parametrize("sum_targets", [False, True])
def test_tsim_matches_numpy_restatement(n, sum_targets):
    rng, vel, rcs, az = [10.0, 23.5, 41.0], [0.0, 12.0, -30.0], [100.0, 10.0, 31.0], [20.0, -35.0, 5.0]
    pos = [0.0, 0.00625, 0.0125, 0.01875]
    fs, fc = 125_000_000, 24e9
    x = crandn(n, seed=7)
    sim = oracle.TargetSimulator(rng, vel, rcs, az, pos, fs, fc)
    got = sim.work(x, sum_targets=sum_targets)
    want = np_tsim(x, rng, vel, rcs, az, pos, fs, fc, sum_targets=sum_targets)
    assert got.shape == (4, n)
    assert rel_err(got, want) < 2e-6
    if not sum_targets:      # as written in the reference every target overwrites the output: the last one is what is left
        last = oracle.TargetSimulator(rng[-1:], vel[-1:], rcs[-1:], az[-1:], pos, fs, fc).work(x)
        np.testing.assert_array_equal(got, last)


def test_filters_closed_form():
    fs, fc, n = 125_000_000, 24e9, 480
    sim = oracle.TargetSimulator([15.0], [25.0], [50.0], [10.0], [0.0, 0.0125], fs, fc)
    fd = sim.filt_doppler(n, 0)
    amp = 3e8 * np.sqrt(50.0) / 44.54662397465366 / 15.0 ** 2 / 24e9          # radar equation, :187
    assert abs(abs(fd[0]) - amp) / amp < 1e-6 and fd[0].imag == 0
    dop = 2 * 25.0 * 24e9 / 3e8
    want = amp * np.exp(2j * np.pi * dop * np.arange(n) / fs)
    assert rel_err(fd, want) < 5e-5                                            # float32 phase integration drifts
    for l, pos in enumerate([0.0, 0.0125]):
        ft = sim.filt_time(n, l, 0)
        assert np.allclose(np.abs(ft), 1.0 / n, rtol=1e-6)
    # the two antennas differ by the steering phase 2*pi*fc*d*sin(az)/c at DC (up to the float32 rounding of ts: fc*eps*ts ~ 1e-2 rad)
    d = np.angle(sim.filt_time(n, 1, 0)[0] / sim.filt_time(n, 0, 0)[0])
    want_d = 2 * np.pi * 24e9 * 0.0125 * np.sin(np.deg2rad(10.0)) / 3e8
    assert abs(np.angle(np.exp(1j * (d - want_d)))) < 0.15


def test_integer_delay_is_a_circular_shift():
    """range chosen so that 2R/c is a whole number of samples and fc*tau is whole: output = amp * roll(x, delay)"""
    fs, n = 100_000_000, 400
    delay = 7
    R = delay * 3e8 / fs / 2          # 10.5 m
    fc = 1e9                          # fc * tau = 70 cycles exactly
    x = crandn(n, seed=3)
    sim = oracle.TargetSimulator([R], [0.0], [10.0], [0.0], [0.0], fs, fc)
    y = sim.work(x)[0]
    amp = 3e8 * np.sqrt(10.0) / 44.54662397465366 / R ** 2 / fc
    assert rel_err(y, amp * np.roll(x, delay)) < 2e-4


def test_self_coupling_and_random_phase():
    fs, fc, n = 125_000_000, 24e9, 320
    x = crandn(n, seed=11)
    args = ([12.0, 30.0], [5.0, -5.0], [20.0, 40.0], [-10.0, 25.0], [0.0, 0.00625], fs, fc)
    ph = np.exp(2j * np.pi * np.array([0.123, 0.877])).astype(np.complex64)
    sim = oracle.TargetSimulator(*args, self_coupling_db=-40.0, rndm_phaseshift=True, self_coupling=True)
    got = sim.work(x, target_phase=ph, sum_targets=True)
    want = np_tsim(x, *args, sum_targets=True, phase=ph, self_coupling_db=-40.0)
    assert rel_err(got, want) < 2e-6
    assert sim.work(np.zeros(0, np.complex64)).shape == (2, 0)
