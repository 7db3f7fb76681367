"""GPU tier: target_simulator on the device through the C ABI against the oracle restatement of
lib/target_simulator_impl.cc:132-385 on the same bursts.  Tolerance: the north star's 1e-4 on ||a-b||_inf/||b||_inf
(the device evaluates the two length-n DFTs as float32 chirp-z transforms, the oracle in double)."""
import numpy as np
import pytest

import oracle
from conftest import crandn, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
FS, FC = 125_000_000, 24e9
POS4 = [0.0, 0.00625, 0.0125, 0.01875]
TGT3 = ([10.0, 23.5, 41.0], [0.0, 12.0, -30.0], [100.0, 10.0, 31.0], [20.0, -35.0, 5.0])


def burst(n, seed):
    return crandn(np.random.default_rng(seed), n)


@pytest.mark.parametrize("n", [1, 2, 7, 80, 1920, 5000, 16384, 16385, 22080])
def test_single_target_any_length(jrc, ctx, n):
    args = ([10.0], [7.5], [100.0], [20.0], POS4, FS, FC)
    x = burst(n, n)
    got = jrc.target_simulator(*args, ctx=ctx).work(x)
    want = oracle.TargetSimulator(*args).work(x)
    assert got.shape == want.shape == (4, n)
    assert rel_err(got, want) < TOL


@pytest.mark.parametrize("n", [40000, 70000, 174080, 300000, 600000])
def test_long_bursts(jrc, ctx, n):
    """chirp-z lengths M = 2^17 .. 2^21: row pass as m x 256 two-step transforms (m = 2..16) and the generic Stockham rows"""
    args = ([35.0], [-12.0], [40.0], [-25.0], POS4[:2], FS, FC)
    x = burst(n, n)
    got = jrc.target_simulator(*args, ctx=ctx).work(x)
    want = oracle.TargetSimulator(*args).work(x)
    assert rel_err(got, want) < TOL


def test_burst_too_long_is_refused(jrc, ctx):
    g = jrc.target_simulator([10.0], [0.0], [1.0], [0.0], [0.0], FS, FC, ctx=ctx)
    with pytest.raises(jrc.JrcError):
        g.work(np.zeros((1 << 20) + 1, np.complex64))


@pytest.mark.parametrize("sum_targets", [False, True])
def test_three_targets(jrc, ctx, sum_targets):
    n = 24 * 80
    x = burst(n, 5)
    got = jrc.target_simulator(*TGT3, POS4, FS, FC, sum_targets=sum_targets, ctx=ctx).work(x)
    want = oracle.TargetSimulator(*TGT3, POS4, FS, FC).work(x, sum_targets=sum_targets)
    assert rel_err(got, want) < TOL
    if not sum_targets:          # as written in the reference the last target overwrites the others
        last = jrc.target_simulator(*[v[-1:] for v in TGT3], POS4, FS, FC, ctx=ctx).work(x)
        np.testing.assert_array_equal(got, last)


def test_random_phase_and_self_coupling(jrc, ctx):
    n = 3000
    x = burst(n, 9)
    g = jrc.target_simulator(*TGT3, POS4[:2], FS, FC, self_coupling_db=-20.0, rndm_phaseshift=True, self_coupling=True,
                             sum_targets=True, seed=4, ctx=ctx)
    ph = g.draw_phases()
    assert ph.shape == (3,) and np.allclose(np.abs(ph), 1.0, atol=1e-6)
    got = g.work(x, target_phase=ph)
    o = oracle.TargetSimulator(*TGT3, POS4[:2], FS, FC, self_coupling_db=-20.0, rndm_phaseshift=True, self_coupling=True)
    want = o.work(x, target_phase=ph, sum_targets=True)
    assert rel_err(got, want) < TOL
    # phases are drawn internally when none are handed in
    assert g.work(x).shape == (2, n)


def test_no_targets_and_empty_burst(jrc, ctx):
    x = burst(100, 1)
    g = jrc.target_simulator([], [], [], [], POS4[:2], FS, FC, self_coupling_db=-6.0, self_coupling=True, ctx=ctx)
    want = oracle.TargetSimulator([], [], [], [], POS4[:2], FS, FC, self_coupling_db=-6.0, self_coupling=True).work(x)
    np.testing.assert_array_equal(g.work(x), want)
    assert g.work(np.zeros(0, np.complex64)).shape == (2, 0)
    with pytest.raises(ValueError):
        jrc.target_simulator([1.0], [], [1.0], [0.0], POS4, FS, FC, ctx=ctx)


def test_burst_length_change_and_new_targets(jrc, ctx):
    """:215-247 / :121-198 — filters follow the burst length and setup_targets()"""
    g = jrc.target_simulator([10.0], [0.0], [100.0], [0.0], POS4, FS, FC, ctx=ctx)
    o = oracle.TargetSimulator([10.0], [0.0], [100.0], [0.0], POS4, FS, FC)
    for n in (640, 1280, 640):
        x = burst(n, n)
        assert rel_err(g.work(x), o.work(x)) < TOL
    g.setup_targets([30.0, 12.0], [3.0, -3.0], [10.0, 20.0], [-40.0, 40.0])
    o2 = oracle.TargetSimulator([30.0, 12.0], [3.0, -3.0], [10.0, 20.0], [-40.0, 40.0], POS4, FS, FC)
    x = burst(640, 2)
    assert rel_err(g.work(x), o2.work(x)) < TOL
    assert g.rx_time_tag()[0] == (640 + 1280 + 640 + 640) // FS


def test_batched_device_form_accumulates_tx_simulators(jrc, ctx):
    """T simulators (one per TX, different virtual array positions) summed into the same RX buffers: the
    blocks_add_xx of the simulation flowgraph absorbed by accumulate_out"""
    import torch
    n, B, T = 2000, 3, 2
    xs = [np.stack([burst(n, 10 * t + b) for b in range(B)]) for t in range(T)]
    pos = [POS4, [p + 0.025 for p in POS4]]
    d_out = torch.zeros((B, 4, n), dtype=torch.complex64, device="cuda")
    want = np.zeros((B, 4, n), np.complex128)
    sims = []
    for t in range(T):
        g = jrc.target_simulator(*TGT3, pos[t], FS, FC, sum_targets=True, max_bursts=B, ctx=ctx)
        sims.append(g)
        d_in = torch.from_numpy(xs[t]).cuda()
        g.run_dev(d_in, d_out, B, n, accumulate_out=(t > 0))
        ctx.sync()
        o = oracle.TargetSimulator(*TGT3, pos[t], FS, FC)
        for b in range(B):
            want[b] += o.work(xs[t][b], sum_targets=True)
    assert rel_err(d_out.cpu().numpy(), want) < TOL
    with pytest.raises(ValueError):
        sims[0].run_dev(d_in, d_out, B + 1, n)


def test_synthetic_target_is_recovered_by_the_radar_chain(jrc, ctx):
    """simulator -> CP removal + FFT -> mimo_ofdm_radar -> range FFT: the peak sits at the simulated range"""
    N, cp, S = 64, 16, 16
    rng = np.random.default_rng(0)
    tx = ((rng.integers(0, 2, (S, N)) * 2 - 1) + 1j * (rng.integers(0, 2, (S, N)) * 2 - 1)).astype(np.complex64) / np.sqrt(2)
    td = jrc.ofdm_mod(tx, N, cp, ctx=ctx).ravel()
    R_true = 12.0
    rx = jrc.target_simulator([R_true], [0.0], [100.0], [0.0], [0.0], FS, FC, ctx=ctx).work(td)[0]
    rxf = jrc.ofdm_cyclic_prefix_remover(N, cp, ctx=ctx).work(rx, fused_fft=True).reshape(S, N)
    H = (rxf * np.conj(tx)).sum(0)
    prof = np.abs(np.fft.ifft(np.fft.ifftshift(H), 8 * N))
    r_axis = np.arange(8 * N) * 3e8 / (2 * FS * 8)
    assert abs(r_axis[int(prof.argmax())] - R_true) < 3e8 / (2 * FS) / 2
