"""GPU tier: target_simulator on the device through the C ABI against the oracle restatement of
lib/target_simulator_impl.cc:132-385 on the same bursts.  Tolerance: the north star's 1e-4 on ||a-b||_inf/||b||_inf
(the device evaluates the two length-n DFTs in float32 — as a direct four-step split n = n1 x 2^a where the length allows it, as chirp-z
transforms otherwise — the oracle in double)."""
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
    """long bursts: chirp-z lengths M = 2^17 .. 2^21 (row pass as m x 256 two-step transforms, m = 2..16, and the generic Stockham rows) for the
    lengths outside the direct route; 174080 = 85 x 2048 (config D's burst) takes the direct route here and the chirp-z one in
    test_direct_four_step_lengths"""
    args = ([35.0], [-12.0], [40.0], [-25.0], POS4[:2], FS, FC)
    x = burst(n, n)
    got = jrc.target_simulator(*args, ctx=ctx).work(x)
    want = oracle.TargetSimulator(*args).work(x)
    assert rel_err(got, want) < TOL


# burst lengths of the direct four-step route (tsim.hip td_*): n = n1 x n2, n2 the power of two in n (16 .. 4096), n1 = the rest (<= 512)
DIRECT = [
    (256, 1, 256), (4096, 1, 4096), (16, 1, 16),                      # n1 = 1: the row pass alone
    (80, 5, 16), (2400, 75, 32), (3 * 64 * 7, 21, 64), (1920, 15, 128),   # 64-carrier flowgraph bursts: rows through the LDS Stockham kernel
    (11520, 45, 256), (23040, 45, 512), (9216, 9, 1024),              # config B's burst (72 symbols x 320): 45 x 512
    (174080, 85, 2048), (28672, 7, 4096), (1 << 20, 256, 4096),       # config D's burst (136 x 1280): 85 = 5 x 17 (a 17-term pass); 2^20 = 256 x 4096
    (22080, 345, 64), (64 * 509, 509, 64), (32 * 512, 4, 4096), (512 * 121, 121, 512), (6 * 256 * 13, 39, 512),   # 23 / 509 / 11 / 13 as radices
]


@pytest.mark.parametrize("n,n1,n2", DIRECT)
def test_direct_four_step_lengths(jrc, ctx, monkeypatch, n, n1, n2):
    """every split the direct route takes — all three row kernels, every row length, column lengths that are 1, smooth, prime or carry a large
    prime — against the oracle (1e-4), and against the chirp-z route on the same burst (JRC_TSIM_BLUESTEIN=1): two independent evaluations of
    the same two DFTs"""
    assert n == n1 * n2
    args = ([35.0], [-12.0], [40.0], [-25.0], POS4 if n <= 30000 else POS4[:2], FS, FC)
    x = burst(n, n)
    got = jrc.target_simulator(*args, ctx=ctx).work(x)
    want = oracle.TargetSimulator(*args).work(x)
    assert got.shape == want.shape
    assert rel_err(got, want) < 2e-6, rel_err(got, want)               # measured <= 8e-7: the direct split rounds less than the chirp-z route
    monkeypatch.setenv("JRC_TSIM_BLUESTEIN", "1")
    old = jrc.target_simulator(*args, ctx=ctx).work(x)
    monkeypatch.delenv("JRC_TSIM_BLUESTEIN")
    assert rel_err(old, want) < TOL and rel_err(got, old) < TOL


@pytest.mark.parametrize("n", [23040, 2400, 11520, 174080])
@pytest.mark.parametrize("R", [1, 2, 3, 4])
def test_direct_route_targets_phases_coupling_accumulate(jrc, ctx, n, R):
    """the direct route through every option of the block: three targets summed or last-only, random phases, self coupling, bursts batched on the
    device and accumulated into RX buffers that already hold another simulator's output — for 1..4 RX antennas (antenna groups of 4 / 2 / 1)"""
    import torch
    pos = POS4[:R]
    x = np.stack([burst(n, 3 * n + b) for b in range(2)])
    for sum_targets in (False, True):
        g = jrc.target_simulator(*TGT3, pos, FS, FC, self_coupling_db=-20.0, rndm_phaseshift=True, self_coupling=True, sum_targets=sum_targets,
                                 seed=4, max_bursts=2, ctx=ctx)
        ph = g.draw_phases()
        o = oracle.TargetSimulator(*TGT3, pos, FS, FC, self_coupling_db=-20.0, rndm_phaseshift=True, self_coupling=True)
        want = np.stack([o.work(x[b], target_phase=ph, sum_targets=sum_targets) for b in range(2)])
        assert rel_err(g.work(x[0], target_phase=ph), want[0]) < 2e-6
        base = (np.arange(2 * R * n, dtype=np.float32).reshape(2, R, n) % 7).astype(np.complex64)
        d_out = torch.from_numpy(base.copy()).cuda()
        g.run_dev(torch.from_numpy(x).cuda(), d_out, 2, n, accumulate_out=True, target_phase=ph)
        ctx.sync()
        assert rel_err(d_out.cpu().numpy() - base, want) < 2e-5         # (the sum with the buffer's 0..6 rounds at 6 x 2^-24)


def test_direct_route_on_buffers_that_are_only_8_byte_aligned(jrc, ctx):
    """the column passes move two samples (16 bytes) per lane; a caller's buffers need only be gr_complex-aligned: input and output one sample off
    a 16-byte boundary give the same bytes as aligned ones"""
    import torch
    n, B, R = 11520, 2, 3
    x = np.stack([burst(n, 7 + b) for b in range(B)])
    g = jrc.target_simulator(*TGT3, POS4[:R], FS, FC, sum_targets=True, max_bursts=B, ctx=ctx)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((B, R, n), dtype=torch.complex64, device="cuda")
    g.run_dev(d_in, d_out, B, n)
    ctx.sync()
    flat_in = torch.zeros(B * n + 1, dtype=torch.complex64, device="cuda")
    flat_out = torch.zeros(B * R * n + 1, dtype=torch.complex64, device="cuda")
    flat_in[1:] = d_in.reshape(-1)
    assert flat_in[1:].data_ptr() % 16 == 8 and flat_out[1:].data_ptr() % 16 == 8
    g.run_dev(flat_in[1:], flat_out[1:], B, n)
    ctx.sync()
    assert torch.equal(flat_out[1:].reshape(B, R, n), d_out) and flat_out[0].item() == 0


@pytest.mark.parametrize("n", [23040, 174080])
def test_closed_forms_at_the_baseline_burst_sizes(jrc, ctx, n):
    """properties that need no second implementation, at config B's (72 x 320 = 45 x 512) and config D's (136 x 1280 = 85 x 2048) burst length:
    a target whose delay is a whole number of samples and whose carrier phase is a whole number of turns gives amp x the burst rolled by that
    delay (lib/target_simulator_impl.cc:291-305: the time-shift filter is then exp(-j 2 pi k delay / n) / n exactly); a Doppler-only target
    multiplies the burst by a tone; and the block is linear in its input"""
    fs, fc, delay = 100_000_000, 1e9, 7
    R = delay * 3e8 / fs / 2                                        # 10.5 m: 2R/c = 7 samples, fc x tau = 70 turns
    rng = np.random.default_rng(n)
    x, y2 = crandn(rng, n), crandn(rng, n)
    sim = jrc.target_simulator([R], [0.0], [10.0], [0.0], [0.0, 0.0], fs, fc, ctx=ctx)
    amp = 3e8 * np.sqrt(10.0) / 44.54662397465366 / R ** 2 / fc
    out = sim.work(x)
    assert rel_err(out[0], amp * np.roll(x, delay)) < 2e-4 and np.array_equal(out[0], out[1])      # (the reference's float filter phases: 2e-4, as in the CPU tier)
    a, b = np.complex64(0.7 - 0.2j), np.complex64(-1.1 + 0.4j)
    lin = sim.work((a * x + b * y2).astype(np.complex64))
    assert rel_err(lin, a * out + b * sim.work(y2)) < 1e-5
    v = 15.0
    dop = jrc.target_simulator([R], [v], [10.0], [0.0], [0.0], fs, fc, ctx=ctx).work(x)[0]
    tone = np.exp(2j * np.pi * (2 * v * fc / 3e8) / fs * np.arange(n))
    # doppler first (:345), then the delay.  The reference advances the Doppler phase by a FLOAT recurrence (:281-287: fmod of a float sum, n steps), which
    # the device's filter table follows step for step; against the exact tone that recurrence has drifted 4e-4 after 23,040 steps and 1.5e-3 after 174,080
    assert rel_err(dop, amp * np.roll(x * tone, delay)) < (5e-4 if n < 30000 else 3e-3)


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


@pytest.mark.parametrize("n", [2000, 11520, 23040])
@pytest.mark.parametrize("sum_targets", [False, True])
def test_simulators_summed_on_the_spectrum(jrc, ctx, n, sum_targets):
    """jrc_tsim_run_sum_dev: the simulators of the flowgraph's TX ports (same targets, other antenna positions) and the blocks_add_xx behind
    them as one pass — against the oracle's sum, against the simulators run one by one with accumulate_out (equal to the rounding of the sum's
    order), on top of a buffer that already holds something, with random phases and self coupling; and refused for a length outside the
    direct route"""
    import torch
    B, T, R = 3, 3, 4
    xs = [np.stack([burst(n, 100 * t + b + n) for b in range(B)]) for t in range(T)]
    pos = [[p + 0.025 * t for p in POS4] for t in range(T)]
    kw = dict(self_coupling_db=-25.0, rndm_phaseshift=True, self_coupling=True, sum_targets=sum_targets, max_bursts=B, ctx=ctx)
    sims = [jrc.target_simulator(*TGT3, pos[t], FS, FC, seed=t, **kw) for t in range(T)]
    phases = [g.draw_phases() for g in sims]
    d_ins = [torch.from_numpy(x).cuda() for x in xs]
    want = np.zeros((B, R, n), np.complex128)
    for t in range(T):
        o = oracle.TargetSimulator(*TGT3, pos[t], FS, FC, self_coupling_db=-25.0, rndm_phaseshift=True, self_coupling=True)
        for b in range(B):
            want[b] += o.work(xs[t][b], target_phase=phases[t], sum_targets=sum_targets)
    d_out = torch.zeros((B, R, n), dtype=torch.complex64, device="cuda")
    jrc.target_simulator.run_sum_dev(sims, d_ins, d_out, B, n, target_phases=phases)
    ctx.sync()
    got = d_out.cpu().numpy()
    assert rel_err(got, want) < 2e-6
    d_one = torch.zeros_like(d_out)
    for t in range(T):
        sims[t].run_dev(d_ins[t], d_one, B, n, target_phase=phases[t], accumulate_out=(t > 0))
    ctx.sync()
    assert rel_err(got, d_one.cpu().numpy()) < 2e-6
    base = (np.arange(B * R * n, dtype=np.float32).reshape(B, R, n) % 5).astype(np.complex64)
    d_acc = torch.from_numpy(base.copy()).cuda()
    jrc.target_simulator.run_sum_dev(sims, d_ins, d_acc, B, n, target_phases=phases, accumulate_out=True)
    ctx.sync()
    assert rel_err(d_acc.cpu().numpy() - base, want) < 2e-5
    with pytest.raises(jrc.JrcError) as e:                           # 5000 = 8 x 625: the chirp-z route, which has no summed form
        jrc.target_simulator.run_sum_dev(sims, [d[:, :5000].contiguous() for d in d_ins], d_out, B, 5000)
    assert e.value.status == jrc.JRC_ERR_UNSUPPORTED
    with pytest.raises(ValueError):
        jrc.target_simulator.run_sum_dev(sims, d_ins, d_out, B + 1, n)
    other = jrc.target_simulator([10.0], [0.0], [5.0], [0.0], pos[0], FS, FC, seed=9, **kw)       # another target count: one launch cannot serve both
    with pytest.raises(jrc.JrcError) as e:
        jrc.target_simulator.run_sum_dev([sims[0], other], d_ins[:2], d_out, B, n)
    assert e.value.status == jrc.JRC_ERR_UNSUPPORTED


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


@pytest.mark.parametrize("n", [23040, 2400])
def test_more_summed_targets_than_the_direct_route_holds(jrc, ctx, n):
    """the direct four-step route carries at most 32 (simulator, target) pairs per launch and the number of targets of a simulator is
    unbounded (jrc_tsim_create / _set_targets): a simulator that SUMS 40 targets at a direct-route burst length must take the chirp-z route
    (any K) instead of failing; 32 summed targets and 40 targets with only the last observable (the reference's own behaviour: one pair) stay
    on the direct route.  All three against the oracle."""
    rng = np.random.default_rng(40)
    for K, sum_targets in ((40, True), (32, True), (40, False)):
        tg = (list(rng.uniform(5.0, 60.0, K)), list(rng.uniform(-40.0, 40.0, K)), list(rng.uniform(10.0, 100.0, K)), list(rng.uniform(-60.0, 60.0, K)))
        x = burst(n, n + K)
        g = jrc.target_simulator(*tg, POS4[:2], FS, FC, sum_targets=sum_targets, ctx=ctx)
        got = g.work(x)
        want = oracle.TargetSimulator(*tg, POS4[:2], FS, FC).work(x, sum_targets=sum_targets)
        assert got.shape == want.shape == (2, n)
        assert rel_err(got, want) < TOL, (K, sum_targets, rel_err(got, want))
