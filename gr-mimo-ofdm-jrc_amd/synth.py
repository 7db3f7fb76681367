"""Seeded synthetic MIMO-OFDM radar frames — the bytes both the HIP path and the CPU
baseline consume.  Host-side numpy; it stands in for precoder -> channel -> RX demod of the radar
simulation flowgraph (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2165-2232) with the point-target
model of target_simulator (reference lib/target_simulator_impl.cc:164,177,188) summed over targets.
"""
import numpy as np

C0 = 3e8
SEED0 = 0x4A5243


def hadamard(n):
    """+-1 orthogonal P matrix; for n == 4 exactly the reference's P_ltf (ofdm_config in the .grc)"""
    if n == 4:
        return np.array([[1, -1, 1, 1], [1, 1, -1, 1], [1, 1, 1, -1], [-1, 1, 1, 1]], np.float64)
    h = np.array([[1.0]])
    while h.shape[0] < n:
        h = np.block([[h, h], [h, -h]])
    return h[:n, :n]


def ltf_sequence(n, rng):
    """length-n +-1 LTF with DC (index n/2) and the band edges nulled like the 64-carrier 802.11 LTF"""
    s = rng.choice([-1.0, 1.0], size=n)
    guard = max(1, n // 16)
    s[:guard] = 0
    s[n - guard + 1:] = 0
    s[n // 2] = 0
    return s


class Scenario:
    def __init__(self, fft_len, N_tx, N_rx, N_sym, N_pre=5, samp_rate=125e6, center_freq=24e9,
                 noise_figure_db=10.0, targets=None, n_random_targets=0):
        self.N, self.T, self.R, self.S, self.Npre = fft_len, N_tx, N_rx, N_sym, N_pre
        self.fs, self.fc = samp_rate, center_freq
        self.cp = fft_len // 4
        self.noise_var = 4.00388616e-21 * samp_rate * 10 ** (noise_figure_db / 10.0)   # .grc noise_var
        self.targets = targets            # list of (range_m, az_deg, vel_mps, rcs_m2) or None
        self.n_random = n_random_targets
        self.R_max = C0 * fft_len / (2 * samp_rate)


def config_B():
    return Scenario(256, 4, 4, 64, targets=[(10.0, 20.0, 0.0, 100.0)])


def config_D():
    return Scenario(1024, 4, 4, 128, n_random_targets=8)


def config_A():
    return Scenario(64, 1, 1, 16, targets=[(10.0, 0.0, 0.0, 100.0)])


def make_frames(sc, n_frames, first_frame=0, tx_gain=0.1):
    """returns complex64 [n_frames, T+R, Npre+S, N]: ports TX0..TX(T-1) (frequency-domain TX reference
    symbols) then RX0..RX(R-1) (received frequency-domain symbols), DC at index N/2."""
    N, T, R, S, Npre = sc.N, sc.T, sc.R, sc.S, sc.Npre
    n_items = Npre + S
    out = np.zeros((n_frames, T + R, n_items, N), np.complex64)
    lam = C0 / sc.fc
    f_sc = (np.arange(N) - N // 2) * sc.fs / N
    rng0 = np.random.default_rng(SEED0)
    ltf = ltf_sequence(N, rng0)
    Pm = hadamard(T)
    sym_t = np.arange(n_items) * (N + sc.cp) / sc.fs
    for fi in range(n_frames):
        rng = np.random.default_rng(SEED0 + first_frame + fi)
        tx = np.zeros((T, n_items, N), np.complex128)
        # preamble part the radar block skips: arbitrary but non-zero
        tx[:, :Npre, :] = (rng.choice([-1.0, 1.0], size=(T, Npre, N)) + 0j) * (ltf != 0)
        n_ltf = min(T, S)
        for t in range(T):
            for l in range(n_ltf):
                tx[t, Npre + l, :] = Pm[t, l] * ltf
        if S > n_ltf:
            q = rng.integers(0, 4, size=(T, S - n_ltf, N))
            pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2.0)   # QPSK / 2
            tx[:, Npre + n_ltf:, :] = pts[q] * (ltf != 0)
        if sc.targets is not None:
            targets = sc.targets
        else:
            targets = [(rng.uniform(5, 0.8 * sc.R_max), rng.uniform(-60, 60), rng.uniform(-40, 40),
                        rng.uniform(10, 100)) for _ in range(sc.n_random)]
        rx = np.zeros((R, n_items, N), np.complex128)
        for (rng_m, az, vel, rcs) in targets:
            amp = C0 * np.sqrt(rcs) / (4 * np.pi) ** 1.5 / rng_m ** 2 / sc.fc     # target_simulator_impl.cc:188
            fd = 2 * vel * sc.fc / C0                                              # :164
            dop = np.exp(2j * np.pi * fd * sym_t)[:, None]
            for r in range(R):
                for t in range(T):
                    pos = (r * T + t + 2) * lam / 2                                # uniform lambda/2 virtual array
                    tau = (2 * rng_m - pos * np.sin(np.deg2rad(az))) / C0          # :177
                    h = np.exp(-2j * np.pi * tau * (f_sc + sc.fc))[None, :]
                    rx[r] += amp * tx_gain * dop * h * tx[t]
        sig = np.sqrt(sc.noise_var / 2)
        rx += sig * (rng.standard_normal(rx.shape) + 1j * rng.standard_normal(rx.shape))
        out[fi, :T] = tx
        out[fi, T:] = rx
    return out
