"""Mints tests/golden/path_fixtures_v2.npz: seeded inputs and the outputs the CPU oracle produces for them, one small case per
row of SURVEY.md §8(c)'s fixture list (the reference ships no vectors of its own).  These are regression fixtures of the
restatement — they pin the oracle against silent change and give the GPU tier committed data to compare with; they are
not reference outputs (the reference cannot be built here, see oracle/jrc_oracle.h).

    python tests/golden/make_path_fixtures.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def crandn(rng, *shape, scale=1.0):
    return (scale * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))).astype(np.complex64)


def ra_map(Nr, Na, kr, ka, amp):
    """estimator input built from a closed form (not stored): a low deterministic floor plus one peak"""
    k, a = np.meshgrid(np.arange(Nr), np.arange(Na), indexing="ij")
    ph = ((k * 31 + a * 17 + k * a) % 97) * (2 * np.pi / 97)
    mp = (0.01 * (1 + ((k + 2 * a) % 5) / 5.0) * np.exp(1j * ph)).astype(np.complex64)
    mp[kr, ka] = np.complex64(amp * np.exp(0.3j))
    return mp


def ra_fields(r):
    return np.array([r.peak_range_idx, r.peak_angle_idx, r.angle_null_idx, r.n_noise_samples, r.published], np.int64), \
        np.array([r.peak_power, r.noise_power, r.snr_est, r.range_val, r.angle_val], np.float32)


def main():
    o = np.load(os.path.join(HERE, "ofdm_config_64.npz"))
    fx = {}
    rng = np.random.default_rng(20240601)

    # A1 mimo_ofdm_radar: T=4, R=2, N=64, S=4, Npre=5; Ir 1 and 8; interleave on/off; background ring over 3 frames
    T, R, N, S, Npre = 4, 2, 64, 4, 5
    tx = crandn(rng, 3, T, Npre + S, N)
    rx = crandn(rng, 3, R, Npre + S, N)
    fx["radar_tx"], fx["radar_rx"] = tx, rx
    for Ir in (1, 8):
        for il in (0, 1):
            rad = oracle.Radar(N, T, R, S, Npre, interp_factor=Ir, enable_tx_interleave=bool(il))
            fx["radar_out_Ir%d_il%d" % (Ir, il)] = rad.work(list(tx[0]), list(rx[0]))
    rad = oracle.Radar(N, T, R, S, Npre, background_removal=True, background_recording=True, record_len=2, interp_factor=1)
    fx["radar_bg_out"] = np.stack([rad.work(list(tx[f]), list(rx[f])) for f in range(3)])

    # stock fft_vxx stages (numpy-checked elsewhere): reverse N*Ir, forward+shift P*Ia, odd size
    x = crandn(rng, 3, 512)
    fx["fft_in"] = x
    fx["fft_rev_512"] = oracle.fft_vcc(x, False, False)
    fx["fft_fwd_shift_512"] = oracle.fft_vcc(x, True, True)
    fx["fft_fwd_shift_45"] = oracle.fft_vcc(x[:, :45].copy(), True, True)
    fx["fft_rev_shift_96"] = oracle.fft_vcc(x[:, :96].copy(), False, True)

    # A3 matrix_transpose 16 x 8 -> interp 2
    m = crandn(rng, 8, 16)
    fx["transpose_in"], fx["transpose_out"] = m, oracle.matrix_transpose(m, 16, 8, 2)

    # A5 range_angle_estimator: Nr=512, Na=128, peaks at bins 0 / 63 / 64 / 127, a wrap-around case, below threshold
    Nr, Na = 512, 128
    rb = np.linspace(0, 3e8 * 64 / (2 * 125e6), Nr).astype(np.float32)
    ab = (np.arcsin(2 / Na * (np.arange(Na) - Na // 2 + 0.5)) * 180 / np.pi).astype(np.float32)
    fx["ra_range_bins"], fx["ra_angle_bins"] = rb, ab
    cases = [(100, 0, 5.0), (100, 63, 5.0), (100, 64, 5.0), (100, 127, 5.0), (1, 70, 5.0), (510, 3, 5.0), (200, 40, 0.02)]
    ints, flts = [], []
    for (kr, ka, amp) in cases:
        r = oracle.ra_estimate(ra_map(Nr, Na, kr, ka, amp), rb, ab, 2.4, 14.0, 15.0, 0.0)
        i, f = ra_fields(r)
        ints.append(i); flts.append(f)
    fx["ra_cases"], fx["ra_ints"], fx["ra_floats"] = np.array(cases, np.float64), np.stack(ints), np.stack(flts)

    # A6 cp remover; B1 peak detect {bin <= n/2, > n/2, below threshold}
    s = crandn(rng, 5 * 80 + 3)
    fx["cp_in"], fx["cp_out"] = s, oracle.cp_remove(s, 64, 16)
    pk = []
    spec = np.stack([crandn(rng, 1024, scale=1e-3) for _ in range(3)])
    spec[0, 100] = 2.0; spec[1, 900] = 1.5 * np.exp(1j); spec[2, 300] = 1e-4
    for k in range(3):
        kk, f, p, mm = oracle.fft_peak_detect(spec[k], 125000000, 8.0, -20.0 if k < 2 else 10.0, 10)
        pk.append([kk, f, p, mm])
    fx["peak_in"], fx["peak_out"] = spec, np.array(pk, np.float64)

    # SIG KATs: 6 MCS x 2 packet types
    fx["sig_kat"] = np.stack([oracle.sig_encode(48, mcs, pt, 100 + 7 * mcs) for mcs in range(6) for pt in (1, 2)])

    # C2 precoder (DFT branch) -> flat channel -> C1 equalizer {NDP+LS, DATA+LS, NDP+STA}
    dc, pc = o["data_subcarriers"], o["pilot_subcarriers"]
    ps, sw, ml, ltf = o["pilot_symbols"], o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], o["ltf_64"]
    pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
    h = crandn(rng, 4)
    fx["comm_h"] = h
    for tag, est, ptype in (("ndp_ls", 0, 1), ("data_ls", 0, 2), ("ndp_sta", 1, 1)):
        pre = oracle.Precoder(64, 4, 1, dc, pc, ps, sw, ml)
        eq = oracle.Equalizer(est, 24e9, 125e6, 64, 16, dc, pc, ps, ltf, ml, 4)
        nbytes, mcs = 45, 2
        ns = oracle.n_ofdm_sym(mcs, 48, nbytes)
        sym = pts[rng.integers(0, 4, ns * 48)].astype(np.complex64)
        txf = pre.work(sym, mcs, ptype, nbytes)
        y = np.tensordot(h, txf, axes=(0, 0))
        y = np.concatenate([y[3:4], y[3:]], axis=0)
        y = (y + 2e-3 * (rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape))).astype(np.complex64)
        r = eq.general_work(y, [(0, 0.011)])
        fx["comm_%s_sym" % tag], fx["comm_%s_tx" % tag], fx["comm_%s_rx" % tag], fx["comm_%s_eq" % tag] = sym, txf, y, r["out"]
        if r["chan_est"] is not None:
            fx["comm_%s_chan_est" % tag] = r["chan_est"]

    # §8(f): target simulator, bit codec, sync front end
    burst = crandn(rng, 1920)
    sim = oracle.TargetSimulator([10.0, 23.5], [0.0, 12.0], [100.0, 10.0], [20.0, -35.0], [0.0, 0.00625], 125000000, 24e9)
    fx["tsim_in"], fx["tsim_out"] = burst, sim.work(burst, sum_targets=True)
    pdu = bytes([2]) + rng.integers(0, 256, 76, dtype=np.uint8).tobytes()
    fx["codec_pdu"] = np.frombuffer(pdu, np.uint8)
    for mcs in range(6):
        fx["codec_sym_mcs%d" % mcs] = oracle.stream_encode(mcs, 48, pdu, 1 + 20 * mcs)[0]
    noisy = fx["codec_sym_mcs3"] + crandn(rng, fx["codec_sym_mcs3"].size, scale=0.25)
    ok, payload = oracle.stream_decode(3, 48, len(pdu) + 4, noisy)
    fx["codec_noisy_mcs3"], fx["codec_noisy_ok"], fx["codec_noisy_payload"] = noisy, np.array([int(ok)]), np.frombuffer(payload, np.uint8)
    ic = np.zeros(3000, np.float32)
    ic[400:470] = 0.9; ic[1500:1511] = 0.8; ic[2000:2005] = 0.95
    ia = crandn(rng, 3000)
    xs = crandn(rng, 3000)
    det = oracle.FrameDetector(64, 16, 0.6, 10, 640)
    dout, dtags = det.run(xs, ia, ic)
    fx["fd_in"], fx["fd_in_abs"], fx["fd_in_cor"], fx["fd_out"] = xs, ia, ic, dout
    fx["fd_tags"] = np.array([[t[0], t[1]] for t in dtags], np.float64)

    path = os.path.join(HERE, "path_fixtures_v2.npz")
    np.savez_compressed(path, **fx)
    print("wrote %s: %d arrays, %.0f KiB" % (path, len(fx), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
