"""The committed fixture set tests/golden/path_fixtures_v2.npz (minted by tests/golden/make_path_fixtures.py from the CPU oracle, one
small case per row of SURVEY.md §8(c)'s list): the oracle must keep reproducing it (CPU tier), and the HIP path must match it
through the C ABI (GPU tier) — bit-exact for integer / index work and the A1 accumulation, 1e-4 for floating point."""
import os
import sys

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, rel_err

sys.path.insert(0, GOLDEN)
from make_path_fixtures import ra_fields, ra_map  # noqa: E402

TOL = 1e-4
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "path_fixtures_v2.npz"))


RA_INT = ("peak_range_idx", "peak_angle_idx", "angle_null_idx", "n_noise_samples", "published")
RA_FLT = ("peak_power", "noise_power", "snr_est", "range_val", "angle_val")


# ---------------------------------------------------------------- CPU tier: the oracle still produces the fixtures
def test_oracle_reproduces_radar_fft_transpose_cp(fx):
    tx, rx = fx["radar_tx"], fx["radar_rx"]
    for Ir in (1, 8):
        for il in (0, 1):
            rad = oracle.Radar(64, 4, 2, 4, 5, interp_factor=Ir, enable_tx_interleave=bool(il))
            assert np.array_equal(rad.work(list(tx[0]), list(rx[0])), fx["radar_out_Ir%d_il%d" % (Ir, il)])
    rad = oracle.Radar(64, 4, 2, 4, 5, background_removal=True, background_recording=True, record_len=2, interp_factor=1)
    assert np.array_equal(np.stack([rad.work(list(tx[f]), list(rx[f])) for f in range(3)]), fx["radar_bg_out"])
    x = fx["fft_in"]
    assert np.array_equal(oracle.fft_vcc(x, False, False), fx["fft_rev_512"])
    assert np.array_equal(oracle.fft_vcc(x, True, True), fx["fft_fwd_shift_512"])
    assert np.array_equal(oracle.fft_vcc(x[:, :45].copy(), True, True), fx["fft_fwd_shift_45"])
    assert np.array_equal(oracle.matrix_transpose(fx["transpose_in"], 16, 8, 2), fx["transpose_out"])
    assert np.array_equal(oracle.cp_remove(fx["cp_in"], 64, 16), fx["cp_out"])


def test_oracle_reproduces_estimator_peak_sig(fx):
    rb, ab = fx["ra_range_bins"], fx["ra_angle_bins"]
    for c, (kr, ka, amp) in enumerate(fx["ra_cases"]):
        i, f = ra_fields(oracle.ra_estimate(ra_map(512, 128, int(kr), int(ka), amp), rb, ab, 2.4, 14.0, 15.0, 0.0))
        assert np.array_equal(i, fx["ra_ints"][c]) and np.array_equal(f, fx["ra_floats"][c], equal_nan=True)
    for k in range(3):
        kk, f, p, m = oracle.fft_peak_detect(fx["peak_in"][k], 125000000, 8.0, -20.0 if k < 2 else 10.0, 10)
        assert np.array_equal(np.array([kk, f, p, m], np.float64), fx["peak_out"][k], equal_nan=True)
    kat = np.stack([oracle.sig_encode(48, mcs, pt, 100 + 7 * mcs) for mcs in range(6) for pt in (1, 2)])
    assert np.array_equal(kat, fx["sig_kat"])


def test_oracle_reproduces_comm_and_next_rows(fx, ofdm64):
    o = ofdm64
    args = (o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"])
    for tag, est, ptype in (("ndp_ls", 0, 1), ("data_ls", 0, 2), ("ndp_sta", 1, 1)):
        pre = oracle.Precoder(64, 4, 1, *args, o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"])
        assert np.array_equal(pre.work(fx["comm_%s_sym" % tag], 2, ptype, 45), fx["comm_%s_tx" % tag])
        eq = oracle.Equalizer(est, 24e9, 125e6, 64, 16, *args, o["ltf_64"], o["ltf_mapped_sc__ss_sym"], 4)
        r = eq.general_work(fx["comm_%s_rx" % tag], [(0, 0.011)])
        assert np.array_equal(r["out"], fx["comm_%s_eq" % tag])
    sim = oracle.TargetSimulator([10.0, 23.5], [0.0, 12.0], [100.0, 10.0], [20.0, -35.0], [0.0, 0.00625], 125000000, 24e9)
    assert np.array_equal(sim.work(fx["tsim_in"], sum_targets=True), fx["tsim_out"])
    pdu = fx["codec_pdu"].tobytes()
    for mcs in range(6):
        assert np.array_equal(oracle.stream_encode(mcs, 48, pdu, 1 + 20 * mcs)[0], fx["codec_sym_mcs%d" % mcs])
    ok, payload = oracle.stream_decode(3, 48, len(pdu) + 4, fx["codec_noisy_mcs3"])
    assert int(ok) == int(fx["codec_noisy_ok"][0]) and payload == fx["codec_noisy_payload"].tobytes()
    dout, dtags = oracle.FrameDetector(64, 16, 0.6, 10, 640).run(fx["fd_in"], fx["fd_in_abs"], fx["fd_in_cor"])
    assert np.array_equal(dout, fx["fd_out"]) and np.array_equal(np.array([[t[0], t[1]] for t in dtags], np.float64), fx["fd_tags"])


# ---------------------------------------------------------------- GPU tier: the HIP path against the committed data
@gpu
def test_hip_radar_fft_transpose_cp_match_fixtures(jrc, ctx, fx):
    tx, rx = fx["radar_tx"], fx["radar_rx"]
    for Ir in (1, 8):
        for il in (0, 1):
            blk = jrc.mimo_ofdm_radar(64, 4, 2, 4, 5, False, False, 8, Ir, bool(il), "", ctx=ctx)
            assert np.array_equal(blk.general_work(list(tx[0]), list(rx[0])), fx["radar_out_Ir%d_il%d" % (Ir, il)])      # A1 bit-exact
    blk = jrc.mimo_ofdm_radar(64, 4, 2, 4, 5, True, True, 2, 1, False, "", ctx=ctx)
    got = np.stack([blk.general_work(list(tx[f]), list(rx[f])) for f in range(3)])
    assert rel_err(got, fx["radar_bg_out"]) < 1e-6
    x = fx["fft_in"]
    assert rel_err(jrc.fft_vcc(512, False, None, False, ctx=ctx).work(x), fx["fft_rev_512"]) < TOL
    assert rel_err(jrc.fft_vcc(512, True, None, True, ctx=ctx).work(x), fx["fft_fwd_shift_512"]) < TOL
    assert rel_err(jrc.fft_vcc(45, True, None, True, ctx=ctx).work(x[:, :45].copy()), fx["fft_fwd_shift_45"]) < TOL
    assert rel_err(jrc.fft_vcc(96, False, None, True, ctx=ctx).work(x[:, :96].copy()), fx["fft_rev_shift_96"]) < TOL
    assert np.array_equal(jrc.matrix_transpose(16, 8, 2, ctx=ctx).work(fx["transpose_in"]), fx["transpose_out"])
    assert np.array_equal(jrc.ofdm_cyclic_prefix_remover(64, 16, ctx=ctx).work(fx["cp_in"]), fx["cp_out"])


@gpu
def test_hip_estimator_peak_sig_match_fixtures(jrc, ctx, fx):
    rb, ab = fx["ra_range_bins"], fx["ra_angle_bins"]
    est = jrc.range_angle_estimator(128, rb, ab, 2.4, 14.0, 15.0, 0.0, "", False, ctx=ctx)
    for c, (kr, ka, amp) in enumerate(fx["ra_cases"]):
        r = est.work(ra_map(512, 128, int(kr), int(ka), amp))
        assert [getattr(r, k) for k in RA_INT] == list(fx["ra_ints"][c])
        assert np.array_equal(np.array([getattr(r, k) for k in RA_FLT], np.float32), fx["ra_floats"][c], equal_nan=True)
    for k in range(3):
        d = jrc.fft_peak_detect(125000000, 8.0, -20.0 if k < 2 else 10.0, 10, ctx=ctx)
        kk, f, p, m = d.work(fx["peak_in"][k])
        want = fx["peak_out"][k]
        assert kk == int(want[0]) and np.allclose([f, p, m], want[1:], rtol=1e-6, equal_nan=True)
    kat = np.stack([jrc.sig_encode(48, mcs, pt, 100 + 7 * mcs) for mcs in range(6) for pt in (1, 2)])
    assert np.array_equal(kat, fx["sig_kat"])


@gpu
def test_hip_comm_and_next_rows_match_fixtures(jrc, ctx, fx, ofdm64):
    o = ofdm64
    args = (o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"])
    for tag, est, ptype in (("ndp_ls", 0, 1), ("data_ls", 0, 2), ("ndp_sta", 1, 1)):
        pre = jrc.mimo_precoder(64, 4, 1, *args, o["l_stf_ltf_64"], o["ltf_mapped_sc__ss_sym"], ctx=ctx)
        txf = pre.work(fx["comm_%s_sym" % tag], 2, ptype, 45)
        assert rel_err(txf, fx["comm_%s_tx" % tag]) < 1e-6 and np.array_equal(txf[:, :5], fx["comm_%s_tx" % tag][:, :5])
        eq = jrc.mimo_ofdm_equalizer(est, 24e9, 125e6, 64, 16, *args, o["ltf_64"], o["ltf_mapped_sc__ss_sym"], 4, ctx=ctx)
        r = eq.general_work(fx["comm_%s_rx" % tag], [(0, 0.011)])
        assert r["out"].shape == fx["comm_%s_eq" % tag].shape and rel_err(r["out"], fx["comm_%s_eq" % tag]) < TOL
        if "comm_%s_chan_est" % tag in fx.files:
            assert rel_err(r["chan_est"], fx["comm_%s_chan_est" % tag]) < TOL
    sim = jrc.target_simulator([10.0, 23.5], [0.0, 12.0], [100.0, 10.0], [20.0, -35.0], [0.0, 0.00625], 125000000, 24e9, sum_targets=True, ctx=ctx)
    assert rel_err(sim.work(fx["tsim_in"]), fx["tsim_out"]) < TOL
    pdu = fx["codec_pdu"].tobytes()
    for mcs in range(6):
        enc = jrc.stream_encoder(mcs, 48, ctx=ctx)
        enc.d_scrambler = 1 + 20 * mcs
        assert np.array_equal(enc.work(pdu)[0], fx["codec_sym_mcs%d" % mcs])                        # integer work: exact
    dec = jrc.stream_decoder(48, ctx=ctx)
    ok, payload = dec.work(fx["codec_noisy_mcs3"], dict(mcs=3, data_bytes=len(pdu) + 4))
    assert int(ok) == int(fx["codec_noisy_ok"][0]) and payload == fx["codec_noisy_payload"].tobytes()
    dout, dtags = jrc.frame_detector(64, 16, 0.6, 10, 640, ctx=ctx).run(fx["fd_in"], fx["fd_in_abs"], fx["fd_in_cor"])
    assert dout.shape == fx["fd_out"].shape and rel_err(dout, fx["fd_out"]) < TOL
    assert [t[0] for t in dtags] == list(fx["fd_tags"][:, 0].astype(int)) and np.allclose([t[1] for t in dtags], fx["fd_tags"][:, 1], atol=1e-6)


# ---------------------------------------------------------------- the reference flowgraph's own operating point (its Python expressions)
@pytest.fixture(scope="module")
def fg():
    """tests/golden/radar_flowgraph_point.npz: the numbers the reference's radar simulation flowgraph evaluates its block parameters to
    (minted by tests/golden/make_flowgraph_point_golden.py from the .grc's own Python expressions)"""
    return np.load(os.path.join(GOLDEN, "radar_flowgraph_point.npz"))


def test_host_side_axes_equal_the_flowgraphs_expressions(fg):
    """jrc_amd.radar_axes is what every test, the bench and the examples hand to the estimator: it must be the flowgraph's
    range_bins / angle_bins (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:1390-1400) bit for bit as float32"""
    import jrc_amd
    N, T, R, S, Npre, rec_len, Ir = (int(v) for v in fg["radar_ints"])
    Ia = int(fg["var_interp_factor_angle"])
    rb, ab = jrc_amd.radar_axes(N, float(fg["var_samp_rate"]), Ir, T * R, Ia)
    assert rb.dtype == np.float32 and np.array_equal(rb, fg["estimator_range_bins_f32"])
    assert ab.dtype == np.float32 and np.array_equal(ab, fg["estimator_angle_bins_f32"])
    assert np.array_equal(np.asarray(fg["var_angle_axis"], np.float32), ab)        # the GUI axis is the same expression
    vlen, ndr, nda, snr_thr, pow_thr = fg["estimator_scalars"]
    assert (int(vlen), snr_thr, pow_thr) == (T * R * Ia, 15.0, 0.0)
    assert ndr == 2 * 3e8 / (2 * float(fg["var_samp_rate"])) and nda == 2 * float(np.rad2deg(np.arcsin(2 / (T * R))))
    # the chain between the blocks: transpose geometry and the two stock FFTs (sizes, direction, shift, rectangular windows)
    assert list(fg["transpose_ints"]) == [N * Ir, T * R, Ia]
    assert list(fg["fft_range_size_forward_shift"]) == [N * Ir, 0, 0] and list(fg["fft_angle_size_forward_shift"]) == [T * R * Ia, 1, 1]
    assert np.all(fg["fft_range_window"] == 1.0) and np.all(fg["fft_angle_window"] == 1.0)
    assert list(fg["fft_rx_demod_size_forward_shift"]) == [N, 1, 1] and fg["fft_rx_demod_window"].size == 0
    assert list(fg["fft_tx_mod_size_forward_shift"]) == [N, 0, 1] and np.all(fg["fft_tx_mod_window"] == 1 / 64 ** 0.5)
    assert (Npre, S, rec_len) == (5, T, 8) and not fg["radar_flags"].any()
    assert list(fg["cp_remover_ints"]) == [N, N // 4] and int(fg["zero_pad_tail"]) == 3 * (N + N // 4)


def test_synthetic_scenario_follows_the_flowgraphs_channel_parameters(fg):
    """synth.py (the frames the bench and the parity tests consume) uses the flowgraph's noise variance, carrier frequency and virtual
    array: TXn_RXs of the .grc (:107-153) are element positions (1 + t/2 + 2 r) wavelengths = (r T + t + 2) half-wavelengths"""
    from jrc_amd import synth
    T, R = int(fg["var_N_tx"]), int(fg["var_N_rx"])
    sc = synth.Scenario(int(fg["var_fft_len"]), T, R, T, samp_rate=float(fg["var_samp_rate"]), center_freq=float(fg["var_rf_freq"]),
                        noise_figure_db=float(fg["var_noise_figure_dB"]))
    assert sc.noise_var == float(fg["var_noise_var"])
    assert sc.cp == int(fg["var_cp_len"]) and sc.R_max == float(fg["var_R_max"])
    lam = float(fg["var_wavelength"])
    assert lam == synth.C0 / sc.fc
    for t in range(T):
        pos = fg["var_TX%d_RXs" % (t + 1)]
        for r in range(R):
            assert abs(pos[r] - (r * T + t + 2) * lam / 2) < 1e-15
    assert float(fg["tsim_scalars"][0]) == 10 ** (float(fg["var_trgt_rcs_dbsm"]) / 10.0)


def test_comm_flowgraph_operating_point(fg, ofdm64):
    """the comm simulation flowgraph's sync front end as its own expressions evaluate (frame_detector threshold / peaks / ignore_gap,
    frame_sync length and matched-filter taps, moving-average window, padding): what examples/comm_sim_flowgraph.py and the sync tests use"""
    N, cp = (int(v) for v in fg["comm_frame_sync_ints"][:2])
    T = int(ofdm64["N_tx"])
    assert (N, cp) == (64, 16) and int(fg["comm_frame_sync_ints"][2]) == 4 * (N + cp)
    fft_len, cp_len, thr, peaks, gap = fg["comm_frame_detector"]
    assert (fft_len, cp_len, thr, peaks) == (N, cp, 0.6, 10) and gap == (len(ofdm64["l_stf_ltf_64"]) + T) * (N + cp)
    assert np.array_equal(fg["comm_frame_sync_ltf_fir"], np.asarray(ofdm64["l_ltf_fir"], np.complex64))
    assert list(fg["comm_moving_avg"]) == [N // 2, 1.0, 16000.0]
    assert list(fg["comm_zero_pad"]) == [5, 6 * (N + cp) + 10]
    freq, bw, n, c, nl = fg["comm_equalizer_scalars"]
    assert (freq, bw, n, c, nl) == (24e9, 125e6, N, cp, T)
    assert np.array_equal(fg["comm_equalizer_long_seq"], np.asarray(ofdm64["l_stf_ltf_64"][3], np.complex64))      # long_seq = sync word 3
    assert int(fg["comm_decoder_n_data_carriers"]) == int(fg["comm_encoder_data_len"]) == len(ofdm64["data_subcarriers"]) == 48
