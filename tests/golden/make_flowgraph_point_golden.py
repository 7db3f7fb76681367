"""Mints tests/golden/radar_flowgraph_point.npz: the operating point of the reference's radar simulation flowgraph, i.e. the values its
OWN Python expressions evaluate to — the `variable` blocks of examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc and the parameter
expressions of the hot-path blocks (range_angle_estimator's range_bins / angle_bins / noise_discard_*, matrix_transpose's sizes,
mimo_ofdm_radar's N_pre / N_sym, the stock fft_vxx sizes and windows, target_simulator's antenna positions ...).  GRC evaluates these
strings with Python; so does this script, from where the flowgraph lies (needs /root/reference: build container only).  Only the
resulting numbers are committed; the build's host code (jrc_amd.radar_axes, examples/radar_sim_flowgraph.py, synth.py) is tested
against them in tests/test_golden_fixtures.py.

    python tests/golden/make_flowgraph_point_golden.py
"""
import cmath
import math
import os
import sys

import numpy as np
import yaml

GRC = "/root/reference/examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc"


class _Window:
    """gnuradio.fft.window.rectangular as the flowgraph uses it (a vector of ones); nothing else of gr-fft is referenced"""
    @staticmethod
    def rectangular(n):
        return [1.0] * int(n)


COMM_GRC = "/root/reference/examples/simulation/communication/mimo_ofdm_jrc_comm_sim.grc"


def evaluate(grc):
    """namespace of a flowgraph: its embedded module(s) executed, its variables evaluated; returns (namespace, blocks by name)"""
    d = yaml.safe_load(open(grc))
    ns = {"np": np, "numpy": np, "cmath": cmath, "math": math, "os": os, "window": _Window}
    for b in d["blocks"]:
        if b["id"] == "epy_module":
            mod = {}
            exec(compile(b["parameters"]["source_code"], grc + ":" + b["name"], "exec"), mod)
            ns[b["name"]] = type("module", (), mod)
    pending = {}
    for b in d["blocks"]:
        if b["id"] == "variable":
            pending[b["name"]] = b["parameters"]["value"]
        elif b["id"].startswith("variable_qtgui"):
            pending[b["name"]] = b["parameters"].get("value", b["parameters"].get("false", "0"))
    for _ in range(20):
        for k in list(pending):
            try:
                ns[k] = eval(str(pending[k]), ns)
                del pending[k]
            except (NameError, AttributeError):
                pass
    return ns, {b["name"]: b for b in d["blocks"]}


def comm_point(out):
    """the comm simulation flowgraph's sync front end and codec parameters (frame_detector, frame_sync, moving_avg, zero_pad, decoder)"""
    ns, blocks = evaluate(COMM_GRC)

    def par(block, key):
        return eval(str(blocks[block]["parameters"][key]), ns)

    fd, fs, ma, zp = "mimo_ofdm_jrc_frame_detector_0", "mimo_ofdm_jrc_frame_sync_0", "mimo_ofdm_jrc_moving_avg_0", "mimo_ofdm_jrc_zero_pad_0"
    out["comm_frame_detector"] = np.array([par(fd, "fft_len"), par(fd, "cp_len"), par(fd, "threshold"), par(fd, "min_n_peaks"), par(fd, "ignore_gap")], np.float64)
    out["comm_frame_sync_ints"] = np.array([par(fs, "fft_len"), par(fs, "cp_len"), par(fs, "sync_length")], np.int64)
    out["comm_frame_sync_ltf_fir"] = np.asarray(par(fs, "ltf_seq_time"), np.complex64)
    out["comm_moving_avg"] = np.array([par(ma, "length"), par(ma, "scale"), par(ma, "max_iter")], np.float64)
    out["comm_zero_pad"] = np.array([par(zp, "pad_front"), par(zp, "pad_tail")], np.int64)
    eq = "mimo_ofdm_jrc_mimo_ofdm_equalizer_0"
    out["comm_equalizer_scalars"] = np.array([par(eq, "freq"), par(eq, "bw"), par(eq, "fft_len"), par(eq, "cp_len"), par(eq, "n_mimo_ltf")], np.float64)
    out["comm_equalizer_long_seq"] = np.asarray(par(eq, "long_seq"), np.complex64)
    out["comm_decoder_n_data_carriers"] = np.int64(par("mimo_ofdm_jrc_stream_decoder_0", "n_data_carriers"))
    out["comm_encoder_data_len"] = np.int64(par("mimo_ofdm_jrc_stream_encoder_1", "data_len"))
    out["comm_noise_var_path_loss"] = np.array([ns["noise_var"], ns["path_loss"], ns["wavelength"], ns["distance"]], np.float64)


def main():
    d = yaml.safe_load(open(GRC))
    ns = {"np": np, "numpy": np, "cmath": cmath, "math": math, "os": os, "window": _Window}
    for b in d["blocks"]:                                     # the embedded ofdm_config module first
        if b["id"] == "epy_module":
            mod = {}
            exec(compile(b["parameters"]["source_code"], GRC + ":" + b["name"], "exec"), mod)
            ns[b["name"]] = type("module", (), mod)
    pending = {}
    for b in d["blocks"]:
        if b["id"] == "variable":
            pending[b["name"]] = b["parameters"]["value"]
        elif b["id"] in ("variable_qtgui_range", "variable_qtgui_check_box", "variable_qtgui_chooser", "variable_qtgui_push_button"):
            pending[b["name"]] = b["parameters"].get("value", b["parameters"].get("false", "0"))
    for _ in range(20):                                       # variables reference each other: evaluate until everything resolves
        for k in list(pending):
            try:
                ns[k] = eval(str(pending[k]), ns)
                del pending[k]
            except NameError:
                pass
        if not pending:
            break
    assert not pending, pending
    out = {}
    for k in ("fft_len", "cp_len", "samp_rate", "freq", "rf_freq", "wavelength", "R_max", "R_res", "angle_res", "interp_factor_range",
              "interp_factor_angle", "N_tx", "N_rx", "N_ltf", "noise_var", "TX1_RXs", "TX2_RXs", "TX3_RXs", "TX4_RXs", "angle_axis",
              "trgt_range", "trgt_angle", "trgt_velocity", "trgt_rcs_dbsm", "noise_figure_dB", "data_carriers_64", "pilot_carriers_64"):
        out["var_" + k] = np.asarray(ns[k], dtype=np.float64)
    blocks = {b["name"]: b for b in d["blocks"]}

    def par(block, key):
        return eval(str(blocks[block]["parameters"][key]), ns)

    est = "mimo_ofdm_jrc_range_angle_estimator_0"
    # std::vector<float> parameters: what reaches the C++ constructor is the float32 rounding of the Python doubles
    out["estimator_range_bins_f32"] = np.asarray(par(est, "range_bins"), np.float64).astype(np.float32)
    out["estimator_angle_bins_f32"] = np.asarray(par(est, "angle_bins"), np.float64).astype(np.float32)
    out["estimator_scalars"] = np.array([par(est, "vlen"), par(est, "noise_discard_range"), par(est, "noise_discard_angle"),
                                         par(est, "snr_threshold"), par(est, "power_threshold")], np.float64)
    rad = "mimo_ofdm_jrc_mimo_ofdm_radar_0"
    out["radar_ints"] = np.array([par(rad, k) for k in ("fft_len", "N_tx", "N_rx", "N_sym", "N_pre", "record_len", "interp_factor")], np.int64)
    out["radar_flags"] = np.array([bool(par(rad, k)) for k in ("background_removal", "background_record", "enable_tx_interleave")])
    tr = "mimo_ofdm_jrc_matrix_transpose_0"
    out["transpose_ints"] = np.array([par(tr, k) for k in ("input_len", "output_len", "interp_factor")], np.int64)
    for name, tag in (("fft_vxx_0_1", "range"), ("fft_vxx_0_1_0", "angle"), ("fft_vxx_0_0", "rx_demod"), ("fft_vxx_0", "tx_mod")):
        out["fft_%s_size_forward_shift" % tag] = np.array([par(name, "fft_size"), par(name, "forward") is True, par(name, "shift") is True], np.int64)
        out["fft_%s_window" % tag] = np.asarray(par(name, "window"), np.float64)
    ts = "mimo_ofdm_jrc_target_simulator_0"
    out["tsim_scalars"] = np.array([par(ts, "rcs"), par(ts, "center_freq"), par(ts, "samp_rate"), par(ts, "self_coupling_db")], np.float64)
    out["zero_pad_tail"] = np.int64(par("mimo_ofdm_jrc_zero_pad_0", "pad_tail"))
    cpr = "mimo_ofdm_jrc_ofdm_cyclic_prefix_remover_0"
    out["cp_remover_ints"] = np.array([par(cpr, "fft_len"), par(cpr, "cp_len")], np.int64)
    comm_point(out)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "radar_flowgraph_point.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst)
    for k, v in out.items():
        print("  ", k, v.shape, v.ravel()[:4])


if __name__ == "__main__":
    sys.exit(main())
