"""Mints tests/golden/ofdm_config_64.npz from the reference's ONLY Python on this path: the `ofdm_config`
epy_module embedded (as a YAML string) in the simulation flowgraphs.  Runs in the build container only
(needs /root/reference); the module text is executed from where it lies and is never copied — only the
constant tables it produces (data) are committed.

    python tests/golden/make_ofdm_config_golden.py
"""
import os
import sys

import numpy as np
import yaml

REF = "/root/reference/examples/simulation"
GRCS = [os.path.join(REF, "radar", "mimo_ofdm_jrc_radar_sim.grc"),
        os.path.join(REF, "communication", "mimo_ofdm_jrc_comm_sim.grc")]


def load_module(grc):
    d = yaml.safe_load(open(grc))
    for b in d["blocks"]:
        if b["id"] == "epy_module" and b["name"] == "ofdm_config":
            ns = {}
            exec(compile(b["parameters"]["source_code"], grc + ":ofdm_config", "exec"), ns)
            return ns, d
    raise RuntimeError("ofdm_config not found in " + grc)


def variables(d):
    return {b["name"]: b["parameters"]["value"] for b in d["blocks"] if b["id"] == "variable"}


def main():
    ns, d = load_module(GRCS[0])
    ns2, _ = load_module(GRCS[1])
    keys = ["data_subcarriers", "pilot_subcarriers", "pilot_symbols", "l_stf_ltf_64", "ltf_64", "P_ltf",
            "ltf_mapped_sc__ss_sym", "l_ltf_fir"]
    out = {}
    for k in keys:
        a = np.asarray(ns[k])
        b = np.asarray(ns2[k])
        assert a.shape == b.shape and np.array_equal(a, b), "radar/comm flowgraphs disagree on " + k
        out[k] = a.astype(np.complex64) if np.iscomplexobj(a) else a.astype(np.int32)
    out["N_tx"] = np.int32(ns["N_tx"])
    out["N_ltf"] = np.int32(ns["N_ltf"])
    out["N_sc"] = np.int32(ns["N_sc"])
    v = variables(d)
    # radar flowgraph operating point (strings evaluated by hand: plain literals only)
    out["radar_vars"] = np.array([str((k, v[k])) for k in
                                  ("fft_len", "samp_rate", "interp_factor_range", "interp_factor_angle", "N_rx",
                                   "freq", "noise_figure_dB") if k in v])
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ofdm_config_64.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: getattr(val, "shape", None) for k, val in out.items()})


if __name__ == "__main__":
    sys.exit(main())
