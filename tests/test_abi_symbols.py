"""The C-ABI library must load without a GPU and export every symbol include/jrc.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "jrc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(jrc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for must in ("jrc_create", "jrc_radar_work", "jrc_fft_vcc", "jrc_matrix_transpose", "jrc_ra_estimate",
                 "jrc_cp_remove", "jrc_fft_peak_detect", "jrc_chain_run_dev"):
        assert must in names


def test_library_exports_every_declared_symbol(jrc):
    lib = ctypes.CDLL(jrc.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, "include/jrc.h declares symbols the library does not export: %s" % missing


def test_abi_version_and_strerror(jrc):
    L = jrc.load()
    assert L.jrc_abi_version() == 1
    assert b"MATRIX TRANSPOSE" in L.jrc_strerror(jrc.JRC_ERR_LENGTH_MISMATCH)
    assert b"no CPU fallback" in L.jrc_strerror(jrc.JRC_ERR_NO_DEVICE)


def test_no_silent_cpu_fallback(jrc):
    """Without a HIP device the product must fail loudly (never route through oracle/ or numpy)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tier")
    assert jrc.load().jrc_device_count() == 0
    with pytest.raises(jrc.JrcError) as e:
        jrc.Context(0)
    assert e.value.status == jrc.JRC_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "jrc_oracle" not in txt, f
