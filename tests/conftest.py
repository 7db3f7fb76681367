import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ofdm64():
    """constant tables minted from the reference's embedded ofdm_config module (tests/golden/)"""
    return np.load(os.path.join(GOLDEN, "ofdm_config_64.npz"))


@pytest.fixture(scope="session")
def jrc():
    import jrc_amd
    jrc_amd.load(build_if_missing=True)
    # the tests prepare device buffers with torch, on torch's stream; the library launches on its own, which does not order itself against it:
    # every call into the library first waits for what torch has queued (torch's stream only — the library's own stream ordering is not touched)
    jrc_amd.set_torch_stream_sync(True)
    return jrc_amd


@pytest.fixture(scope="session")
def ctx(jrc):
    """HIP context; a missing extension or device is a hard failure on the GPU tier, never a skip-to-CPU."""
    return jrc.Context(0)


def rel_err(a, b):
    """per-tensor  ||a-b||_inf / ||b||_inf"""
    a = np.asarray(a)
    b = np.asarray(b)
    d = np.abs(b).max()
    return float(np.abs(a - b).max() / (d if d > 0 else 1.0))


def crandn(rng, *shape, scale=1.0):
    return (scale * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))).astype(np.complex64)
