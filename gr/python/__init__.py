"""GNU Radio 3.8 Python module `mimo_ofdm_jrc` of the MI355X build: re-exports the SWIG-wrapped blocks (same role as the reference's
python/__init__.py:20-42).  The GPU is chosen per block with the environment: JRC_DEVICE=<n>, or JRC_DEVICES=0,1,... for radar_chain."""
from __future__ import unicode_literals

try:
    from .mimo_ofdm_jrc_swig import *      # noqa: F401,F403
except ImportError as exc:                 # a half-installed module must not look like an empty one
    raise ImportError("mimo_ofdm_jrc: the SWIG module is missing or its libraries (gnuradio-mimo_ofdm_jrc, libjrc_hip.so) "
                      "cannot be loaded: %s" % exc)
