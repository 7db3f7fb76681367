"""Import alias: the package directory is named `gr-mimo-ofdm-jrc_amd` (not a valid identifier), so
`import jrc_amd` loads it through importlib and re-exports it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("gr-mimo-ofdm-jrc_amd")
sys.modules[__name__] = _pkg
