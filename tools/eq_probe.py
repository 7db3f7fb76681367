"""equalizer config C timing (tools only): python tools/eq_probe.py   [JRC_EQ_WPE=2|4|6|8]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extra as be
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
print(json.dumps(be.equalizer_config_c(S=S)))
