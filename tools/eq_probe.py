"""equalizer config C timing (tools only): python tools/eq_probe.py [symbols [fft_len [frames]]]   [JRC_EQ_WPE=2|4|6|8] [JRC_EQ_THREADS=-1|64|128|256]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extra as be
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
F = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
print(json.dumps(be.equalizer_config_c(n_frames=F, S=S, N=N)))
