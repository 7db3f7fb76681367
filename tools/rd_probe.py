"""range-Doppler (row D) timing (tools only): python tools/rd_probe.py [cfg frames]   [JRC_RD_FOLD=1 for the fold kernel]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extra as be
cases = [(sys.argv[1], int(sys.argv[2]))] if len(sys.argv) > 2 else [("D", 8), ("D", 32), ("B", 64), ("B", 256)]
for cfg, F in cases:
    r = be.range_doppler(cfg, F)
    print(json.dumps({k: r[k] for k in ("frames_per_step", "ms_per_step", "frames_per_s", "GBps_algorithmic")}), cfg)
