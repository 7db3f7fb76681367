import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extra as be
for cfg, F in (("B", 256), ("B", 512), ("D", 64), ("D", 256)):
    print(json.dumps(be.radar_with_demod(cfg, F)), flush=True)
