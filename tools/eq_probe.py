import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extra as be
for nf in (64, 512, 2048, 64, 512, 2048):
    r = be.equalizer_config_c(nf, 4)
    print(nf * 4, "streams", round(r["ms_per_step"], 3), "ms", round(r["lane_frames_per_s"]))
