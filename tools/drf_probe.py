import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_extra as be
# usage: drf_probe.py [packets per pass] [grc]   — the device-resident simulation flowgraph at config B's geometry, or (grc) at the .grc's own;
# JRC_DRF_FUSED_MOD / JRC_TSIM_ONCHIP in the environment select the round-6 kernels
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = be.device_resident_flowgraph_grc(F) if len(sys.argv) > 2 and sys.argv[2] == "grc" else be.device_resident_flowgraph(F)
print("F=%d %.4f ms per pass -> %.0f packets/s" % (F, r["ms_per_step"], r["frames_per_s"]))
