"""fused range-angle kernel time at any shape (tools only): python tools/shape_probe.py N T R IR FRAMES [IA]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jrc_amd
from jrc_amd import synth
N, T, R, Ir, F = (int(x) for x in sys.argv[1:6])
Ia = int(sys.argv[6]) if len(sys.argv) > 6 else 16
sc = synth.Scenario(N, T, R, max(T, 8), targets=[(10.0, 20.0, 0.0, 100.0)])
ctx = jrc_amd.Context(0)
P = T * R
rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
ch = jrc_amd.RadarChain(sc.N, T, R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 20.0, 15.0, 0.0, max_frames=F, ctx=ctx)
bufs = ch.alloc(F, "cuda:0")
fr = synth.make_frames(sc, 4)
hf = torch.from_numpy(fr.view(np.float32).reshape((4,) + tuple(bufs["frames"].shape[1:])))
for f0 in range(0, F, 4):
    bufs["frames"][f0:f0 + 4].copy_(hf[:min(4, F - f0)])
torch.cuda.synchronize()
for _ in range(10):
    ch.run(bufs, F)
ctx.sync(); ch.set_timing(True)
for _ in range(30):
    ch.run(bufs, F)
ctx.sync(); kt = ch.get_timing()
NR, NA = N * Ir, P * Ia
byts = F * (P * N * 8 + NR * NA * 8)
ms = kt["range_angle_fused"]
print("N=%d %dx%d Ir=%d Ia=%d F=%d: fused %.4f ms, %.0f GB/s = %.3f of 8 TB/s (launches per run %d; chanest %.4f ms, finalize %.4f ms)"
      % (N, T, R, Ir, Ia, F, ms, byts / ms / 1e6, byts / ms / 1e6 / 8000, ch.launches_per_run(F), kt["radar_chanest"], kt["ra_finalize"]))
