"""fused-kernel time at a 4x4, fft_len 512 shape (tools only): python tools/n512.py IR FRAMES"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jrc_amd
from jrc_amd import synth
Ir, F = int(sys.argv[1]), int(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 3 else 512
sc = synth.Scenario(N, 4, 4, 32, targets=[(10.0, 20.0, 0.0, 100.0)])
ctx = jrc_amd.Context(0)
rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, 16, 16)
ch = jrc_amd.RadarChain(sc.N, 4, 4, sc.S, sc.Npre, Ir, 16, rb, ab, 2.4, 20.0, 15.0, 0.0, max_frames=F, ctx=ctx)
bufs = ch.alloc(F, "cuda:0")
fr = synth.make_frames(sc, 4)
hf = torch.from_numpy(fr.view(np.float32).reshape((4,) + tuple(bufs["frames"].shape[1:])))
for f0 in range(0, F, 4): bufs["frames"][f0:f0 + 4].copy_(hf[:min(4, F - f0)])
torch.cuda.synchronize()
for _ in range(10): ch.run(bufs, F)
ctx.sync(); ch.set_timing(True)
for _ in range(30): ch.run(bufs, F)
ctx.sync(); kt = ch.get_timing()
NR, NA = N * Ir, 256
byts = F * (16 * N * 8 + NR * NA * 8)
ms = kt["range_angle_fused"]
print("N=%d Ir=%d F=%d: fused %.4f ms, %.0f GB/s = %.3f of 8 TB/s (launches per run %d)" % (N, Ir, F, ms, byts / ms / 1e6, byts / ms / 1e6 / 8000, ch.launches_per_run(F)))
