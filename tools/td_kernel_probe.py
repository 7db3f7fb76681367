"""standalone timing of the fused A6+A7+A1 kernel (jrc_radar_chanest_td_dev), tools only:  python tools/td_kernel_probe.py B 256"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jrc_amd
from jrc_amd import synth
cfg, F = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sc = {"B": synth.config_B, "D": synth.config_D}[cfg]()
ctx = jrc_amd.Context(0)
n_items = sc.Npre + sc.S
L = n_items * (sc.N + sc.cp)
tx = torch.randn((F, sc.T, n_items, sc.N, 2), device="cuda:0")
rx = torch.randn((F, sc.R, L, 2), device="cuda:0")
H = torch.empty((F, sc.T * sc.R, sc.N, 2), device="cuda:0")
torch.cuda.synchronize()
def run():
    ctx.check(ctx.lib.jrc_radar_chanest_td_dev(ctx.h, sc.N, sc.cp, sc.T, sc.R, sc.S, sc.Npre, n_items, L, 0, F, tx.data_ptr(), rx.data_ptr(), H.data_ptr(), None))
for _ in range(5): run()
ctx.sync(); t0 = time.perf_counter()
for _ in range(steps): run()
ctx.sync(); t = (time.perf_counter() - t0) / steps
byts = F * ((sc.T + sc.R) * sc.S * sc.N * 8 + sc.T * sc.R * sc.N * 8)
print("config %s F=%d: %.4f ms  %.0f GB/s algorithmic (RX samples of the used symbols + TX rows once + H)" % (cfg, F, t * 1e3, byts / t / 1e9))
