"""standalone timing of the batched A1 kernel (tools only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jrc_amd
from jrc_amd import synth
ctx = jrc_amd.Context(0)
for cfg, F in (("B", 512), ("B", 1024), ("D", 128), ("D", 256), ("D", 512)):
    sc = {"B": synth.config_B, "D": synth.config_D}[cfg]()
    n_items = sc.Npre + sc.S
    fr = torch.randn((F, sc.T + sc.R, n_items, sc.N, 2), device="cuda:0")
    H = torch.empty((F, sc.T * sc.R, sc.N, 2), device="cuda:0")
    torch.cuda.synchronize()
    def run():
        ctx.check(ctx.lib.jrc_radar_chanest_dev(ctx.h, sc.N, sc.T, sc.R, sc.S, sc.Npre, n_items, 0, F, fr.data_ptr(), H.data_ptr(), None))
    for _ in range(5): run()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(50): run()
    ctx.sync(); t = (time.perf_counter() - t0) / 50
    byts = F * ((sc.T + sc.R) * sc.S * sc.N * 8 + sc.T * sc.R * sc.N * 8)
    print("config %s F=%d: %.4f ms  %.0f GB/s" % (cfg, F, t * 1e3, byts / t / 1e9))
