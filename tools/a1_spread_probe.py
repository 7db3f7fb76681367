"""Why does the A1 launch (radar_chanest_x2_kernel) cost 0.092 ms in the headline leg and 0.126-0.130 ms in the detect-only / power-map
legs?  One process, config B, 512 frames; the chain's own HIP events time A1 in each scenario (tools only).
JRC_A1_ONLY=<scenario name> runs a single scenario (for rocprofv3 --pmc FETCH_SIZE passes)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import jrc_amd
from jrc_amd import synth

F = int(os.environ.get("JRC_A1_FRAMES", "512"))
sc = synth.config_B()
Ir, Ia, P = 8, 16, sc.T * sc.R
rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
ctx = jrc_amd.Context(0)
only = os.environ.get("JRC_A1_ONLY")


def make_chain():
    return jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 14.36, 15.0, 0.0, max_frames=F, ctx=ctx)


def fill(bufs):
    fr = synth.make_frames(sc, 8)
    hf = torch.from_numpy(fr.view(np.float32).reshape((8,) + tuple(bufs["frames"].shape[1:])))
    for f0 in range(0, F, 8):
        bufs["frames"][f0:f0 + 8].copy_(hf[:min(8, F - f0)])
    torch.cuda.synchronize()


def measure(chain, bufs, steps=30, warm=5, between=None):
    for _ in range(warm):
        chain.run(bufs, F)
    ctx.sync()
    chain.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        if between is not None:
            ctx.sync()
            between()
        chain.run(bufs, F)
    ctx.sync()
    wall = (time.perf_counter() - t0) / steps
    kt = chain.get_timing()
    chain.set_timing(False)
    return dict(a1_ms=kt["radar_chanest"], fused_ms=kt["range_angle_fused"], fin_ms=kt["ra_finalize"], wall_ms=wall * 1e3)


out = {}


def scenario(name):
    return only is None or only == name


chain = make_chain()
bufs = chain.alloc(F, "cuda:0")
fill(bufs)
if scenario("map"):
    out["map"] = measure(chain, bufs)
if scenario("a1_alone"):
    n_items = sc.Npre + sc.S
    fr = bufs["frames"]
    H = bufs["chanest"]

    def a1():
        ctx.check(ctx.lib.jrc_radar_chanest_dev(ctx.h, sc.N, sc.T, sc.R, sc.S, sc.Npre, n_items, 0, F, fr.data_ptr(), H.data_ptr(), None))
    for _ in range(5):
        a1()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        a1()
    ctx.sync()
    out["a1_alone"] = dict(a1_ms=(time.perf_counter() - t0) / 50 * 1e3)
if scenario("map_idle"):
    out["map_idle"] = measure(chain, bufs, between=lambda: time.sleep(0.002))

# detect-only on the SAME chain object and input buffer (what tools/bench_extra.py does)
if scenario("detect") or scenario("detect_idle") or scenario("detect_flush") or scenario("detect_keepmap"):
    chain.set_write_map(False)
    if scenario("detect_keepmap"):
        out["detect_keepmap"] = measure(chain, bufs)          # the 2 GiB map stays allocated (untouched)
    keep = bufs["map"]
    bufs["map"] = None
    if not scenario("detect_keepmap") or only is None:
        del keep
        torch.cuda.empty_cache()
    if scenario("detect"):
        out["detect"] = measure(chain, bufs)
    if scenario("detect_idle"):
        out["detect_idle"] = measure(chain, bufs, between=lambda: time.sleep(0.002))
    if scenario("detect_flush"):
        junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")       # 1 GiB written between steps: whatever sat in the caches is gone

        def flush():
            junk.fill_(1)
            torch.cuda.synchronize()
        out["detect_flush"] = measure(chain, bufs, between=flush)
        del junk
    chain.set_write_map(True)

# a fresh chain + fresh buffers in detect-only mode from the start (no map ever allocated)
if scenario("detect_fresh"):
    chain2 = make_chain()
    chain2.set_write_map(False)
    b2 = chain2.alloc(F, "cuda:0", with_map=False)
    fill(b2)
    out["detect_fresh"] = measure(chain2, b2)
    chain2.close()

if scenario("power"):
    chain3 = make_chain()
    chain3.set_map_format(True)
    b3 = chain3.alloc(F, "cuda:0", power_map=True)
    fill(b3)
    out["power"] = measure(chain3, b3)
    out["power_idle"] = measure(chain3, b3, between=lambda: time.sleep(0.002))
    chain3.set_write_map(False)                               # the same chain and buffers in detect-only mode: is it the buffers or the kernel before?
    out["power_buffers_detect_mode"] = measure(chain3, b3)
    chain3.set_write_map(True)
    out["power_again"] = measure(chain3, b3)
    chain3.close()

if scenario("map_again"):
    torch.cuda.empty_cache()
    chain4 = make_chain()
    b4 = chain4.alloc(F, "cuda:0")
    fill(b4)
    out["map_again"] = measure(chain4, b4)

print(json.dumps(out))
