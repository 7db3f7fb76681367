"""A1 (radar_chanest_x2_kernel) against the Infinity Cache: the same launch timed with HIP events on one stream
  * alone, back to back, for batches whose input is smaller / larger than the 256 MiB cache;
  * at 512 config-B frames (537 MB of input) with X MB written by another kernel between two launches.
If what precedes A1 decides how much of its cyclically re-read input still sits in the cache, the time follows X (tools only)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import jrc_amd
from jrc_amd import synth

sc = synth.config_B()
ctx = jrc_amd.Context(0)
n_items = sc.Npre + sc.S
s = torch.cuda.Stream()
out = {"alone": {}, "after_write_MB": {}, "rotating_inputs": {}}


def a1(fr, H, F):
    ctx.check(ctx.lib.jrc_radar_chanest_dev(ctx.h, sc.N, sc.T, sc.R, sc.S, sc.Npre, n_items, 0, F, fr.data_ptr(), H.data_ptr(), s.cuda_stream))


def timed(fn_before, fn, reps=30, warm=5):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    with torch.cuda.stream(s):
        for i in range(warm + reps):
            if fn_before is not None:
                fn_before(i)
            if i >= warm:
                ev[i - warm][0].record(s)
            fn(i)
            if i >= warm:
                ev[i - warm][1].record(s)
    s.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


with torch.cuda.stream(s):
    for F in (64, 128, 256, 512, 1024, 2048):
        fr = torch.randn((F, sc.T + sc.R, n_items, sc.N, 2), device="cuda:0")
        H = torch.empty((F, sc.T * sc.R, sc.N, 2), device="cuda:0")
        s.synchronize()
        ms = timed(None, lambda i: a1(fr, H, F))
        byts = F * ((sc.T + sc.R) * sc.S * sc.N * 8 + sc.T * sc.R * sc.N * 8)
        out["alone"][F] = dict(ms=ms, us_per_frame=ms * 1e3 / F, GBps=byts / ms / 1e6, input_MB=fr.numel() * 4 / 1e6)
        del fr, H
    F = 512
    fr = torch.randn((F, sc.T + sc.R, n_items, sc.N, 2), device="cuda:0")
    H = torch.empty((F, sc.T * sc.R, sc.N, 2), device="cuda:0")
    junk = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")
    for mb in (0, 16, 64, 128, 256, 512, 1024, 2048):
        n = mb << 20
        ms = timed((lambda i: junk[:n].fill_(i & 255)) if mb else None, lambda i: a1(fr, H, F))
        out["after_write_MB"][mb] = ms
    del junk
    # four input sets used in turn: nothing of a set can still be cached when its turn comes again (4 x 537 MB)
    sets = [fr] + [torch.randn_like(fr) for _ in range(3)]
    for k in (1, 2, 4):
        out["rotating_inputs"][k] = timed(None, lambda i: a1(sets[i % k], H, F))
print(json.dumps(out))
