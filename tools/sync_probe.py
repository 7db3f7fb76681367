#!/usr/bin/env python3
"""Times the device-resident sync front end (jrc_sync_frontend_dev) on a long capture in HBM, and the full comm receive chain
behind it (RX FFT -> equalizer -> Viterbi decoder) per frame.   usage: tools/sync_probe.py [--frames F]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    import torch
    import jrc_amd
    from _streams import CP, N, BurstMaker, ofdm_config
    o = ofdm_config()
    rng = np.random.default_rng(0)
    ctx = jrc_amd.Context(0)
    bm = BurstMaker(ctx)
    parts = []
    for k in range(8):                                   # eight distinct bursts, tiled
        payload = bytes([2]) + rng.integers(0, 256, 199, dtype=np.uint8).tobytes()
        parts.append(bm.burst(payload, rng, lead=500 + 13 * k, tail=1500, cfo=0.01))
    unit = np.concatenate(parts)
    reps = (a.frames + 7) // 8
    x = np.tile(unit, reps)
    n = x.size
    fe = jrc_amd.SyncFrontEnd(N, CP, 0.6, 10, 8 * (N + CP), 4 * (N + CP), o["l_ltf_fir"], max_frames=reps * 8 + 8, max_symbols=64, ctx=ctx)
    d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()).cuda()
    for _ in range(40):                                  # warm-up long enough for the clocks to come up on a cold GPU
        fe.run(d_x, n)
    nf, info = fe.results()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fe.run(d_x, n)
    ctx.sync()
    dt = (time.perf_counter() - t0) / a.iters
    print("capture of %d samples (%.1f ms at 125 MS/s), %d frames found: %.3f ms -> %.0f M samples/s, %.0f k frames/s"
          % (n, n / 125e3, nf, dt * 1e3, n / dt / 1e6, nf / dt / 1e3))


if __name__ == "__main__":
    main()
