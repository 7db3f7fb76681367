#!/usr/bin/env python3
"""Times the device target simulator (jrc_tsim_run_dev) on bursts resident in HBM (accuracy against the oracle is what
tests/test_gpu_tsim.py checks).   usage: tools/tsim_probe.py [--config B|D] [--bursts N] [--targets K]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--bursts", type=int, default=64)
    ap.add_argument("--targets", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-oracle", action="store_true", help="accepted for compatibility; the probe never calls the oracle")
    a = ap.parse_args()
    import torch
    import jrc_amd
    N, S = (256, 64) if a.config == "B" else (1024, 128)
    K = a.targets or (1 if a.config == "B" else 8)
    n = (5 + S + 3) * (N + N // 4)            # preamble + symbols + 3 pad symbols (radar_sim.grc:1548)
    fs, fc = 125_000_000, 24e9
    rng = np.random.default_rng(1)
    tg = (rng.uniform(5, 60, K), rng.uniform(-40, 40, K), rng.uniform(10, 100, K), rng.uniform(-60, 60, K))
    pos = [0.0, 0.00625, 0.0125, 0.01875]
    ctx = jrc_amd.Context(0)
    B = a.bursts
    sim = jrc_amd.target_simulator(*tg, pos, fs, fc, sum_targets=True, max_bursts=B, ctx=ctx)
    x = (rng.standard_normal((B, n)) + 1j * rng.standard_normal((B, n))).astype(np.complex64)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((B, 4, n), dtype=torch.complex64, device="cuda")
    sim.run_dev(d_in, d_out, B, n)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        sim.run_dev(d_in, d_out, B, n)
    ctx.sync()
    dt = (time.perf_counter() - t0) / a.iters
    n2 = 1
    while n2 < 4096 and n % (n2 * 2) == 0:
        n2 *= 2
    direct = n2 >= 16 and n // n2 <= 512 and os.environ.get("JRC_TSIM_BLUESTEIN", "0") in ("", "0")
    alg = B * n * 8 * (1 + 4)                                   # a burst in, R = 4 bursts out
    if direct:                                                  # tsim.hip td_*: (Kz + Kz + Kz... ) see docs/history.md §3.3
        Kz = K
        traffic = B * n * 8 * (Kz + Kz + Kz + 4 + 4 + 4)        # col fwd: in -> U; rows: U -> G; col inv: G -> out
        route = "direct four-step %d x %d" % (n // n2, n2)
    else:
        M = 32768
        while M < 2 * n - 1:
            M *= 2
        traffic = B * K * (M * 8 * (3 + 4 * 4) + n * 8 * (2 + 4 + 4 * 2))
        route = "chirp-z M=%d" % M
    print("config %s: n=%d %s K=%d R=4 bursts=%d: %.3f ms/launch-set, %.0f bursts/s, %.1f M samples/s in, algorithmic %.0f GB/s, ~%.0f GB/s work-buffer traffic (%.1fx algorithmic)"
          % (a.config, n, route, K, B, dt * 1e3, B / dt, B * n / dt / 1e6, alg / dt / 1e9, traffic / dt / 1e9, traffic / alg))


if __name__ == "__main__":
    main()
