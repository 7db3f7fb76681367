#!/usr/bin/env python3
"""Frames that live in HOST memory (what a GNU Radio scheduler hands a block) through the radar branch on the GPU.

jrc_amd.ChainFeed keeps `n_slots` batches in flight: while batch k runs through A1..A5, batch k+1 is being copied in and the
48-byte results of batch k-1 are on their way back; results arrive in submission order.  The range-angle maps stay in HBM
(ask for the first `maps_per_slot` maps of a batch if you want to look at some).

    python examples/host_fed_radar.py            # needs an MI355X
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jrc_amd
from jrc_amd import synth


def main():
    sc = synth.Scenario(64, 4, 2, 4, targets=[(10.0, 20.0, 0.0, 100.0)])         # the reference flowgraph's shape: 4 TX x 2 RX, 64 carriers
    Ir, Ia, P = 8, 16, sc.T * sc.R
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    feed = jrc_amd.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 28.96, 15.0, 0.0,
                             n_slots=3, frames_per_slot=32, maps_per_slot=1)
    frames = synth.make_frames(sc, 32)                                           # [32, T+R, n_items, N] complex64 in pageable memory
    n_batches, done, t0 = 40, 0, time.perf_counter()
    submitted = 0
    while done < n_batches:
        while submitted < n_batches and feed.pending() < feed.n_slots:
            feed.submit(frames)                                                  # or: feed.acquire()[:] = frames; feed.submit()
            submitted += 1
        results, maps = feed.collect(want_maps=True)
        done += 1
        if done == 1:
            r = results[0]
            print("first frame: range %.2f m, angle %.2f deg, snr %.1f dB; map %s" % (r.range_val, r.angle_val, r.snr_est, maps[0].shape))
    dt = time.perf_counter() - t0
    print("%d frames from host memory in %.1f ms: %.0f k frames/s" % (n_batches * 32, dt * 1e3, n_batches * 32 / dt / 1e3))
    feed.close()

    # The same stream for a consumer of the detections only: no map is stored (detect-only mode, results bit-identical), the batches are
    # dealt over every GPU of the node from this one process (a host thread per GPU stages and enqueues: submit_many), and the clutter
    # that does not move is removed with mimo_ofdm_radar's background history when the stream stays on one GPU.
    devices = list(range(jrc_amd._device_count()))
    feed = jrc_amd.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 28.96, 15.0, 0.0, n_slots=2, frames_per_slot=32,
                             devices=devices if len(devices) > 1 else None)
    feed.set_write_map(False)
    if len(devices) == 1:
        feed.set_background(True, True, 8)
    done, t0 = 0, time.perf_counter()
    for _ in range(10):
        free = feed.n_slots - feed.pending()
        feed.submit_many([frames] * free)
        while feed.pending():
            done += len(feed.collect()[0])
    dt = time.perf_counter() - t0
    print("detect-only on %d GPU(s)%s: %d frames in %.1f ms" % (len(devices), ", background removal on" if len(devices) == 1 else "", done, dt * 1e3))
    feed.close()


if __name__ == "__main__":
    main()
