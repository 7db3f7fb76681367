"""PCIe-inclusive rate of the radar chain: frames start in HOST memory (tools only; bench.py's `value` is HBM-resident).

  python tools/feed_probe.py [--config B] [--seconds 2]

One JSON line per setting: slots x frames per slot, hipGraph replay on/off, frames filled in place in the pinned staging
("pinned") or staged from pageable memory by the library ("pageable", one host thread doing the memcpy)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jrc_amd
from jrc_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="B")
ap.add_argument("--seconds", type=float, default=2.0)
a = ap.parse_args()
sc = {"A": synth.config_A, "B": synth.config_B, "D": synth.config_D}[a.config]()
Ir, Ia = 8, 16
P = sc.T * sc.R
rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
ndr = 2 * 3e8 / (2 * sc.fs)
nda = 2 * float(np.rad2deg(np.arcsin(2 / P))) if P > 2 else 30.0
ctx = jrc_amd.Context(0)
base = synth.make_frames(sc, 8)

for fps, slots in ((1, 4), (4, 4), (16, 3), (64, 3), (128, 3)):
    if a.config == "D" and fps > 16:
        continue
    src = np.concatenate([base] * (fps // 8 + 1))[:fps].copy()
    for graph in (False, True):
        for mode in ("pinned", "pageable"):
            feed = jrc_amd.ChainFeed(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, ndr, nda, 15.0, 0.0, ctx=ctx,
                                     n_slots=slots, frames_per_slot=fps, graph=graph)
            for s in range(slots):                    # fill every slot's staging once; "pinned" then re-submits in place
                feed.acquire()[:] = src
                feed.submit(None, fps)
            for s in range(slots):
                feed.collect()
            done, lat = 0, []
            t0 = time.perf_counter()
            while True:
                while feed.pending() < slots:
                    if mode == "pinned":
                        feed.acquire()
                        feed.submit(None, fps)
                    else:
                        feed.submit(src)
                r, _ = feed.collect()
                done += len(r)
                if time.perf_counter() - t0 > a.seconds:
                    break
            while feed.pending():
                done += len(feed.collect()[0])
            el = time.perf_counter() - t0
            # latency of one batch through an otherwise idle pipeline
            for _ in range(20):
                t1 = time.perf_counter()
                feed.acquire(); feed.submit(None, fps); feed.collect()
                lat.append(time.perf_counter() - t1)
            print(json.dumps({"config": a.config, "frames_per_slot": fps, "slots": slots, "graph": graph, "source": mode,
                              "frames_per_s": done / el, "host_GBps": done * feed.frame_bytes / el / 1e9,
                              "idle_latency_us": 1e6 * float(np.median(lat)), "stats": feed.stats()}), flush=True)
            feed.close()
