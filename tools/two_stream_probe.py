"""probe: does splitting the batch over two streams (read-heavy chanest of one half overlapping the write-heavy fused
kernel of the other) beat one stream?  tools only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jrc_amd
from jrc_amd import synth

sc = synth.config_B(); Ir, Ia = 8, 16; P = 16
rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
fr = synth.make_frames(sc, 16)


def setup(F, n):
    out = []
    for i in range(n):
        ctx = jrc_amd.Context(0)
        ch = jrc_amd.RadarChain(sc.N, sc.T, sc.R, sc.S, sc.Npre, Ir, Ia, rb, ab, 2.4, 14.4, 15.0, 0.0, max_frames=F, ctx=ctx)
        detect = bool(os.environ.get("JRC_TSP_DETECT"))
        if detect:
            ch.set_write_map(False)
        b = ch.alloc(F, "cuda:0", with_map=not detect)
        hf = torch.from_numpy(fr.view(np.float32).reshape((16,) + tuple(b["frames"].shape[1:])))
        for f0 in range(0, F, 16):
            b["frames"][f0:f0 + 16].copy_(hf)
        out.append((ctx, ch, b, F))
    torch.cuda.synchronize()
    return out


def run(parts, steps):
    for _ in range(5):
        for ctx, ch, b, F in parts:
            ch.run(b, F)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for ctx, ch, b, F in parts:
            ch.run(b, F)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


TOTAL = int(os.environ.get("JRC_TSP_FRAMES", "256"))
for n in (1, 2, 4, 8):
    parts = setup(TOTAL // n, n)
    t = run(parts, 50)
    print("streams=%d  ms per %d frames: %.4f  (%.3f M frames/s)" % (n, TOTAL, t, TOTAL / t / 1e3))
    del parts
