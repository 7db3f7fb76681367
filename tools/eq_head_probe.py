"""equalizer head cost by number of input symbols (tools only): frames are cut after k symbols"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import jrc_amd
from jrc_amd import synth
import bench_extra as be

N, cp, T, S = 256, 64, 4, int(os.environ.get("EQ_HEAD_S", "8"))
rng = np.random.default_rng(0)
guard = 16
act = [c for c in range(-N // 2 + guard, N // 2 - guard + 1) if c != 0]
pilots = [c for c in act if c % 32 == 16][:8]
data = [c for c in act if c not in pilots]
ltf = np.zeros(N, np.complex64); ltf[np.array(act) + N // 2] = rng.choice([-1.0, 1.0], len(act))
mapped = np.stack([(synth.hadamard(T) * ltf[sc]).reshape(-1) for sc in range(N)]).astype(np.complex64)
pil = np.tile(np.array([1, 1, 1, -1, 1, 1, 1, -1], np.complex64)[:len(pilots)], (8, 1))
sw = np.stack([ltf, ltf, ltf, ltf])
ctx = jrc_amd.Context(0)
pre = jrc_amd.mimo_precoder(N, T, 1, data, pilots, pil, sw, mapped, ctx=ctx)
nd = len(data); mcs = 2; nbytes = (S * nd - 22) // 8
pts = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * (0.707107 / 2)
s = pts[rng.integers(0, 4, S * nd)].astype(np.complex64)
tx = pre.work(s, mcs, jrc_amd.DATA, nbytes)
h = (rng.standard_normal(T) + 1j * rng.standard_normal(T)).astype(np.complex64)
y = np.tensordot(h, tx, axes=(0, 0)); y = np.concatenate([y[3:4], y[3:]], axis=0).astype(np.complex64)
streams = 8192
x = np.stack([y] * streams)
eq = jrc_amd.mimo_ofdm_equalizer(jrc_amd.LS, 24e9, 125e6, N, cp, data, pilots, pil, ltf, mapped, T, n_streams=streams, ctx=ctx)
d_ph = torch.zeros(streams, dtype=torch.float64, device="cuda:0")
for k in [int(v) for v in os.environ.get("EQ_HEAD_KS", "1,2,3,4,7,8,9,15").split(",")]:
    d_in = torch.from_numpy(np.ascontiguousarray(x[:, :k]).view(np.float32).reshape(streams, k, N, 2)).to("cuda:0")
    t = be.timed(lambda: eq.frames_dev(d_in, d_ph, k, S), steps=20, warm=3)
    print("symbols in = %2d: %.4f ms" % (k, t * 1e3), flush=True)
