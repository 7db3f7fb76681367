"""prototype check + timing of the data-only equalizer kernel (round 5, attempt 2): head launch (7 symbols) -> jrc_equalizer_data_proto_dev -> compare with
the one-launch kernel's output, time it.   usage: tools/eq_proto_probe.py [ring]"""
import ctypes as C
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import jrc_amd
from jrc_amd import synth
import bench_extra as be

N, cp, T, S = 256, 64, 4, 64
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
streams = 8192
base = []
for i in range(8):
    h = (rng.standard_normal(T) + 1j * rng.standard_normal(T)).astype(np.complex64)
    y = np.tensordot(h, tx, axes=(0, 0)); y = np.concatenate([y[3:4], y[3:]], axis=0)
    y = y + 1e-3 * (rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape))
    base.append(y.astype(np.complex64))
x = np.stack([base[i % 8] for i in range(streams)])
n_sym = x.shape[1]
os.environ["JRC_EQ_SPLIT"] = "0"
eq = jrc_amd.mimo_ofdm_equalizer(jrc_amd.LS, 24e9, 125e6, N, cp, data, pilots, pil, ltf, mapped, T, n_streams=streams, ctx=ctx)
d_in = torch.from_numpy(x.view(np.float32).reshape(streams, n_sym, N, 2)).to("cuda:0")
d_head = d_in[:, :7].contiguous()
d_ph = torch.zeros(streams, dtype=torch.float64, device="cuda:0")
want, n_out, _ = eq.frames_dev(d_in, d_ph, n_sym, S)
ctx.sync()
want = want.cpu().numpy()
t_full = be.timed(lambda: eq.frames_dev(d_in, d_ph, n_sym, S))
t_head = be.timed(lambda: eq.frames_dev(d_head, d_ph, 7, S))
L = ctx.lib
L.jrc_equalizer_data_proto_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
out = torch.zeros((streams, S, nd, 2), dtype=torch.float32, device="cuda:0")
for ring in [int(v) for v in (sys.argv[1:] or ["2", "3", "4", "6", "8"])]:
    eq.frames_dev(d_head, d_ph, 7, S)                       # leaves the stream state behind the head
    ctx.check(L.jrc_equalizer_data_proto_dev(eq.h, streams, n_sym, 7, S, ring, d_in.data_ptr(), out.data_ptr(), None))
    ctx.sync()
    got = out.cpu().numpy()
    same = np.array_equal(got, want)
    t = be.timed(lambda: ctx.check(L.jrc_equalizer_data_proto_dev(eq.h, streams, n_sym, 7, S, ring, d_in.data_ptr(), out.data_ptr(), None)))
    print("ring %d: data-only kernel %.4f ms (bit-equal to the one-launch kernel: %s); one launch %.4f ms, head launch %.4f ms -> head + data %.4f ms = %.2f M frames/s"
          % (ring, t * 1e3, same, t_full * 1e3, t_head * 1e3, (t_head + t) * 1e3, streams / 4 / (t_head + t) / 1e6), flush=True)
