#!/usr/bin/env python3
"""Times the device bit codec on frame batches resident in HBM.  usage: tools/codec_probe.py [--frames F] [--bytes N] [--mcs M]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8192)
    ap.add_argument("--bytes", type=int, default=500)
    ap.add_argument("--mcs", type=int, default=2)
    ap.add_argument("--carriers", type=int, default=48)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    import torch
    import jrc_amd
    ctx = jrc_amd.Context(0)
    F, nb, ndc = a.frames, a.bytes, a.carriers
    rng = np.random.default_rng(0)
    psdu = rng.integers(0, 256, (F, nb), dtype=np.uint8)
    d_psdu = torch.from_numpy(psdu).cuda()
    d_len = torch.full((F,), nb, dtype=torch.int32, device="cuda")
    d_scr = torch.from_numpy((1 + np.arange(F) % 127).astype(np.uint8)).cuda()
    enc = jrc_amd.stream_encoder(a.mcs, ndc, ctx=ctx)
    dec = jrc_amd.stream_decoder(ndc, ctx=ctx)
    ns = ctx.lib.jrc_stream_n_ofdm_sym(a.mcs, ndc, nb + 4) * ndc
    d_sym = torch.zeros((F, ns), dtype=torch.complex64, device="cuda")
    d_ns = torch.zeros(F, dtype=torch.int32, device="cuda")
    d_mcs = torch.full((F,), a.mcs, dtype=torch.int32, device="cuda")
    d_nb = torch.full((F,), nb + 4, dtype=torch.int32, device="cuda")
    d_pl = torch.zeros((F, nb), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(F, dtype=torch.int32, device="cuda")

    def t(fn):
        fn(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / a.iters
    te = t(lambda: enc.encode_dev(d_psdu, nb, d_len, d_scr, d_sym, ns, d_ns, F))
    td = t(lambda: dec.decode_dev(d_sym, ns, d_mcs, d_nb, d_pl, nb, d_st, F))
    ok = int(d_st.sum().item())
    same = bool((d_pl == d_psdu).all().item())
    print("mcs %d, %d carriers, %d-byte PDUs, %d frames: encode %.3f ms (%.2f M frames/s, %.1f Gbit/s payload), "
          "decode %.3f ms (%.0f k frames/s, %.2f Gbit/s payload); crc ok %d/%d, payload round trip %s"
          % (a.mcs, ndc, nb, F, te * 1e3, F / te / 1e6, F * nb * 8 / te / 1e9, td * 1e3, F / td / 1e3, F * nb * 8 / td / 1e9, ok, F, same))


if __name__ == "__main__":
    main()
