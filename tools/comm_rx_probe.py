#!/usr/bin/env python3
"""The whole comm receive chain device-resident: capture in HBM -> sync front end (metrics, frame_detector, frame_sync) -> RX FFT ->
mimo_ofdm_equalizer (frame batch) -> stream_decoder (frame batch) -> payload bytes in HBM.  Prints frames/s and checks every PDU.
usage: tools/comm_rx_probe.py [--frames F] [--bytes N]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--bytes", type=int, default=200)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    res = run(a.frames, a.bytes, a.iters)
    if a.json:
        import json
        print(json.dumps(res))
    else:
        print("%d-sample capture, %d frames of %d-byte PDUs: %.2f ms -> %.0f k frames/s, %.0f M samples/s; crc ok %d/%d, payloads intact %s"
              % (res["samples"], res["frames_found"], res["pdu_bytes"], res["ms_per_capture"], res["frames_per_s"] / 1e3, res["M_samples_per_s"],
                 res["crc_ok"], res["frames"], res["payloads_intact"]))


def run(frames=1024, nbytes=200, iters=5):
    """build a capture of `frames` PDUs, run the device-resident receive chain on it `iters` times, check every PDU; returns the figures"""
    class _A:
        pass
    a = _A()
    a.frames, a.bytes, a.iters = frames, nbytes, iters
    import torch
    import jrc_amd
    from _streams import CP, N, BurstMaker, ofdm_config
    o = ofdm_config()
    rng = np.random.default_rng(0)
    mcs, ndc = 2, 48
    ctx = jrc_amd.Context(0)
    bm = BurstMaker(ctx, mcs)
    parts, payloads = [], []
    for k in range(8):
        payload = bytes([2]) + rng.integers(0, 256, a.bytes - 1, dtype=np.uint8).tobytes()
        parts.append(bm.burst(payload, rng, lead=500 + 13 * k, tail=1500, cfo=0.01))
        payloads.append(payload)
    reps = (a.frames + 7) // 8
    F = reps * 8
    x = np.tile(np.concatenate(parts), reps)
    n = x.size
    ns = jrc_amd.n_ofdm_sym(mcs, ndc, a.bytes + 4)
    S = 2 + 1 + 4 + ns                                   # LTF x2, SIG, MIMO-LTFs, data
    L = ctx.lib
    fe = jrc_amd.SyncFrontEnd(N, CP, 0.6, 10, 8 * (N + CP), 4 * (N + CP), o["l_ltf_fir"], max_frames=F, max_symbols=S, ctx=ctx)
    eq = jrc_amd.mimo_ofdm_equalizer(0, 24e9, 125e6, N, CP, o["data_subcarriers"], o["pilot_subcarriers"], o["pilot_symbols"], o["ltf_64"],
                                     o["ltf_mapped_sc__ss_sym"], 4, n_streams=F, ctx=ctx)
    dec = jrc_amd.stream_decoder(ndc, ctx=ctx)
    d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()).cuda()
    sym_f = torch.empty_like(fe.frames)
    d_pl = torch.zeros((F, a.bytes), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(F, dtype=torch.int32, device="cuda")
    ev_words = C.sizeof(jrc_amd.EqEvent) // 8

    ts = torch.cuda.Stream()                             # one stream for the library calls and the torch glue between them
    sh = ts.cuda_stream

    def step():
        with torch.cuda.stream(ts):
            fe.run(d_x, n, stream=sh)                                                          # frames of time-domain symbols
            ctx.check(L.jrc_fft_vcc_dev(ctx.h, N, 1, 1, None, F * S, fe.frames.data_ptr(), sym_f.data_ptr(), sh))    # RX FFT, shifted
            tagv = fe.d_info.view(torch.float64)[:, 3].contiguous()                            # frame_sync's tag value per frame
            out, n_out, ev = eq.frames_dev(sym_f, tagv, S, ns, stream=sh)
            evw = ev[:, 0].contiguous().view(torch.int64).reshape(F, ev_words)
            d_nb = evw[:, 2].to(torch.int32).contiguous()                                      # stream_start: data_bytes, mcs
            d_mcs = evw[:, 3].to(torch.int32).contiguous()
            dec.decode_dev(out, ns * ndc, d_mcs, d_nb, d_pl, a.bytes, d_st, F, stream=sh)
        return n_out

    for _ in range(10):
        step()
    ts.synchronize(); ctx.sync(); torch.cuda.synchronize()
    nf, info = fe.results()
    ok = int((d_st == 1).sum().item())
    want = torch.from_numpy(np.frombuffer(b"".join(payloads), np.uint8).reshape(8, a.bytes).copy()).cuda().repeat(reps, 1)
    same = bool((d_pl == want).all().item())
    if os.environ.get("JRC_PROBE_DEBUG"):
        bad = torch.nonzero(d_st != 1).flatten().cpu().numpy()
        print("failing frames:", bad[:20], [(info[i].start, info[i].len, info[i].frame_start, info[i].n_out) for i in bad[:6] if i < nf],
              "ok example:", (info[8].start, info[8].len, info[8].frame_start, info[8].n_out))
    # timed in three blocks of `iters` captures, the median block reported: a single stall (allocator, clocks) in a 5 ms region would
    # otherwise decide the figure
    blocks = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(a.iters):
            step()
        ts.synchronize(); ctx.sync(); torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / a.iters)
    dt = sorted(blocks)[1]
    res = dict(what="comm receive chain, device-resident: capture -> sync front end -> RX FFT -> equalizer -> Viterbi decoder",
               samples=n, frames_found=nf, frames=F, pdu_bytes=a.bytes, crc_ok=ok, payloads_intact=same, ms_per_capture=dt * 1e3,
               frames_per_s=F / dt, M_samples_per_s=n / dt / 1e6)
    return res


if __name__ == "__main__":
    main()
