"""N>1 path on CPU: two gloo ranks shard a frame stream, process their blocks independently (the oracle stands
in for the per-GPU chain here) and all-gather the per-frame result records; the gathered stream must equal
the single-process result frame for frame."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import jrc_amd
from jrc_amd import shard, synth
import oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _process(frames, sc, Ir, Ia):
    P = sc.T * sc.R
    rb, ab = jrc_amd.radar_axes(sc.N, sc.fs, Ir, P, Ia)
    recs = []
    for fr in frames:
        rad = oracle.Radar(sc.N, sc.T, sc.R, sc.S, sc.Npre, interp_factor=Ir)
        m = rad.chain([fr[t] for t in range(sc.T)], [fr[sc.T + r] for r in range(sc.R)], Ia)
        r = oracle.ra_estimate(m, rb, ab, 2.4, 29.0, 15.0, 0.0)
        recs.append(np.frombuffer(ctypes.string_at(ctypes.byref(r), ctypes.sizeof(r)), np.uint8).copy())
    return torch.from_numpy(np.stack(recs)) if recs else torch.zeros((0, ctypes.sizeof(oracle.RaResult)), dtype=torch.uint8)


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.Scenario(64, 2, 2, 4, targets=[(12.0, -25.0, 0.0, 100.0)])
    lo, hi = shard.frame_shard(n_frames, rank, world)
    frames = synth.make_frames(sc, hi - lo, first_frame=lo)      # every rank generates only its own block
    local = _process(frames, sc, 4, 8)
    allr = shard.gather_results(local, n_frames)
    maps = torch.arange((hi - lo) * 6, dtype=torch.float32).reshape(hi - lo, 3, 1, 2) + 1000.0 * rank   # stand-in range-angle maps
    allm = shard.gather_maps(maps, n_frames)
    want = torch.cat([torch.arange(n * 6, dtype=torch.float32).reshape(n, 3, 1, 2) + 1000.0 * r
                      for r, n in enumerate(shard.shard_sizes(n_frames, world))])
    assert torch.equal(allm, want)
    t = shard.max_over_ranks(0.1 * (rank + 1))
    v = shard.max_over_ranks_vec([float(rank), -float(rank)])
    assert v == [float(world - 1), 0.0]
    assert shard.min_over_ranks(1.0 if rank != world - 1 else 0.0) == 0.0
    if rank == 0:
        q.put((allr.numpy().copy(), t))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,world", [(6, 2), (7, 2), (16, 8), (19, 8), (5, 8)])
def test_ranks_equal_one(n_frames, world):
    """even shards (one all_gather_into_tensor straight into frame order), ragged shards, and shards of zero frames (5 frames on 8 ranks)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        gathered, tmax = q.get(timeout=180)
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    sc = synth.Scenario(64, 2, 2, 4, targets=[(12.0, -25.0, 0.0, 100.0)])
    ref = _process(synth.make_frames(sc, n_frames), sc, 4, 8).numpy()
    assert gathered.shape == ref.shape and np.array_equal(gathered, ref)
    assert abs(tmax - 0.1 * world) < 1e-9                 # MAX over ranks of the per-rank elapsed time


def test_frame_shard_partitions_exactly():
    for n in (0, 1, 5, 8, 64, 1000):
        for w in (1, 2, 3, 4, 8):
            blocks = [shard.frame_shard(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = shard.shard_sizes(n, w)
            assert max(sizes) - min(sizes) <= 1
            for f in range(n):
                owner = [r for r, (lo, hi) in enumerate(blocks) if lo <= f < hi]
                assert owner == [f * w // n] or len(owner) == 1


def test_ring_warmup_block():
    """background removal on a sharded stream: each rank replays the <= record_len frames in front of its block"""
    for n, w, L in ((64, 4, 8), (10, 4, 8), (7, 8, 3)):
        for r in range(w):
            first, lo, hi = shard.ring_warmup_block(n, r, w, L)
            assert (lo, hi) == shard.frame_shard(n, r, w)
            assert first == max(0, lo - L) and first <= lo
