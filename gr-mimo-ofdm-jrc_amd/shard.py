"""Frame sharding across GPUs.  Frames are independent units — no reference block on the
radar path carries state across frames unless background removal is enabled — so N GPUs split a frame
stream into contiguous blocks with NO data-path collective.  The only collectives are optional: an
all-gather of the per-frame results (48 bytes per frame) or of selected range-angle maps, and the barrier /
MAX-reduce of the timing contract.
One process per GPU; torch.distributed backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

With background removal enabled (lib/mimo_ofdm_radar_impl.cc:276-300) a frame's estimate depends on the ring of the
`record_len` estimates recorded before it, so frames of one stream stay ordered on one GPU; `ring_warmup_block` gives
the frames a rank must replay in front of its block (the <= record_len estimates at the block boundary, §8(e)).
"""
import torch
import torch.distributed as dist


def frame_shard(n_frames, rank, world):
    """contiguous block [lo, hi) of rank `rank`: frame f belongs to rank f*world//n_frames (sizes differ by <= 1)"""
    lo = -(-rank * n_frames // world)
    hi = -(-(rank + 1) * n_frames // world)
    return lo, hi


def shard_sizes(n_frames, world):
    return [frame_shard(n_frames, r, world)[1] - frame_shard(n_frames, r, world)[0] for r in range(world)]


def ring_warmup_block(n_frames, rank, world, record_len):
    """background removal on a sharded stream: rank `rank` owns [lo, hi) and must first replay frames [lo - w, lo)
    (recording only, outputs dropped) so that its ring holds what a single GPU's ring would hold at frame lo.
    Returns (first_replayed_frame, lo, hi)."""
    lo, hi = frame_shard(n_frames, rank, world)
    w = min(record_len, lo)
    return lo - w, lo, hi


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _needs_host_staging(t, group=None):
    """gloo moves host memory: device tensors are staged through the host (tests of the N>1 path on one GPU only)"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def gather_results(local, n_frames, group=None):
    """all-gather per-frame records (tensor [n_local, ...]) into frame order on every rank: ONE all_gather_into_tensor
    straight into the frame-ordered output when the shards are even; shards that differ by one frame are padded to the
    largest for the collective and compacted afterwards."""
    world = _world(group)
    if world == 1:
        return local
    sizes = shard_sizes(n_frames, world)
    biggest = max(sizes)
    src = local.contiguous()
    stage = _needs_host_staging(src, group)
    dev = src.device
    if stage:
        src = src.cpu()
    tail = tuple(src.shape[1:])
    if min(sizes) == biggest:
        out = src.new_empty((n_frames,) + tail)
        dist.all_gather_into_tensor(out, src, group=group)
    else:
        if src.shape[0] != biggest:
            padded = src.new_zeros((biggest,) + tail)
            padded[:src.shape[0]] = src
            src = padded
        full = src.new_empty((world * biggest,) + tail)
        dist.all_gather_into_tensor(full, src, group=group)
        keep = torch.cat([torch.arange(r * biggest, r * biggest + sizes[r]) for r in range(world)]).to(full.device)
        out = full.index_select(0, keep)
    return out.to(dev) if stage else out


def gather_maps(local_maps, n_frames, group=None):
    """optional exchange named by the north star: all-gather range-angle maps (float32 tensor [n_local, NR, NA, 2]) into
    frame order on every rank.  A config-B map is 4 MiB, a config-D map 16 MiB: per step and per GPU this moves
    (world - 1) x n_local maps over xGMI, so it is meant for a few selected frames (detections), not for whole batches.
    RCCL picks the algorithm; on the fully connected 8-GPU xGMI mesh every peer pair has its own link, so the cost is
    one map per link per peer rather than a 7-step ring."""
    return gather_results(local_maps, n_frames, group=group)


def _reduce(values, op, device="cpu", group=None):
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if _world(group) > 1:
        dist.all_reduce(t, op=op, group=group)
    return [float(v) for v in t.cpu()]


def max_over_ranks(seconds, device="cpu", group=None):
    return _reduce([seconds], dist.ReduceOp.MAX, device, group)[0]


def max_over_ranks_vec(values, device="cpu", group=None):
    return _reduce(values, dist.ReduceOp.MAX, device, group)


def min_over_ranks(value, device="cpu", group=None):
    return _reduce([value], dist.ReduceOp.MIN, device, group)[0]
