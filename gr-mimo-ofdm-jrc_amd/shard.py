"""Frame sharding across GPUs.  Frames are independent units — no reference block on the
radar path carries state across frames unless background removal is enabled — so N GPUs split a frame
stream into contiguous blocks with NO data-path collective.  The only collectives are optional: an
all-gather of the per-frame results (48 bytes per frame) and the barrier / MAX-reduce of the timing contract.
One process per GPU; torch.distributed backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
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


def gather_results(local, n_frames, group=None):
    """all-gather per-frame result records (uint8 tensor [n_local, record_bytes]) into frame order on every rank.
    Shards may differ by one frame, so they are padded to the largest shard for the collective."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    sizes = shard_sizes(n_frames, world)
    biggest = max(sizes)
    padded = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([out[r][:sizes[r]] for r in range(world)], dim=0)


def gather_maps(local_maps, n_frames, group=None):
    """optional exchange named by the north star: all-gather range-angle maps (float32 tensor [n_local, NR, NA, 2]) into
    frame order on every rank.  A config-B map is 4 MiB, a config-D map 16 MiB: per step and per GPU this moves
    (world - 1) x n_local maps over xGMI, so it is meant for a few selected frames (detections), not for whole batches.
    RCCL picks the algorithm; on the fully connected 8-GPU xGMI mesh every peer pair has its own link, so the cost is
    one map per link per peer rather than a 7-step ring."""
    return gather_results(local_maps, n_frames, group=group)


def max_over_ranks(seconds, device="cpu", group=None):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
