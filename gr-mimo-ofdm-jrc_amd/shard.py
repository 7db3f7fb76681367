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


def max_over_ranks(seconds, device="cpu", group=None):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
