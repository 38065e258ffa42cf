"""Multi-GPU partitioning of a batch of independent images.

Images are independent, so N GPUs are N replicas of the decoder, each working on a contiguous
block of the batch: image i goes to rank i // ceil(n / world).  There is no data-path exchange
and therefore no collective; torch.distributed is used only by the harness (barrier, max of
the per-rank wall time, optional gathering of checksums)."""
import math


def shard_bounds(n_images, rank, world):
    """[lo, hi) of the images rank `rank` decodes."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    per = math.ceil(n_images / world) if n_images else 0
    lo = min(n_images, rank * per)
    return lo, min(n_images, lo + per)


def shard(items, rank, world):
    lo, hi = shard_bounds(len(items), rank, world)
    return items[lo:hi]


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (the bench reports the slowest rank's time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
