"""Sharding of a batch of independent images over the GPUs of one node.

SIFT extraction of one image never needs another image (Pyramid::step1 resets all
per-image state, sift_pyramid.cu:363-370), so the multi-GPU path is a partition of the
batch with NO data-path collective: rank r of G takes images
{i : i mod G == r}.  torch.distributed is only used around it (barrier / reduction of
counters and timings)."""


def shard_indices(n_items, rank, world):
    """Indices of the items rank `rank` owns (round-robin keeps shards within one item)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_items, world))


def shard_sizes(n_items, world):
    return [len(range(r, n_items, world)) for r in range(world)]


def reduce_stats(dist, elapsed, counts, device="cpu"):
    """MAX over ranks of the elapsed time, SUM over ranks of the counters."""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    c = torch.tensor(list(counts), dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in c.tolist()]
