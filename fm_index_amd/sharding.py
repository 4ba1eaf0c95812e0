"""Multi-GPU layer: patterns are independent (wrapper.rs:103-124 reads
only immutable index state), so a batch shards contiguously across ranks with the index
REPLICATED on every GPU, and the only exchange is the result gather.  One process per GPU;
torch.distributed is the transport ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests).  No data-path collective exists besides these gathers.
"""
import torch
import torch.distributed as dist


def shard_range(nitems, rank, world):
    """Contiguous shard [lo, hi) of rank: item k belongs to rank floor(k * world / nitems)."""
    lo = (nitems * rank + world - 1) // world
    hi = (nitems * (rank + 1) + world - 1) // world
    return lo, hi


def gather_counts(local, nitems, group=None):
    """All-gather per-pattern values (counts, s or e) back into input order.

    `local` is this rank's 1-D int64 tensor for its shard_range; returns the full tensor of
    `nitems` entries on every rank.  Equal shards use one all_gather_into_tensor; ragged
    shards are padded to the largest shard.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(nitems, r, world)[1] - shard_range(nitems, r, world)[0] for r in range(world)]
    assert local.numel() == sizes[rank]
    if len(set(sizes)) == 1:
        out = torch.empty(nitems, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    buf = torch.empty(mx * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx:r * mx + sizes[r]] for r in range(world)])


def gather_positions(local_counts, local_pos, nitems, group=None):
    """Variable-length gather of locate output.

    Returns (offsets[nitems+1], positions[total]) in input order, identical on every rank and
    identical to what a single GPU produces: counts are gathered first, then the position
    lists padded to the largest shard total.
    """
    world = dist.get_world_size(group)
    counts = gather_counts(local_counts, nitems, group)
    off = torch.zeros(nitems + 1, dtype=torch.int64, device=counts.device)
    off[1:] = torch.cumsum(counts, 0)
    totals = []
    for r in range(world):
        lo, hi = shard_range(nitems, r, world)
        totals.append(int(off[hi] - off[lo]))
    mx = max(max(totals), 1)
    pad = torch.zeros(mx, dtype=local_pos.dtype, device=local_pos.device)
    pad[:local_pos.numel()] = local_pos
    buf = torch.empty(mx * world, dtype=local_pos.dtype, device=local_pos.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    pos = torch.cat([buf[r * mx:r * mx + totals[r]] for r in range(world)])
    return off, pos
