"""Clip-level data parallelism helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL).

The path shards by clip.  Per training step exactly two tensors cross ranks (engine.TrainStep):
the positive-point count (SUM, 1 float) and the flat gradient buffer (SUM).  SUM, not mean: the
reference's losses are sums over the batch divided by an EMA of the GLOBAL positive count
(av_fd_no_recon.py:645-648,662-667), so the full-batch gradient is the sum of the per-rank gradients.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of `n_items` for `rank` (first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_sum_(t, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast_params_(model, src=0, group=None):
    """Identical initial weights on every rank (the reference has a single replica)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=src, group=group)
        for b in model.buffers():
            dist.broadcast(b.data, src=src, group=group)
