"""Batch sharding across GPUs: one process per GPU, replicated weights, no
data-path collective.  x_hat / z_q / indices of a shard are exactly what the reference computes on
that shard.  `embedding_loss` and `perplexity` are batch-GLOBAL scalars in the reference
(models/quantizer.py:63-64,70-71); `merge_vq_stats` rebuilds the full-batch values from per-shard
(squared-error sum, histogram) with ONE all-reduce of K+1 numbers -- optional, off the hot path."""
from __future__ import annotations

import torch


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous, balanced [lo, hi) slice of the batch for `rank` (first n%world ranks get one extra)."""
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def vq_stats_from_outputs(loss: torch.Tensor, hist: torch.Tensor, n_rows: int, D: int, beta: float):
    """Per-shard sufficient statistics: sum((z_q-z)^2) recovered from loss = (1+beta)*mean, and counts."""
    sq = loss.double() / (1.0 + beta) * (n_rows * D)
    return torch.cat([sq.reshape(1), hist.double()])


def merge_vq_stats(stats: torch.Tensor, n_rows_total: int, D: int, beta: float, group=None):
    """All-reduce the (K+1)-vector and return (embedding_loss, perplexity) of the FULL batch,
    following models/quantizer.py:63-64 and :70-71."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        stats = stats.clone()
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    sq, hist = stats[0], stats[1:]
    m = (sq / (n_rows_total * D)).float()
    loss = m + beta * m
    p = (hist / n_rows_total).float()
    perplexity = torch.exp(-torch.sum(p * torch.log(p + 1e-10)))
    return loss, perplexity
