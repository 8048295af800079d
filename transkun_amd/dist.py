"""Host-side helpers for the data-parallel use of the layer (one process per GPU).

Every function of the path treats chains (the last axis of score) independently, so the batch axis shards
with no data-path collective.  The only collective is the training-loss bookkeeping of the reference,
/root/reference/transkun/train.py:215-217 (three scalar all-reduces), fused here into one [3] tensor.
Backend "nccl" is RCCL on ROCm; tests run the same code over "gloo" on CPU.
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_chains(n_chains: int, world_size: int, rank: int, multiple: int = 4) -> Tuple[int, int]:
    """[begin, end) of the chains owned by `rank`: contiguous, balanced, boundaries on multiples of
    `multiple` (the persistent kernels want nBatch % 4 == 0) whenever n_chains allows it."""
    assert 0 <= rank < world_size
    units = n_chains // multiple if n_chains % multiple == 0 else n_chains
    step = multiple if n_chains % multiple == 0 else 1
    base, rem = divmod(units, world_size)
    begin = (rank * base + min(rank, rem)) * step
    end = begin + (base + (1 if rank < rem else 0)) * step
    return begin, end


def fused_loss_allreduce(loss: torch.Tensor, total_len: float, n_batch: float, group=None) -> torch.Tensor:
    """SUM all-reduce of (loss, length, batch count) as ONE [3] fp32 message (train.py:215-217 issues three)."""
    import torch.distributed as dist
    stats = torch.stack([loss.detach().float().reshape(()),
                         torch.tensor(float(total_len), device=loss.device),
                         torch.tensor(float(n_batch), device=loss.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def max_over_ranks(seconds: float, device, group=None) -> float:
    """Elapsed time of the slowest rank (bench.py's timing contract)."""
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
