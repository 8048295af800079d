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


def allreduce_gradients_flat(parameters, group=None, bucket_bytes: int = 64 << 20) -> int:
    """SUM all-reduce of the gradients of `parameters` (no division), the semantics of the reference's
    `average_gradients(model, parallel=True)` (/root/reference/transkun/TrainUtil.py:36-48: one all_reduce PER PARAMETER
    and the divide commented out) -- but as a few flat fp32 buckets instead of hundreds of small messages: xGMI rings
    are per-link bound, so message count, not bytes, dominates for a 54.5 MB model.  Buckets follow parameter
    order and are reduced in place through flat views.  Returns the number of collectives issued."""
    import torch.distributed as dist
    params = [p for p in parameters if p.requires_grad]
    for p in params:
        if p.grad is None:
            raise RuntimeError("allreduce_gradients_flat: a parameter that requires grad has no gradient "
                               "(the reference's checkNoneGradient warns here, TrainUtil.py:24-33)")
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return 0
    n_coll = 0
    bucket, nbytes = [], 0

    def flush():
        nonlocal bucket, nbytes, n_coll
        if not bucket:
            return
        flat = torch.cat([p.grad.detach().reshape(-1).float() for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.detach().copy_(flat[off:off + n].view_as(p.grad))
            off += n
        n_coll += 1
        bucket, nbytes = [], 0

    for p in params:
        sz = p.grad.numel() * 4
        if bucket and (nbytes + sz > bucket_bytes or p.grad.device != bucket[0].grad.device):
            flush()
        bucket.append(p)
        nbytes += sz
    flush()
    return n_coll
