"""The reference's training step around the hot path, one process per GPU (BASELINE.json configs[4]).

    logp = model.log_prob(audio, notes)          # train.py:186   -> ModelTransformer.py:256-266 (scorer + CRF part here)
    loss = -logp.sum(-1).mean()                  # train.py:187
    (loss / 50).backward()                       # train.py:189
    all_reduce(loss), all_reduce(len), all_reduce(batch)     # train.py:215-217  -> ONE [3] fp32 message
    average_gradients(model, ...)                # train.py:229 / TrainUtil.py:36-48: SUM per parameter, no divide
                                                 #                -> a few flat fp32 buckets

The segment (batch) axis is what the ranks shard: every rank runs scorer + CRF on its own segments' 90 chains each; the
CRF needs no collective (chains are independent).  The backbone and the audio front-end are out of scope (SURVEY 2 rows
5-6): `ctx` -- the backbone's output [N, 90, T, 256] -- is the input here, and the model's remaining 13.6 M parameters
take part in the gradient exchange through `rest`, a flat stand-in of the same size whose gradient is filled
synthetically (the exchange moves the same 54.5 MB per step as the reference's 182 all-reduces).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn

from .dist import FlatGradBucket, allreduce_gradients_flat, fused_loss_allreduce
from .scorer import ScaledInnerProductIntervalScorer

MODEL_PARAMS = 13_610_000        # the shipped 2.0.conf model (13.61 M fp32 = 54.5 MB)


class SegmentModel(nn.Module):
    """What of TransKun (ModelTransformer.py:69-148) takes part in this step: the interval scorer (:102-105, size
    baseSize * scoringExpansionFactor = 256) and a flat stand-in for every other parameter."""

    def __init__(self, size: int = 256, total_params: int = MODEL_PARAMS):
        super().__init__()
        self.scorer = ScaledInnerProductIntervalScorer(size, 1)
        n_scorer = sum(p.numel() for p in self.scorer.parameters())
        self.rest = nn.Parameter(torch.zeros(max(total_params - n_scorer, 1)))


def default_log_prob(scorer, ctx, intervals):
    """ModelTransformer.py:199-225 + :263-265 on the HIP path: scorer -> CRF -> logProb with the loss gradient fused into
    the scorer backward (transkun_amd.fused)."""
    from .fused import scorer_crf_logprob
    return scorer_crf_logprob(scorer, ctx, intervals)


def train_step(model: SegmentModel, ctx: torch.Tensor, intervals, seconds_per_segment: float = 16.0, group=None,
               log_prob: Optional[Callable] = None, bucket_bytes: int = 64 << 20, bucket: Optional[FlatGradBucket] = None):
    """One train.py-shaped step on this rank's segments.  ctx: [N, P, T, size]; intervals: N*P lists (chain n*P + p).
    Returns (stats [3] = summed loss / seconds / batch count over ranks, number of gradient collectives issued).

    bucket: a FlatGradBucket over model.parameters() (make it once, pass it every step): gradients live in its persistent
    flat buffer, the exchange is a reduce-scatter + all-gather that the backward pass starts itself on a side stream and
    that overlaps the loss bookkeeping; without it the gradients are concatenated and all-reduced after the backward
    (allreduce_gradients_flat, the round-1 exchange)."""
    N = ctx.shape[0]
    fn = log_prob or default_log_prob
    if bucket is not None:
        bucket.zero()
        bucket.arm()
    else:
        for p in model.parameters():
            p.grad = None
    logp = fn(model.scorer, ctx, intervals).view(N, -1)            # ModelTransformer.py:266
    loss = -logp.sum(-1).mean()                                    # train.py:187
    # The backbone's backward is not part of this path: its gradient is a stand-in of the right size.  In the model it is
    # the LAST gradient to arrive (the backbone sits upstream of the scorer), so it is written after the backward pass here
    # as well -- the exchange starts when it lands, not earlier.
    (loss / 50).backward()                                         # train.py:189
    if bucket is not None:
        with torch.no_grad():
            model.rest.grad.fill_(1e-3)
        bucket._on_grad(model.rest)                                # the stand-in's "hook": the bucket is complete now
    elif model.rest.grad is None:
        model.rest.grad = torch.full_like(model.rest, 1e-3)
    stats = fused_loss_allreduce(loss, seconds_per_segment * N, 1.0, group=group)       # train.py:215-217
    if bucket is not None:
        ncoll = bucket.wait()                                                           # train.py:229
    else:
        ncoll = allreduce_gradients_flat(model.parameters(), group=group, bucket_bytes=bucket_bytes)
    return stats, ncoll
