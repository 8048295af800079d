"""Interval features for the attribute heads -- host-side mirror of TransKun.fetchIntervalFeaturesBatch
(/root/reference/transkun/ModelTransformer.py:501-532) and of the concatenation that feeds the velocity and
onset/offset predictors (:578-582), on top of the HIP gather kernel.

The reference walks the decoded Python lists per segment, builds index tensors on the host, copies them to the device
and runs two index_selects per segment.  `attribute_input_packed` consumes the packed (begin, end) pairs + offsets
that `semicrf_viterbi` leaves in HBM, so decode -> features needs no host round trip; `fetchIntervalFeaturesBatch`
keeps the reference's signature (Python lists in, four tensors out) for callers that already hold lists.
"""
from __future__ import annotations

import importlib
from typing import List, Sequence, Tuple

import torch

from . import _lib

_nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")


class _IntervalFeatures(torch.autograd.Function):
    """out [K, 3D] = [ctx[c,b] | ctx[c,e] | ctx[c,b] * ctx[c,e]]; differentiable w.r.t. ctx (training calls the
    reference method with the ground-truth intervals, ModelTransformer.py:286-290)."""

    @staticmethod
    def forward(ctx_, ctx3, pairs, offsets, K, nSym):
        C, T, D = ctx3.shape
        out = torch.empty(K, 3 * D, dtype=torch.float32, device=ctx3.device)
        sym = torch.empty(K, dtype=torch.int64, device=ctx3.device)
        sc = torch.empty(K, dtype=torch.int64, device=ctx3.device)
        _lib.ops().interval_features_gather(ctx3, C, T, D, ctx3.stride(-2), pairs, int(K), offsets, int(nSym), out, sym, sc)
        ctx_.save_for_backward(ctx3, pairs, offsets)
        ctx_.K = K
        ctx_.mark_non_differentiable(sym, sc)
        return out, sym, sc

    @staticmethod
    def backward(ctx_, gout, gsym, gsc):
        ctx3, pairs, offsets = ctx_.saved_tensors
        C, T, D = ctx3.shape
        dctx = torch.zeros_like(ctx3)
        g = gout.contiguous()
        _lib.ops().interval_features_gather_bwd(g, ctx3, C, T, D, ctx3.stride(-2), pairs, int(ctx_.K), offsets, dctx, dctx.stride(-2))
        return dctx, None, None, None, None


def attribute_input_packed(ctxBatch: torch.Tensor, pairs: torch.Tensor, offsets: torch.Tensor, K: int = None):
    """ctxBatch [N, SYM, T, D] on the GPU; pairs int32 [>=K, 2] and offsets int32 [N*SYM+1] on the same device (chain
    c = n*SYM + sym, the order of NeuralSemiCRFInterval.decode).  Returns (attributeInput [K, 3D], symIdx [K],
    scatterIdx [K]) -- the reference's torch.cat([ctx_a_all, ctx_b_all, ctx_a_all*ctx_b_all], -1), symIdx_all and
    scatterIdx_all (ModelTransformer.py:501-532, :578-582).  K = offsets[-1] if not given (one host sync)."""
    assert ctxBatch.dim() == 4
    N, SYM, T, D = ctxBatch.shape
    _lib.require_gpu(ctxBatch, "ctxBatch")
    if K is None:
        K = int(offsets[-1])
    x = ctxBatch.float()
    if x.stride(-1) != 1 or not x.is_contiguous():
        x = x.contiguous()
    out, sym, sc = _IntervalFeatures.apply(x.view(N * SYM, T, D), pairs, offsets, int(K), SYM)
    return out, sym, sc


def fetchIntervalFeaturesBatch(ctxBatch: torch.Tensor, intervalsBatch: Sequence[Sequence[Sequence[Tuple[int, int]]]]):
    """Same arguments and results as the reference method (ModelTransformer.py:501-532): ctxBatch [N, SYM, T, D],
    intervalsBatch = per segment, per symbol, a list of (begin, end).  Returns (ctx_a_all, ctx_b_all, symIdx_all,
    scatterIdx_all); ctx_a_all / ctx_b_all are views of one [K, 3D] buffer whose last third already holds their product."""
    N, SYM, T, D = ctxBatch.shape
    assert len(intervalsBatch) == N
    flat: List[Sequence[Tuple[int, int]]] = [sym for seg in intervalsBatch for sym in seg]
    assert len(flat) == N * SYM
    pairs, offsets = _nsci.pack_intervals(flat, T, N * SYM, ctxBatch.device)
    K = getattr(pairs, "_semicrf_K", pairs.shape[0])
    if K == 0:
        raise RuntimeError("fetchIntervalFeaturesBatch: no intervals (the reference fails in torch.cat of an empty list)")
    out, sym, sc = attribute_input_packed(ctxBatch, pairs, offsets, K)
    return out[:, :D], out[:, D:2 * D], sym, sc
