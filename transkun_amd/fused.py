"""Scorer + CRF log-probability as ONE autograd node (the loss gradient fused into the scorer
backward).

The reference's training step (ModelTransformer.py:256-266 after :102-105/:222) is

    S, noise = scorer(ctx)                       # [T,T,N,P]
    crf = NeuralSemiCRFInterval(S.flatten(-2), noise.flatten(-2))
    logp = crf.logProb(intervals)                # dense d(logp)/dS [T,T,NBatch] in the backward

and its backward materialises the dense gradient of S (ComputeLogZFasterGrad.backward,
NeuralSemiCRFInterval.py:469-472: 1.48 GB at T=1024, NBatch=352), which the scorer's backward then reads twice.
Here the backward runs the beta sweep only (`semicrf_beta`) and `interval_score_bwd_fused_ws` evaluates the marginals
while it repacks the cotangent per chain for the two GEMMs that produce dq and dk -- the dense gradient in the CRF's
layout never exists (without a workspace: rebuilt tile by tile inside the direct kernels).  The
one-hot part of logProb's gradient (the path cells of evalPath, :540-548) is a few thousand rows, added by
`interval_score_path_bwd`.

Opt-in: `scorer_crf_logprob(scorer, ctx, intervals)` replaces the three lines above; results and gradients are the
same as the unfused route (tests/test_gpu_parity.py::test_fused_scorer_crf).
"""
from __future__ import annotations

import importlib
import math

import torch
import torch.nn.functional as F

from . import _lib
from .scorer import (_bf16x3_selfcheck, BF16X3, BWD_BF16X3, LEN_BF16X3, PROJ_BF16X3, QPAD, contraction_bits, ScaledInnerProductIntervalScorer, _ScorerLinear, _ScorerLinearPacked, _interval_score_raw, bwd_workspace, proj_forward,
                     proj_input_grad, proj_weight_grad, qd_weights, slot_maps, slot_pitch)

_nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")


def _beta_raw(score, noise):
    T, B = score.shape[0], score.shape[2]
    beta = torch.empty(T, B, dtype=torch.float32, device=score.device)
    ws = _lib.leased_workspace(_lib.OP_LOGZ_FWD, T, B, score.device, "beta")
    _lib.ops().beta(score, noise, beta, ws)
    return beta


class _MergedWeights(torch.autograd.Function):
    """merged_weights on the library's two kernels (csrc/merge_weights.hip): one launch forward, one backward, where the torch
    formulation below costs ~10 small kernels forward and ~25 backward (slices, cats, two single-tile hipBLASLt products, fill +
    copy + add per slice gradient)."""

    @staticmethod
    def forward(ctx, W, bias, D):
        size = W.shape[1]
        Wc, bc = W.contiguous(), bias.contiguous()
        # the GEMM kernels' operand layouts come out of the same launch: Wm with its rows padded to whole chunks of 32 (zero rows:
        # scorer_proj_nn's B for the input gradient) and, for sizes that are whole chunks, the main rows transposed (the forward's B)
        rows = (size + QPAD + 31) // 32 * 32
        full = torch.empty(rows, size, dtype=torch.float32, device=W.device)
        bfull = torch.empty(rows, dtype=torch.float32, device=W.device)
        has_t = size % 32 == 0
        WmT = torch.empty(size, size, dtype=torch.float32, device=W.device) if has_t else full
        _lib.ops().merge_weights_fwd(Wc, bc, D, size, rows, full, bfull, WmT, has_t)
        ctx.save_for_backward(Wc, bc)
        ctx.D = D
        Wm, bm = full[:size + QPAD], bfull[:size + QPAD]
        Wm._semicrf_padded = full
        Wm._semicrf_T = WmT if has_t else None
        return Wm, bm

    @staticmethod
    def backward(ctx, dWm, dbm):
        Wc, bc = ctx.saved_tensors
        D, size = ctx.D, Wc.shape[1]
        if dWm is None:
            dWm = torch.zeros(size + QPAD, size, dtype=torch.float32, device=Wc.device)
        if dbm is None:
            dbm = torch.zeros(size + QPAD, dtype=torch.float32, device=Wc.device)
        dW = torch.empty_like(Wc)
        dbias = torch.empty_like(bc)
        ws = torch.empty(size * (size + 1) * 4, dtype=torch.uint8, device=Wc.device)          # dWm's first size + 1 rows, transposed
        _lib.ops().merge_weights_bwd(Wc, bc, dWm.contiguous(), dbm.contiguous(), D, size, size + QPAD, dW, dbias, ws)
        return dW, dbias, None


def merged_weights(W, bias, D):
    """(Wm, bm) of the merged projection -- see merged_weights_torch for the algebra.  On the GPU (fp32 parameters of the Linear's
    own shape, size <= 256, the default implementation) one kernel each way; anything else as torch operations."""
    if (W.is_cuda and W.dtype == torch.float32 and bias.dtype == torch.float32 and W.dim() == 2 and W.shape[0] == 2 * D + 1
            and bias.shape == (2 * D + 1,) and W.shape[1] <= 256 and _lib.get_impl() == 0):
        return _MergedWeights.apply(W, bias, D)
    return merged_weights_torch(W, bias, D)


def merged_weights_torch(W, bias, D):
    """The reference's two projections q = x Wq^T + bq, k = x Wk^T + bk (LayersTransformer.py:392-397, :406-410) enter the score only
    through <q_e, k_b> = <x_e A + v, x_b> + c_e with A = Wq^T Wk, v = bq Wk, c_e = <x_e, Wq^T bk> + <bq, bk>: ONE size -> size
    projection, the second operand of the contraction is x itself.  Returns (Wm [size+QPAD, size], bm [size+QPAD]) of the single
    GEMM  [z | c | diag | 0 0] = x Wm^T + bm.  Differentiable in W and bias (a 513 x 256 x 256 product: microseconds)."""
    Wq, Wk, wd = W[:D], W[D:2 * D], W[2 * D:2 * D + 1]
    bq, bk, bd = bias[:D], bias[D:2 * D], bias[2 * D:2 * D + 1]
    # [Wk^T ; bk] Wq = [A^T ; Wq^T bk] and [bq] [Wk | bk] = [v | c0]: two products instead of four
    top = _mm_blocks(torch.cat([Wk.t(), bk.unsqueeze(0)]), Wq)                         # [size + 1, size]
    vc = bq.unsqueeze(0).mm(torch.cat([Wk, bk.unsqueeze(1)], dim=1)).squeeze(0)        # [size + 1]
    Wm = torch.cat([top, wd, W.new_zeros(QPAD - 2, W.shape[1])])
    bm = torch.cat([vc, bd, bias.new_zeros(QPAD - 2)])
    return Wm, bm


def _mm_blocks(a, b, nb: int = 16):
    """a [m, k] @ b [k, n] for m, k, n of a few hundred: hipBLASLt runs such a product (and the two of its backward) as ONE
    256 x 256 tile on one of the 256 CUs, 67 us apiece -- 0.2 ms of a 1.5 ms training step.  As a batch of nb column blocks the
    same flops spread over 2 nb workgroups (13 us)."""
    k, n = b.shape
    if n % nb != 0:
        return a.mm(b)
    bb = b.reshape(k, nb, n // nb).permute(1, 0, 2)
    out = torch.bmm(a.unsqueeze(0).expand(nb, -1, -1), bb)                             # [nb, m, n / nb]
    return out.permute(1, 0, 2).reshape(a.shape[0], n)


def merged_eligible(size: int, T: int) -> bool:
    """Shapes the row-constant form of the scorer kernels takes (include/semicrf_hip.h: interval_score_fwd_pc): the LDS-tiled
    kernels (size % 64 == 0, size <= 256, T >= 128, 16-byte aligned rows -- x and [z | c | diag | 0 0] are contiguous fp32 rows of
    size and size + 4 floats) of the default implementation; anything else takes projection="separate"."""
    return size % 64 == 0 and size <= 256 and T >= 128 and _lib.get_impl() == 0


class _MergedScorerCRFLogProb(torch.autograd.Function):
    """_ScorerCRFLogProb with the merged projection inside the node: x -> [z | c | diag] (ONE GEMM of half the reference Linear's
    width), S = qscale (<z_e, x_b> + c_e) len + diag, logProb.  The backward hands back dx (the gradient through z, c, diag AND
    through x as the contraction's second operand, accumulated inside one GEMM) and the gradients of Wm, bm."""

    @staticmethod
    def forward(ctx, x, Wm, bm, pairs, offsets, N, P, T, D, mode, fs=2):
        # x: [N,P,T,size] fp32 contiguous; D: the reference's contraction size (size * expansionFactor) -- only its scale survives
        C = N * P
        size = x.shape[-1]
        pitch = slot_pitch(P, T, size, N)
        x3 = x.reshape(C, T, size)
        pp = 1 if int(fs) & PROJ_BF16X3 else 0
        zc = proj_forward(x3.view(-1, size), Wm, bm, size, Wt=getattr(Wm, "_semicrf_T", None), prec=pp).view(C, T, size + QPAD)   # [z | c | diag | 0 0]
        ctx.pp = pp
        ctx.Wp = getattr(Wm, "_semicrf_padded", None)      # Wm with its rows padded to whole chunks (the input gradient's B), if it came so
        qs = 1.0 / math.sqrt(D)
        S, noise = _interval_score_raw(zc[..., :size], x3, zc[..., size + 1], T, C, size, qs, mode, int(fs) & ~(BWD_BF16X3 | PROJ_BF16X3), P, pitch,
                                       rowc=zc[..., size])
        if pitch != P:
            real, offmap = slot_maps(N, P, pitch, S.device)
            offsets_s = offsets.index_select(0, offmap)
        else:
            real, offsets_s = None, offsets
        K = getattr(pairs, "_semicrf_K", pairs.shape[0])
        pairs._semicrf_K = K
        _nsci._ready(pairs)
        # logProb = path - logZ from ONE launch (spare waves of the forward sweep compute the path scores: no eval_path kernel, no
        # dependent boundary behind the sweep) -- the same call the unfused _LogProb node makes
        lp, logz, v = _nsci._logprob_fwd_raw(S, noise, pairs, offsets_s, True)
        ctx.save_for_backward(x3, zc, Wm, S, noise, v, logz, pairs, offsets_s)
        ctx.meta = (N, P, T, D, mode, K, pitch, LEN_BF16X3 if int(fs) & BWD_BF16X3 else 0)
        return lp if real is None else lp.index_select(0, real)

    @staticmethod
    def backward(ctx, g):
        x3, zc, Wm, S, noise, v, logz, pairs, offsets_s = ctx.saved_tensors
        N, P, T, D, mode, K, pitch, x3bit = ctx.meta
        C = N * P
        size = x3.shape[-1]
        qs = 1.0 / math.sqrt(D)
        g = g.reshape(C).to(torch.float32).contiguous()
        if pitch != P:
            real, _ = slot_maps(N, P, pitch, S.device)
            g = torch.zeros(N * pitch, dtype=torch.float32, device=S.device).index_copy_(0, real, g)
        ops = _lib.ops()
        beta = _beta_raw(S, noise)
        gneg = (-g).contiguous()
        z = zc[..., :size]
        dzc = torch.empty(C, T, size + QPAD, dtype=torch.float32, device=S.device)
        dzc[..., size + 2:] = 0
        dz, dc, dd = dzc[..., :size], dzc[..., size], dzc[..., size + 1]
        dx = torch.empty(C, T, size, dtype=torch.float32, device=S.device)           # the part through the second operand
        ws = bwd_workspace(C, T, size, S.device)
        ops.interval_score_bwd_fused_ws(S, v, beta, logz, gneg, z, x3, C, T, size, z.stride(-2), size, qs, mode | x3bit, P, pitch, dz, dx, dd, dc,
                                        dz.stride(-2), size, dd.stride(-1), dc.stride(-1), ws)
        del ws
        if K > 0:
            ops.interval_score_path_bwd(g, pairs, int(K), offsets_s, z, x3, C, T, size, z.stride(-2), size, qs, mode, P, pitch, dz, dx, dd,
                                        dc, dz.stride(-2), size, dd.stride(-1), dc.stride(-1))
        g2 = dzc.view(-1, size + QPAD)
        need = ctx.needs_input_grad
        dx2 = dx.view(-1, size)
        if need[0]:
            proj_input_grad(g2, Wm, out=dx2, Wp=ctx.Wp, prec=ctx.pp)  # + the part through [z | c | diag]
        dWm = dbm = None
        if need[1] or need[2]:
            dWm, dbm = proj_weight_grad(g2, x3.view(-1, size), size, prec=ctx.pp)
        return (dx2.view(N, P, T, size) if need[0] else None, dWm if need[1] else None, dbm if need[2] else None, None, None, None, None, None,
                None, None, None)


class _ScorerCRFLogProb(torch.autograd.Function):
    """S never leaves this node, so its chain axis uses the SLOT layout (include/semicrf_hip.h): every segment's P symbols sit
    in `pitch` slots (96 for the model's 90), the CRF kernels see N*pitch chains of which the ghosts are all-zero and their
    results dropped; intervals, logProb and the cotangent are chain-indexed at the node's boundary."""

    @staticmethod
    def forward(ctx, qd, k, pairs, offsets, N, P, T, D, mode, fs=2):
        # qd: [N,P,T,D+QPAD] = [q | diag | zeros] (one GEMM, see scorer.py); k: [N,P,T,D]
        C = N * P
        pitch = slot_pitch(P, T, D, N)
        qd3, k3 = qd.reshape(C, T, D + QPAD), k.reshape(C, T, D)
        qs = 1.0 / math.sqrt(D)
        S, noise = _interval_score_raw(qd3[..., :D], k3, qd3[..., D], T, C, D, qs, mode, int(fs) & ~(BWD_BF16X3 | PROJ_BF16X3), P, pitch)     # fs: 2 | BF16X3
        if pitch != P:
            real, offmap = slot_maps(N, P, pitch, S.device)
            offsets_s = offsets.index_select(0, offmap)
        else:
            real, offsets_s = None, offsets
        K = getattr(pairs, "_semicrf_K", pairs.shape[0])
        pairs._semicrf_K = K
        _nsci._ready(pairs)
        # logProb = path - logZ from ONE launch (spare waves of the forward sweep compute the path scores: no eval_path kernel, no
        # dependent boundary behind the sweep) -- the same call the unfused _LogProb node makes
        lp, logz, v = _nsci._logprob_fwd_raw(S, noise, pairs, offsets_s, True)
        ctx.save_for_backward(qd3, k3, S, noise, v, logz, pairs, offsets_s)
        ctx.meta = (N, P, T, D, mode, K, pitch, LEN_BF16X3 if int(fs) & BWD_BF16X3 else 0)
        return lp if real is None else lp.index_select(0, real)

    @staticmethod
    def backward(ctx, g):
        qd3, k, S, noise, v, logz, pairs, offsets_s = ctx.saved_tensors
        N, P, T, D, mode, K, pitch, x3bit = ctx.meta
        C = N * P
        qs = 1.0 / math.sqrt(D)
        g = g.reshape(C).to(torch.float32).contiguous()
        if pitch != P:
            real, _ = slot_maps(N, P, pitch, S.device)
            g = torch.zeros(N * pitch, dtype=torch.float32, device=S.device).index_copy_(0, real, g)      # ghost slots: 0
        ops = _lib.ops()
        beta = _beta_raw(S, noise)
        gneg = (-g).contiguous()                                   # d logProb / d logZ = -1
        q = qd3[..., :D]
        dqd = torch.empty(C, T, D + QPAD, dtype=torch.float32, device=S.device)       # gradient of [q | diag | pad]
        dqd[..., D + 1:] = 0
        dq, dd = dqd[..., :D], dqd[..., D]
        dk = torch.empty(C, T, D, dtype=torch.float32, device=S.device)
        # with a workspace: marginals evaluated by the repack kernel + two tiled GEMMs (scorer_bwd_gemm.hip)
        ws = bwd_workspace(C, T, D, S.device)
        ops.interval_score_bwd_fused_ws(S, v, beta, logz, gneg, q, k, C, T, D, q.stride(-2), k.stride(-2), qs, mode | x3bit, P, pitch, dq, dk, dd,
                                        dd, dq.stride(-2), D, dd.stride(-1), 0, ws)
        del ws
        if K > 0:
            # + g on the path cells: dq[c,e] += g qs len(e-b) k[c,b], dk[c,b] += g qs len(e-b) q[c,e], ddiag[c,t] += g (b == e)
            ops.interval_score_path_bwd(g, pairs, int(K), offsets_s, q, k, C, T, D, q.stride(-2), k.stride(-2), qs, mode, P, pitch, dq, dk,
                                        dd, dd, dq.stride(-2), D, dd.stride(-1), 0)
        return (dqd.view(N, P, T, D + QPAD), dk.view(N, P, T, D), None, None, None, None, None, None, None, None)


def scorer_crf_logprob(scorer: ScaledInnerProductIntervalScorer, ctx: torch.Tensor, intervals, projection: str = "merged") -> torch.Tensor:
    """log p(intervals | ctx) per chain, [N*P], differentiable w.r.t. ctx and the scorer's parameters.

    ctx: [N, P, T, size] on the GPU; intervals: List (len N*P, chain index n*P + p) of Lists of (begin, end).
    Equivalent to NeuralSemiCRFInterval(*[x.flatten(-2) for x in scorer(ctx)]).logProb(intervals).
    projection: "merged" (default; shapes merged_eligible takes, else "separate") -- ONE size -> size GEMM instead of the reference's
    size -> 2 D + 1 (merged_weights): the same scores up to fp32 reassociation, half the Linear's flops; "separate": q and k
    projected as the reference does, bit-identical scores to ScaledInnerProductIntervalScorer.forward."""
    assert ctx.dim() == 4
    N, P, T, _ = ctx.shape
    D = scorer.size * scorer.expansionFactor
    _lib.require_gpu(ctx, "ctx")
    if D % 32 != 0 or D > 256:
        S, b = scorer(ctx)                      # contraction sizes the fused kernel does not take: unfused route
        return _nsci.NeuralSemiCRFInterval(S.flatten(-2, -1), b.flatten(-2, -1)).logProb(intervals)
    lin = scorer.map[0]
    W, bias = lin.weight, lin.bias
    x = ctx.float()
    pairs, offsets = _nsci.pack_intervals(intervals, T, N * P, ctx.device)
    fs = 2 | contraction_bits(getattr(scorer, "contraction", "fp32"))
    if fs & (BF16X3 | BWD_BF16X3 | PROJ_BF16X3):
        _bf16x3_selfcheck(x.device)
    if projection not in ("merged", "separate"):
        raise ValueError(f"projection must be 'merged' or 'separate', not {projection!r}")
    if projection == "merged" and merged_eligible(scorer.size, T):
        Wm, bm = merged_weights(W, bias, D)
        return _MergedScorerCRFLogProb.apply(x.contiguous(), Wm, bm, pairs, offsets, N, P, T, D, _lib.LEN_MODES[scorer.lengthScaling], fs)
    if _ScorerLinearPacked.eligible(x, W, bias, D):
        qd, k = _ScorerLinearPacked.apply(x, W, bias, D, 1 if fs & PROJ_BF16X3 else 0)
    else:
        Wqd, bqd = qd_weights(W, bias, D)
        qd, k = _ScorerLinear.apply(x, Wqd, bqd, W[D:2 * D], bias[D:2 * D])
    return _ScorerCRFLogProb.apply(qd, k, pairs, offsets, N, P, T, D, _lib.LEN_MODES[scorer.lengthScaling], fs)
