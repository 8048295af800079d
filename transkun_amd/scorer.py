"""ScaledInnerProductIntervalScorer -- host-side mirror of the reference module
(/root/reference/transkun/LayersTransformer.py:381-441) on top of the HIP interval-score kernel.

Same constructor arguments, same parameter names (`map.0.weight`, `map.0.bias`) so reference
checkpoints load, same outputs: S [T, T, N, P] and the all-zero noise score [T-1, N, P].
The Linear map stays stock GEMMs (hipBLASLt through torch) with a backward of its own (_ScorerLinear: the weight
gradients as batched GEMMs over row chunks); everything after it --
scaling, the per-chain q.k^T contraction, length scaling, diagonal, and the permute to the
CRF layout -- is one kernel that writes the chain-contiguous layout directly.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


def slot_pitch(P: int, T: int, D: int, N: int = 1) -> int:
    """The chain pitch this package's own glue gives N groups of P symbols (include/semicrf_hip.h, "SLOT LAYOUT"): the CRF
    kernels stream 32-chain pieces of the flattened chain axis, which are whole 128-byte lines when the total slot count
    N * pitch is a multiple of 32 (T=691, 90 symbols: 22 % faster at pitch 96).  The smallest such pitch >= P (a multiple
    of 4) where the LDS-tiled scorer kernels and the packed backward apply; P itself (the contiguous layout) otherwise."""
    if (N * P) % 32 == 0 or T < 128 or D not in (64, 128, 256) or _lib.get_impl() != 0:
        return P
    step = max(4, 32 // math.gcd(N, 32))
    return (P + step - 1) // step * step


_SLOT_MAPS = {}


def slot_maps(N: int, P: int, pitch: int, device):
    """(slot of every chain [N*P], offsets gather map [N*pitch + 1]) for chain c = n*P + p <-> slot n*pitch + p: interval
    offsets by slot are offsets_by_chain[map] (a ghost slot is empty: it starts and ends where the next group starts)."""
    key = (N, P, pitch, str(device))
    m = _SLOT_MAPS.get(key)
    if m is None:
        n = torch.arange(N).view(N, 1)
        p = torch.arange(pitch).view(1, pitch)
        real = (n * pitch + torch.arange(P).view(1, P)).reshape(-1)
        off = torch.where(p < P, n * P + p, (n + 1) * P).reshape(-1)
        off = torch.cat([off, torch.tensor([N * P])])
        m = _SLOT_MAPS[key] = (real.to(device), off.to(device))
    return m


def _interval_score_raw(q, k, diag, T: int, C: int, D: int, qscale: float, mode: int, full_square, group: int = 0, pitch: int = 0,
                        rowc=None):
    """q,k: [C,T,D] views with unit stride in d; diag: [C,T] view.  Returns S [T,T,Cs], noise [T-1,Cs]; Cs = C, or with a
    slot layout (group, pitch: include/semicrf_hip.h) C / group * pitch with exact zeros in the ghost slots.
    full_square: False/0 lower triangle + zeros above, True/1 the full square, 2 lower triangle only (the cells with
    begin > end stay uninitialised: for S that only this library's CRF kernels read); | BF16X3 (4): the opt-in three-limb
    bf16 contraction (include/semicrf_hip.h: SEMICRF_SCORE_BF16X3).  rowc: [C,T] view, a per-(chain, end) constant inside the
    contraction (the merged projection, include/semicrf_hip.h: interval_score_fwd_pc)."""
    dev = q.device
    assert q.stride(-1) == 1 and k.stride(-1) == 1
    if not group:
        group = pitch = C
    Cs = C // group * pitch
    # full_square=False: the library computes e >= b and zero-fills the rest itself (half the bytes of torch.zeros) -- once per
    # pooled buffer: like the CRF's dense gradient (transkun_amd/CRF: _GradPool), a score tensor whose memory nobody else
    # references any more and nobody has written in place since keeps the zeros this library wrote above the diagonal, and the
    # next call asks for the lower triangle only (full_square 2): 0.74 GB of writes less at T=1024, 352 chains
    pool_key = None
    if (int(full_square) & 3) == 0 and q.is_cuda:
        S, have_zeros, pool_key = _score_pool().take(T, Cs, dev)
        if have_zeros:
            full_square = (int(full_square) & ~3) | 2
    else:
        S = torch.empty(T, T, Cs, dtype=torch.float32, device=dev)
    if (int(full_square) & 3) == 2 and os.environ.get("SEMICRF_POISON_UNWRITTEN"):
        S.fill_(float("nan"))           # test hook: whatever reads begin > end of a lower-triangle-only S shows up as NaN
    noise = torch.empty(max(T - 1, 0), Cs, dtype=torch.float32, device=dev)
    _lib.ops().interval_score_fwd(q, k, diag, rowc if rowc is not None else diag, C, T, D, q.stride(-2), k.stride(-2), diag.stride(-1),
                                  rowc.stride(-1) if rowc is not None else 0, float(qscale), int(mode), int(full_square), int(group),
                                  int(pitch), S, noise)
    if pool_key is not None:
        _score_pool().give(pool_key, S)
    return S, noise


_SCORE_POOL = None


def _score_pool():
    global _SCORE_POOL
    if _SCORE_POOL is None:
        import importlib
        _SCORE_POOL = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")._GradPool()
    return _SCORE_POOL


_BWD_WS = {}


def bwd_workspace(C: int, T: int, D: int, device) -> torch.Tensor:
    """Workspace of the packed scorer backward (0 bytes when that path does not apply)."""
    key = (C, T, D)
    n = _BWD_WS.get(key)
    if n is None:
        n = _BWD_WS[key] = int(_lib.load().interval_score_bwd_workspace_bytes(C, T, D))
    return torch.empty(n, dtype=torch.uint8, device=device)


BF16X3 = 4      # SEMICRF_SCORE_BF16X3: OR into full_square
LEN_BF16X3 = 16 # SEMICRF_LEN_BF16X3: OR into the backward's length-scaling mode (the two products on the three-limb bf16 kernels)
BWD_BF16X3 = 8  # this package's autograd nodes only (never handed to the library): "the backward of this forward uses LEN_BF16X3"
PROJ_BF16X3 = 32  # this package's autograd nodes only: the projection's forward and input gradient on proj_gemm3.hip (scorer_proj_nn3)
CONTRACTIONS = {"fp32": 0, "bf16x3": BF16X3 | BWD_BF16X3, "bf16x3-fwd": BF16X3, "bf16x3-bwd": BWD_BF16X3,
                "bf16x3-train": BWD_BF16X3 | PROJ_BF16X3, "bf16x3-all": BF16X3 | BWD_BF16X3 | PROJ_BF16X3}


def contraction_bits(name: str) -> int:
    """scorer.contraction -> flag bits of the autograd nodes: 'fp32' (default, exact), 'bf16x3' (forward contraction and backward
    products on the three-limb bf16 kernels), 'bf16x3-fwd' / 'bf16x3-bwd' (one side only: at the model's training shape the
    backward gains 0.23 ms per step and the forward loses 0.13 against the exact kernel, profiles/r05_train_step_*), 'bf16x3-train'
    (the backward products and the merged projection's two NN GEMMs -- the fused route's fastest setting), 'bf16x3-all' (everything)."""
    try:
        return CONTRACTIONS[name]
    except KeyError:
        raise ValueError(f"contraction must be one of {sorted(CONTRACTIONS)}, not {name!r}") from None


_BF16X3_CHECKED = {}


def _bf16x3_selfcheck(device) -> None:
    """First use of a three-limb contraction on `device`: one small problem (1 x 4 chains x T = 256 x size 256) through
    "bf16x3-all" and through the exact fp32 kernels, scores and all three gradients compared at 1e-4 of their largest value (the
    three-limb error bound is ~1e-6 there; a stale or torn operand is O(1)).  Why: the three-limb kernels issue their operand
    loads and wait for them in separate asm statements (csrc/proj_gemm3.hip, scorer_bwd_gemm.hip `landed()`), which is only
    correct while the compiler keeps the loaded registers untouched in between -- true for the build this was measured on, not a
    property a toolchain bump must preserve.  A mismatch raises instead of training on garbage; the exact default never runs this."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if (_BF16X3_CHECKED.get(key) or device.type != "cuda" or torch.cuda.is_current_stream_capturing()
            or torch.is_inference_mode_enabled()):        # (inference tensors cannot require gradients: checked at the first ordinary use)
        return
    _BF16X3_CHECKED[key] = True              # (set first: the check itself goes through forward())
    from . import synth
    size, P, T = 256, 4, 256
    res = {}
    with torch.enable_grad(), torch.random.fork_rng(devices=[]):      # (nn.Linear's initialisation draws from the global CPU generator: not ours to advance)
        for mode in ("fp32", "bf16x3-all"):
            m = ScaledInnerProductIntervalScorer(size).to(device)
            with torch.no_grad():
                m.map[0].weight.copy_(synth.hash_normal(m.map[0].weight.numel(), 901, device).view_as(m.map[0].weight) / 16.0)
                m.map[0].bias.copy_(synth.hash_normal(m.map[0].bias.numel(), 902, device) / 4.0)
            m.contraction = mode
            x = synth.hash_normal(P * T * size, 903, device).view(1, P, T, size).requires_grad_()
            S, _ = m(x)
            w = synth.hash_normal(S.numel(), 904, device).view_as(S)
            (torch.tril(S.permute(2, 3, 0, 1)) * torch.tril(w.permute(2, 3, 0, 1))).sum().backward()
            res[mode] = (S.detach(), x.grad, m.map[0].weight.grad, m.map[0].bias.grad)
    for name, a, b in zip(("scores", "d ctx", "d weight", "d bias"), res["fp32"], res["bf16x3-all"]):
        a, b = torch.tril(a.permute(2, 3, 0, 1)) if name == "scores" else a, torch.tril(b.permute(2, 3, 0, 1)) if name == "scores" else b
        err, ref = float((a - b).abs().max()), float(a.abs().max())
        if not err <= 1e-4 * ref:
            _BF16X3_CHECKED[key] = False
            raise RuntimeError(f"transkun_amd.scorer: the three-limb bf16 kernels disagree with the exact fp32 kernels on this device "
                               f"({name}: max |difference| {err:.3e} against max |value| {ref:.3e}); use scorer.contraction = 'fp32'")


QPAD = 4        # [q | diag | 3 zero columns]: one GEMM instead of a D-wide and a 1-wide one, rows stay 16-byte aligned


def qd_weights(W, bias, D):
    """Rows of the reference's Linear ([q (D) | k (D) | diag (1)], LayersTransformer.py:392-397) regrouped for the
    [q | diag | pad] GEMM.  Differentiable views/cats of the same parameter (state_dict stays map.0.weight/bias)."""
    Wqd = torch.cat([W[:D], W[2 * D:2 * D + 1], W.new_zeros(QPAD - 1, W.shape[1])])
    bqd = torch.cat([bias[:D], bias[2 * D:2 * D + 1], bias.new_zeros(QPAD - 1)])
    return Wqd, bqd


SPLITK_ROWS = 2764       # rows per chunk of the weight-gradient contraction (measured: 2000..4000 rows are equally good)


def _tn_splitk(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dy^T x for [M, n] and [M, k] with M >> n, k: a batched GEMM over row chunks, then a sum over the chunks."""
    M, n = dy.shape
    k = x.shape[1]
    S = M // SPLITK_ROWS
    if S < 2:
        return dy.t().mm(x)
    Mc = M // S
    out = torch.bmm(dy[:S * Mc].view(S, Mc, n).transpose(1, 2), x[:S * Mc].view(S, Mc, k)).sum(0)
    if S * Mc < M:
        out.addmm_(dy[S * Mc:].t(), x[S * Mc:])
    return out


def _column_chunks(D: int, widths):
    out, d0 = [], 0
    while d0 < D:
        w = next((w for w in widths if D - d0 >= w), None)
        if w is None:
            raise ValueError(f"contraction size {D} cannot be cut into column chunks of {widths}")
        out.append((d0, w))
        d0 += w
    return out


def score_backward_hip(g, q, k, C: int, T: int, D: int, qs: float, mode: int, P: int, pitch: int, dq, dk, dd) -> None:
    """dq [C,T,D], dk [C,T,D], dd [C,T] (views with unit stride in d; dd may be None) from the cotangent g [T,T,Cs] -- the autograd
    of LayersTransformer.py:410-433 -- on the HIP kernels for EVERY contraction size.  D is an OUTPUT dimension of both
    products (dq[e,:] = sum_b G[e,b] k[b,:]), so a size beyond what one launch takes (D <= 256; the packed GEMMs: 64 / 128 / 256)
    runs as column chunks of the same operands (D = 512, expansionFactor 2 at size 256: two launches of 256 columns), and a size
    that is no multiple of 32 on zero-padded copies of q and k.  qs stays 1 / sqrt(D) of the whole contraction."""
    ops = _lib.ops()
    dev = g.device
    if D % 32 != 0:
        Dp = (D + 31) // 32 * 32
        qp, kp = F.pad(q, (0, Dp - D)), F.pad(k, (0, Dp - D))
        dqp = torch.empty(C, T, Dp, dtype=torch.float32, device=dev)
        dkp = torch.empty(C, T, Dp, dtype=torch.float32, device=dev)
        score_backward_hip(g, qp, kp, C, T, Dp, qs, mode, P, pitch, dqp, dkp, dd)
        dq.copy_(dqp[..., :D])
        dk.copy_(dkp[..., :D])
        return
    none = torch.empty(0, dtype=torch.float32, device=dev)
    widths = (256, 128, 64) if pitch != P else (256, 128, 64, 32)          # a padded slot pitch needs the packed GEMMs
    for i, (d0, w) in enumerate(_column_chunks(D, widths)):
        ws = bwd_workspace(C, T, w, dev)                                    # 0 bytes: shapes the packed path does not take -> direct kernels
        ddi = dd if (i == 0 and dd is not None) else none
        ops.interval_score_bwd_ws(g, q[..., d0:d0 + w], k[..., d0:d0 + w], C, T, w, q.stride(-2), k.stride(-2), qs, mode, P, pitch,
                                  dq[..., d0:d0 + w], dk[..., d0:d0 + w], ddi, none, dq.stride(-2), dk.stride(-2),
                                  ddi.stride(-1) if ddi.numel() else 1, 0, ws)


# ---- the projection on this library's own GEMM kernels (csrc/proj_gemm.hip) ---------------------------------------------------------
_PROJ_SIZES = (64, 128, 256)


def _proj_ok(a2: torch.Tensor, n_cols: int) -> bool:
    """a2 [M, w]: a row-major fp32 matrix on the GPU whose rows are 16-byte aligned and short enough for 32-bit buffer offsets."""
    return (a2.is_cuda and a2.dtype == torch.float32 and a2.dim() == 2 and a2.is_contiguous() and a2.shape[1] % 4 == 0
            and a2.data_ptr() % 16 == 0 and a2.shape[0] >= 1 and a2.shape[0] * (max(a2.shape[1], n_cols) + 8) * 4 < 2 ** 31
            and _lib.get_impl() == 0 and not os.environ.get("SEMICRF_TORCH_PROJECTION"))


def _stock_gemm_note(what: str, x: torch.Tensor) -> None:
    """One warning per reason: a projection of this module left the library's exact-fp32 GEMMs (csrc/proj_gemm.hip: sizes 64 / 128 /
    256, 16-byte aligned contiguous rows, < 2 GiB per operand) for torch's stock GEMM -- same result to fp32 rounding, a different
    speed; silent until round 5."""
    if x.is_cuda and _lib.get_impl() == 0 and not os.environ.get("SEMICRF_TORCH_PROJECTION"):
        from .CRF.NeuralSemiCRFInterval import _warn_once
        _warn_once("stock_gemm_" + what, f"transkun_amd.scorer: {what} runs on torch's stock GEMM (shape {tuple(x.shape)}: outside the "
                                          "library kernels' sizes {64,128,256} / alignment / 2 GiB limits)")


def _proj_nn(prec: int, K: int, N: int, dev, *args) -> None:
    """scorer_proj_nn, or scorer_proj_nn3 (prec != 0: three-limb bf16 contraction, N == 256) with its workspace."""
    if prec and N == 256:
        n = int(_lib.load().scorer_proj_nn3_workspace_bytes(int(K), int(N)))
        _lib.ops().proj_nn3(*args, torch.empty(n, dtype=torch.uint8, device=dev))
    else:
        _lib.ops().proj_nn(*args)


def proj_forward(x2: torch.Tensor, W: torch.Tensor, b: torch.Tensor, n_main: int, Wt: torch.Tensor = None, prec: int = 0) -> torch.Tensor:
    """y [M, Nout] = x2 W^T + b for the PACKED projection outputs of this package (LayersTransformer.py:388-397, :406-410 regrouped):
    W [Nout, K] holds n_main main rows, then -- if Nout > n_main -- two extra rows ([diag | 0] or [c | diag]) and zero rows.  On the
    library's exact-fp32 GEMM (scorer_proj_nn) where it applies, torch's GEMM otherwise."""
    M, K = x2.shape
    Nout = W.shape[0]
    if not (_proj_ok(x2, Nout) and K in _PROJ_SIZES and n_main in _PROJ_SIZES and (Nout == n_main or Nout >= n_main + 2) and Nout % 4 == 0):
        _stock_gemm_note("the projection's forward", x2)
        return F.linear(x2, W, b)
    if Wt is None or Wt.shape != ((K + 31) // 32 * 32, n_main) or not Wt.is_contiguous():
        Wt = W.new_zeros((K + 31) // 32 * 32, n_main)        # the contraction index as row, whole chunks of 32 rows
        Wt[:K] = W[:n_main].t()
    y = torch.empty(M, Nout, dtype=torch.float32, device=x2.device)
    has2 = Nout > n_main
    w2 = W[n_main:n_main + 2].contiguous() if has2 else b
    b2 = b[n_main:n_main + 2].contiguous() if has2 else b
    _proj_nn(prec, K, n_main, x2.device, x2, K, M, K, Wt, n_main, n_main, y, Nout, b[:n_main].contiguous(), True, w2, b2, has2,
             Nout - n_main - 2 if has2 else 0, False)
    return y


def proj_input_grad(dy2: torch.Tensor, W: torch.Tensor, out: torch.Tensor = None, Wp: torch.Tensor = None, prec: int = 0) -> torch.Tensor:
    """dy2 [M, Nout] W [Nout, K] -> [M, K]; with `out` the product is ADDED to it (the gradient through a second use of the input)."""
    M, Nout = dy2.shape
    K = W.shape[1]
    if not (_proj_ok(dy2, K) and K in _PROJ_SIZES and (out is None or (out.is_contiguous() and out.shape == (M, K)))):
        _stock_gemm_note("the projection's input gradient", dy2)
        return dy2.mm(W) if out is None else out.addmm_(dy2, W)
    if Wp is None or Wp.shape != ((Nout + 31) // 32 * 32, K) or not Wp.is_contiguous():       # (W with zero rows up to whole chunks)
        Wp = W.new_zeros((Nout + 31) // 32 * 32, K)
        Wp[:Nout] = W
    dx = out if out is not None else torch.empty(M, K, dtype=torch.float32, device=dy2.device)
    none = _lib_none(dy2.device)
    _proj_nn(prec, Nout, K, dy2.device, dy2, Nout, M, Nout, Wp, K, K, dx, K, none, False, none, none, False, 0, out is not None)
    return dx


PROJ_TN_BF16X3 = 0x40000000     # SEMICRF_PROJ_TN_BF16X3: OR into proj_tn's total_rows


def proj_weight_grad(dy2: torch.Tensor, x2: torch.Tensor, n_main: int, prec: int = 0):
    """(dW [Nout, K], db [Nout]) = (dy2^T x2, column sums of dy2) with the contraction over the M rows split into slices (partial sums
    in a workspace, fixed summation order)."""
    M, Nout = dy2.shape
    K = x2.shape[1]
    if not (_proj_ok(dy2, K) and _proj_ok(x2, Nout) and K in _PROJ_SIZES and 1 <= n_main <= Nout and (Nout == n_main or Nout >= n_main + 2)):
        _stock_gemm_note("the projection's weight gradient", dy2)
        return _tn_splitk(dy2, x2), dy2.sum(0)
    dW = torch.empty(Nout, K, dtype=torch.float32, device=dy2.device)
    db = torch.empty(Nout, dtype=torch.float32, device=dy2.device)
    key = ("tn", M, n_main, K, dy2.device.index)
    n = _BWD_WS.get(key)
    if n is None:
        n = _BWD_WS[key] = int(_lib.load().scorer_proj_tn_workspace_bytes(M, n_main, K))
    ws = torch.empty(n, dtype=torch.uint8, device=dy2.device)
    _lib.ops().proj_tn(dy2, Nout, M, n_main, n_main if Nout > n_main else -1, Nout | (PROJ_TN_BF16X3 if prec else 0), x2, K, K, dW, K, db, ws)
    return dW, db


_NONE = {}


def _lib_none(device):
    e = _NONE.get(device)
    if e is None:
        e = _NONE[device] = torch.empty(0, dtype=torch.float32, device=device)
    return e


class _ScorerLinear(torch.autograd.Function):
    """The scorer's Linear map (LayersTransformer.py:392-397, :408) as its two GEMMs [q | diag | pad] and k with a backward
    of its own.  Stock autograd computes a weight gradient as ONE GEMM dY^T x: 260 x 256 outputs over a contraction of
    N*P*T = 2.5e5 rows, which hipBLASLt runs on 33 output tiles of the 256 CUs (0.79 / 0.83 ms at 4 x 90 x 691 rows, a
    quarter of the train-shaped step).  Here the rows are cut into chunks of ~2800: one batched GEMM and a sum over the chunks,
    0.28 / 0.32 ms; the two input gradients accumulate inside the second GEMM instead of a separate 255 MB add."""

    @staticmethod
    def forward(ctx, x, Wqd, bqd, Wk, bk):
        ctx.save_for_backward(x, Wqd, Wk)
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        D = Wk.shape[0]
        if _proj_ok(x2, D + QPAD) and K in _PROJ_SIZES and D in _PROJ_SIZES:
            # this library's exact-fp32 GEMMs (csrc/proj_gemm.hip): [q | diag | 0 0 0] with the diagonal term as a dot product of the
            # rows the kernel reads anyway, then k
            qd = proj_forward(x2, Wqd, bqd, D).view(*x.shape[:-1], D + QPAD)
            k = proj_forward(x2, Wk, bk, D).view(*x.shape[:-1], D)
            return qd, k
        _stock_gemm_note("the scorer's Linear", x2)
        return F.linear(x, Wqd, bqd), F.linear(x, Wk, bk)

    @staticmethod
    def backward(ctx, dqd, dk):
        x, Wqd, Wk = ctx.saved_tensors
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        # (an output nobody differentiated arrives as None)
        g1 = dqd.reshape(-1, Wqd.shape[0]) if dqd is not None else None
        g2 = dk.reshape(-1, Wk.shape[0]) if dk is not None else None
        need = ctx.needs_input_grad
        dx = None
        if g1 is not None and not g1.is_contiguous():
            g1 = g1.contiguous()
        if g2 is not None and not g2.is_contiguous():
            g2 = g2.contiguous()
        if need[0]:
            dx = proj_input_grad(g1, Wqd) if g1 is not None else None
            if g2 is not None:
                dx = proj_input_grad(g2, Wk) if dx is None else proj_input_grad(g2, Wk, out=dx)
            dx = dx.view(x.shape) if dx is not None else None
        D = Wk.shape[0]
        dW1 = db1 = dW2 = db2 = None
        if (need[1] or need[2]) and g1 is not None:
            dW1, db1 = proj_weight_grad(g1, x2, D)
        if (need[3] or need[4]) and g2 is not None:
            dW2, db2 = proj_weight_grad(g2, x2, D)
        return (dx, dW1 if need[1] else None, db1 if need[2] else None, dW2 if need[3] else None, db2 if need[4] else None)


class _ScorerLinearPacked(torch.autograd.Function):
    """_ScorerLinear on the Linear's OWN parameters (W [2 D + 1, size], bias), for the shapes the projection kernels take: the
    regrouping ([q | diag | pad] and k) happens in the operand layouts of ONE staging launch (scorer_stage_linear) and the two
    weight gradients are written straight into the rows of ONE dW / dbias -- where qd_weights + _ScorerLinear cost ~25 small torch
    kernels per step (cats, zero fills, staging copies, fill + copy + add per slice gradient)."""

    @staticmethod
    def eligible(x, W, bias, D):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        return (_proj_ok(x2, D + QPAD) and K in _PROJ_SIZES and D in _PROJ_SIZES and W.is_cuda and W.dtype == torch.float32
                and bias.dtype == torch.float32 and tuple(W.shape) == (2 * D + 1, K) and tuple(bias.shape) == (2 * D + 1,))

    @staticmethod
    def forward(ctx, x, W, bias, D, prec=0):
        # prec = 1: the four NN GEMMs on the three-limb bf16 kernels (csrc/proj_gemm3.hip; scorer.contraction = "bf16x3-train" / "bf16x3-all")
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        Wc, bc = W.contiguous(), bias.contiguous()
        dev = x.device
        rows_pad = (D + QPAD + 31) // 32 * 32
        BT = torch.empty(K, 2 * D, dtype=torch.float32, device=dev)
        Wqd = torch.empty(rows_pad, K, dtype=torch.float32, device=dev)
        w2 = torch.empty(2, K, dtype=torch.float32, device=dev)
        b2 = torch.empty(2, dtype=torch.float32, device=dev)
        ops = _lib.ops()
        ops.stage_linear(Wc, bc, D, K, rows_pad, BT, Wqd, w2, b2)
        qd = torch.empty(M, D + QPAD, dtype=torch.float32, device=dev)
        k = torch.empty(M, D, dtype=torch.float32, device=dev)
        none = _lib_none(dev)
        flat = BT.view(-1)
        _proj_nn(prec, K, D, dev, x2, K, M, K, flat, 2 * D, D, qd, D + QPAD, bc[:D], True, w2, b2, True, QPAD - 2, False)
        _proj_nn(prec, K, D, dev, x2, K, M, K, flat[D:], 2 * D, D, k, D, bc[D:2 * D], True, none, none, False, 0, False)
        ctx.save_for_backward(x, Wc, Wqd)
        ctx.D = D
        ctx.prec = prec
        return qd.view(*x.shape[:-1], D + QPAD), k.view(*x.shape[:-1], D)

    @staticmethod
    def backward(ctx, dqd, dk):
        x, Wc, Wqd = ctx.saved_tensors
        D = ctx.D
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        dev = x.device
        ops = _lib.ops()
        none = _lib_none(dev)
        g1 = dqd.reshape(M, D + QPAD).contiguous() if dqd is not None else None
        g2 = dk.reshape(M, D).contiguous() if dk is not None else None
        # (a contiguous VIEW may start anywhere: the kernels want 16-byte aligned rows)
        if g1 is not None and g1.data_ptr() % 16:
            g1 = g1.clone()
        if g2 is not None and g2.data_ptr() % 16:
            g2 = g2.clone()
        need = ctx.needs_input_grad
        dx = None
        if need[0] and (g1 is not None or g2 is not None):
            dx = torch.empty(M, K, dtype=torch.float32, device=dev)
            if g1 is not None:
                _proj_nn(ctx.prec, D + QPAD, K, dev, g1, D + QPAD, M, D + QPAD, Wqd, K, K, dx, K, none, False, none, none, False, 0, False)
            if g2 is not None:                                                 # B = the k rows of W as they lie
                _proj_nn(ctx.prec, D, K, dev, g2, D, M, D, Wc.view(-1)[D * K:], K, K, dx, K, none, False, none, none, False, 0, g1 is not None)
            dx = dx.view(x.shape)
        dW = db = None
        if need[1] or need[2]:
            # one dW / dbias: the q call writes rows 0 .. D + QPAD - 1 (its extra row D is the diagonal row's gradient: moved to row
            # 2 D), then the k call writes rows D .. 2 D - 1 over what the first one left there
            rows = max(2 * D + 1, D + QPAD)
            dW = torch.zeros(rows, K, dtype=torch.float32, device=dev) if (g1 is None or g2 is None) else torch.empty(rows, K, dtype=torch.float32, device=dev)
            db = torch.zeros(rows, dtype=torch.float32, device=dev) if (g1 is None or g2 is None) else torch.empty(rows, dtype=torch.float32, device=dev)
            if g1 is not None:
                key = ("tn", M, D, K, dev.index)              # (the geometry depends on that device's compute-unit count)
                n = _BWD_WS.get(key)
                if n is None:
                    n = _BWD_WS[key] = int(_lib.load().scorer_proj_tn_workspace_bytes(M, D, K))
                ws = torch.empty(n, dtype=torch.uint8, device=dev)
                ops.proj_tn(g1, D + QPAD, M, D, D, (D + QPAD) | (PROJ_TN_BF16X3 if ctx.prec else 0), x2, K, K, dW, K, db, ws)
                dW[2 * D].copy_(dW[D])
                db[2 * D:2 * D + 1].copy_(db[D:D + 1])
                if g2 is None:
                    dW[D:2 * D].zero_(); db[D:2 * D].zero_()
            if g2 is not None:
                key = ("tn", M, D, K, dev.index)              # (the geometry depends on that device's compute-unit count)
                n = _BWD_WS.get(key)
                if n is None:
                    n = _BWD_WS[key] = int(_lib.load().scorer_proj_tn_workspace_bytes(M, D, K))
                ws = torch.empty(n, dtype=torch.uint8, device=dev)
                ops.proj_tn(g2, D, M, D, -1, D | (PROJ_TN_BF16X3 if ctx.prec else 0), x2, K, K, dW.view(-1)[D * K:], K, db[D:], ws)
            dW, db = dW[:2 * D + 1], db[:2 * D + 1]
        return dx, dW if need[1] else None, db if need[2] else None, None, None


class _IntervalScore(torch.autograd.Function):
    """S = lenscale * (q*qscale) k^T + diag, chain-minor layout; forward and backward are HIP kernels
    (every contraction size: wider than 256 runs as column chunks, a size that is no multiple of 32 is zero-padded).
    qd: [N,P,T,D+QPAD] = [q | diag | zeros]; k: [N,P,T,D]."""

    @staticmethod
    def forward(ctx, qd, k, N, P, T, D, mode, full_square, pitch=0):
        # pitch > P: the slot layout -- S comes back as [T, T, N, pitch] (zeros in the slots P.. of every segment)
        C = N * P
        pitch = pitch or P
        qd3, k3 = qd.reshape(C, T, D + QPAD), k.reshape(C, T, D)
        qscale = 1.0 / math.sqrt(D)
        S, noise = _interval_score_raw(qd3[..., :D], k3, qd3[..., D], T, C, D, qscale, mode, int(full_square) & ~(BWD_BF16X3 | PROJ_BF16X3), P, pitch)
        ctx.save_for_backward(qd3, k3)
        ctx.meta = (N, P, T, D, mode | (LEN_BF16X3 if int(full_square) & BWD_BF16X3 else 0), (int(full_square) & 3) == 1, pitch)
        return S.view(T, T, N, pitch), noise.view(max(T - 1, 0), N, pitch)

    @staticmethod
    def backward(ctx, dS, dnoise):
        qd3, k = ctx.saved_tensors
        N, P, T, D, mode, full, pitch = ctx.meta
        C = N * P
        qs = 1.0 / math.sqrt(D)
        q = qd3[..., :D]
        # HIP kernels for every shape: dq/dk from dS in its native [T,T,C] layout on the matrix cores (exact fp32), written
        # straight into the gradient of [q | diag | pad]
        g = dS.reshape(T, T, N * pitch)
        if g.dtype != torch.float32:
            g = g.float()
        if not g.is_contiguous():
            g = g.contiguous()
        dqd = torch.empty(C, T, D + QPAD, dtype=torch.float32, device=g.device)
        dqd[..., D + 1:] = 0
        dq, dd = dqd[..., :D], dqd[..., D]
        dk = torch.empty(C, T, D, dtype=torch.float32, device=g.device)
        score_backward_hip(g, q, k, C, T, D, qs, mode, P, pitch, dq, dk, dd)
        if full:
            # the reference's full square (fullSquare=True): the cells begin > end were computed as well, S[e,b] = qs <q_e, k_b>
            # len(b-e), so a consumer's gradient there flows too.  Transposed, that strict upper triangle is a strict lower
            # triangle with the roles of q and k exchanged: the same kernels once more (a rarely used compatibility path: it
            # pays a transposed copy of dS).
            gT = g.transpose(0, 1).contiguous()
            gT.diagonal(dim1=0, dim2=1).zero_()
            dq2 = torch.empty(C, T, D, dtype=torch.float32, device=g.device)
            dk2 = torch.empty(C, T, D, dtype=torch.float32, device=g.device)
            score_backward_hip(gT, k, q, C, T, D, qs, mode, P, pitch, dk2, dq2, None)
            dq.add_(dq2)
            dk.add_(dk2)
        return (dqd.view(N, P, T, D + QPAD), dk.view(N, P, T, D), None, None, None, None, None, None, None)


class ScaledInnerProductIntervalScorer(nn.Module):
    def __init__(self, size, expansionFactor=1, dropoutProb=0.0, withScoreEps=False, lengthScaling="linear"):
        super().__init__()
        self.size = size
        if withScoreEps:
            # the reference allocates one extra output it never uses (LayersTransformer.py:392-395)
            self.map = nn.Sequential(nn.Linear(size, 2 * size * expansionFactor + 1 + 1))
        else:
            self.map = nn.Sequential(nn.Linear(size, 2 * size * expansionFactor + 1))
        self.dropout = nn.Dropout(dropoutProb)      # defined but never applied, as in the reference (:397)
        self.expansionFactor = expansionFactor
        if lengthScaling not in _lib.LEN_MODES:
            raise Exception("Unrecognized lengthScaling")
        self.lengthScaling = lengthScaling
        self.withScoreEps = withScoreEps
        self.fullSquare = False   # True: also materialise e<b like the reference (the CRF never reads it); 2: leave e<b
                                  # uninitialised (only for S that goes straight into this package's CRF)
        self.slotPitch = None      # an int > P (a multiple of 4): forward() returns S as [T, T, N, slotPitch] and the noise score as
                                  # [T-1, N, slotPitch] with zeros in the slots P.. of every segment (include/semicrf_hip.h, "SLOT
                                  # LAYOUT") -- for callers that hand flatten(-2, -1) of both straight to NeuralSemiCRFInterval and
                                  # drop the ghost chains' results; None: the reference's [T, T, N, P]
        self.contraction = "fp32"  # opt-in contraction on the bf16 matrix instructions (contraction_bits): operands split exactly into
                                  # three bf16 limbs, six limb products, fp32 accumulation -- fp32-grade (|error| <= 2^-21 *
                                  # sum_d |q_d k_d| * scale), NOT bit-identical to "fp32".  "bf16x3-fwd": the forward contraction
                                  # only (the backward stays the exact fp32 one); "bf16x3-bwd": the backward's two products only;
                                  # "bf16x3": both (since round 5 -- until then this name meant the forward only: gradients under
                                  # "bf16x3" are fp32-grade now, no longer bit-identical to "fp32"; use "bf16x3-fwd" for the old
                                  # meaning); "bf16x3-train": backward + the projection's NN GEMMs; "bf16x3-all": everything.
                                  # The first use of any of them on a device runs a self-check against the exact kernels
                                  # (_bf16x3_selfcheck).

    def forward(self, ctx):
        # ctx: [N, P, T, size]
        assert ctx.dim() == 4
        N, P, T, _ = ctx.shape
        D = self.size * self.expansionFactor
        _lib.require_gpu(ctx, "ctx")
        # the Linear map as two GEMMs over regrouped rows of the same parameter (state_dict stays map.0.weight/bias):
        # [q | diag | pad] and k come out with 16-byte aligned rows, no split copy of a packed [.., 2D+1] tensor and no
        # 1-wide GEMM for the diagonal term (0.9 ms fwd+bwd on its own at T=1024, NBatch=352)
        lin = self.map[0]
        W, bias = lin.weight, lin.bias
        x = ctx.float()
        fs = int(self.fullSquare) | contraction_bits(self.contraction)
        if fs & (BF16X3 | BWD_BF16X3 | PROJ_BF16X3):
            _bf16x3_selfcheck(x.device)
        if _ScorerLinearPacked.eligible(x, W, bias, D):
            qd, k = _ScorerLinearPacked.apply(x, W, bias, D, 1 if fs & PROJ_BF16X3 else 0)
        else:
            Wqd, bqd = qd_weights(W, bias, D)
            qd, k = _ScorerLinear.apply(x, Wqd, bqd, W[D:2 * D], bias[D:2 * D])
        S, b = _IntervalScore.apply(qd, k, N, P, T, D, _lib.LEN_MODES[self.lengthScaling], fs, int(self.slotPitch or 0))
        return S, b
