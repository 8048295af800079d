"""CPU oracle for the semi-CRF hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (transkun_amd) never imports it and never falls back to it.

Two things live here:

* ctypes bindings to oracle/_build/libsemicrf_oracle.so (semicrf_oracle.c, the scalar
  plain-C restatement; see that file's header for the reference citations and for how
  it is pinned against golden vectors produced by the reference itself).
* `oploop_*`: a torch-CPU op-loop restatement of the same recurrences (row-by-row
  logsumexp / max over [i, B] slices, the way the reference's TorchScript loops work,
  /root/reference/transkun/CRF/NeuralSemiCRFInterval.py:402-410 and :31-51).  It exists
  only to be *timed* as bench.py's "cpu_baseline" (kind "port") on the GPU box's host
  cores, where the reference's own Python is not available.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsemicrf_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_i64p = ctypes.POINTER(ctypes.c_long)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "semicrf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libsemicrf_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_viterbi_backward.restype = ctypes.c_long
        _lib.oracle_viterbi_forward.restype = ctypes.c_long
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a: np.ndarray, t=_f32p):
    return a.ctypes.data_as(t)


def alpha(score, noise) -> Tuple[np.ndarray, np.ndarray]:
    """Forward sweep: returns (v[T,B], logZ[B])."""
    score, noise = _f32(score), _f32(noise)
    T, B = score.shape[0], score.shape[2]
    v = np.empty((T, B), np.float32)
    logz = np.empty((B,), np.float32)
    lib().oracle_alpha(_p(score), _p(noise), T, B, _p(v), _p(logz))
    return v, logz


def beta(score, noise) -> np.ndarray:
    score, noise = _f32(score), _f32(noise)
    T, B = score.shape[0], score.shape[2]
    q = np.empty((T, B), np.float32)
    lib().oracle_beta(_p(score), _p(noise), T, B, _p(q))
    return q


def forward_backward(score, noise):
    """Returns (logZ[B], grad[T,T,B], gradNoise[T-1,B], v[T,B], q[T,B])."""
    score, noise = _f32(score), _f32(noise)
    T, B = score.shape[0], score.shape[2]
    logz = np.empty((B,), np.float32)
    grad = np.empty((T, T, B), np.float32)
    gn = np.empty((max(T - 1, 0), B), np.float32)
    v = np.empty((T, B), np.float32)
    q = np.empty((T, B), np.float32)
    lib().oracle_forward_backward(_p(score), _p(noise), T, B, _p(logz), _p(grad), _p(gn), _p(v), _p(q))
    return logz, grad, gn, v, q


def forward_backward_f64(score, noise):
    """Double-precision truth of forward_backward (same recurrences, REAL=double):
    returns (logZ[B], grad[T,T,B], gradNoise[T-1,B], v[T,B], q[T,B]) as float64."""
    score, noise = _f32(score), _f32(noise)
    T, B = score.shape[0], score.shape[2]
    dp = ctypes.POINTER(ctypes.c_double)
    logz = np.empty((B,), np.float64)
    grad = np.empty((T, T, B), np.float64)
    gn = np.empty((max(T - 1, 0), B), np.float64)
    v = np.empty((T, B), np.float64)
    q = np.empty((T, B), np.float64)
    lib().oracle_forward_backward_f64(_p(score), _p(noise), T, B, _p(logz, dp), _p(grad, dp), _p(gn, dp),
                                      _p(v, dp), _p(q, dp))
    return logz, grad, gn, v, q


def alpha_f64(score, noise):
    score, noise = _f32(score), _f32(noise)
    T, B = score.shape[0], score.shape[2]
    dp = ctypes.POINTER(ctypes.c_double)
    v = np.empty((T, B), np.float64)
    logz = np.empty((B,), np.float64)
    lib().oracle_alpha_f64(_p(score), _p(noise), T, B, _p(v, dp), _p(logz, dp))
    return v, logz


def _unpack(pairs: np.ndarray, offsets: np.ndarray) -> List[List[Tuple[int, int]]]:
    flat = pairs.reshape(-1, 2).tolist()
    off = offsets.tolist()
    return [[(a, b) for a, b in flat[off[c]:off[c + 1]]] for c in range(len(off) - 1)]


def viterbi(score, noise, forcedStartPos: Optional[Sequence[int]] = None, forward: bool = False):
    """decode(): List[List[(begin, end)]], like the reference's viterbi / viterbiBackward."""
    score, noise = _f32(score), _f32(noise)
    T, B = score.shape[0], score.shape[2]
    cap = 2 * T * B + B
    pairs = np.empty((cap * 2,), np.int32)
    offsets = np.empty((B + 1,), np.int64)
    start = None
    sp = None
    if forcedStartPos is not None:
        start = np.ascontiguousarray(np.asarray(forcedStartPos, dtype=np.int32))
        assert start.shape == (B,)
        sp = _p(start, _i32p)
    fn = lib().oracle_viterbi_forward if forward else lib().oracle_viterbi_backward
    n = fn(_p(score), _p(noise), T, B, sp, _p(pairs, _i32p), ctypes.c_long(cap), _p(offsets, _i64p))
    assert n >= 0
    return _unpack(pairs[: 2 * n], offsets)


def pack_intervals(intervals: Sequence[Sequence[Tuple[int, int]]]):
    """List[List[(begin,end)]] -> (pairs int32 [K,2], offsets int64 [B+1])."""
    counts = [len(x) for x in intervals]
    offsets = np.zeros((len(intervals) + 1,), np.int64)
    np.cumsum(counts, out=offsets[1:])
    flat = [p for lst in intervals for p in lst]
    pairs = np.asarray(flat, dtype=np.int32).reshape(-1, 2)
    return np.ascontiguousarray(pairs), offsets


def eval_path(intervals, score, noise) -> np.ndarray:
    score, noise = _f32(score), _f32(noise)
    T, B = score.shape[0], score.shape[2]
    pairs, offsets = pack_intervals(intervals)
    out = np.empty((B,), np.float32)
    lib().oracle_eval_path(_p(score), _p(noise), T, B, _p(pairs, _i32p), _p(offsets, _i64p), _p(out))
    return out


def interval_score(q, k, diag, length_scaling: str = "linear") -> np.ndarray:
    """q,k: [C,T,D] (q not yet divided by sqrt(D)); diag [C,T] -> S [T,T,C]."""
    q, k, diag = _f32(q), _f32(k), _f32(diag)
    C, T, D = q.shape
    mode = {"linear": 0, "sqrt": 1, "none": 2}[length_scaling]
    S = np.empty((T, T, C), np.float32)
    lib().oracle_interval_score(_p(q), _p(k), _p(diag), C, T, D, mode, _p(S))
    return S


# --------------------------------------------------------------------------------------
# torch-CPU op-loop port (timed as cpu_baseline only)
# --------------------------------------------------------------------------------------

def oploop_forward_backward(score, noise):
    """Row-by-row torch restatement of forward_backward (NSCI:375-456): returns
    (logZ, grad, gradNoise).  Same op pattern as the reference: a concatenated 2B-chain
    sweep, then dense marginals."""
    import torch
    import torch.nn.functional as F

    T, _, B = score.shape
    sflip = torch.flip(score, dims=[0, 1]).transpose(0, 1)
    nflip = torch.flip(noise, (0,))
    sfb = torch.cat([score, sflip], dim=-1)
    nfb = torch.cat([noise, nflip], dim=-1)
    sp = F.softplus(torch.diagonal(sfb, dim1=0, dim2=1)).transpose(-1, -2)
    v = score.new_zeros(T, 2 * B)
    v[0] = sp[0]
    for i in range(1, T):
        v[i] = torch.logaddexp(v[i - 1] + nfb[i - 1], torch.logsumexp(v[:i] + sfb[i, :i], dim=0))
        v[i] += sp[i]
    v, q = torch.chunk(v, 2, dim=-1)
    q = torch.flip(q, (0,))
    logz = v[-1]
    del sfb, nfb
    grad = v.unsqueeze(0) + ((q.unsqueeze(1) - logz) + score)
    d = torch.diagonal(grad, dim1=0, dim2=1)
    d -= (2 * F.softplus(torch.diagonal(score, dim1=0, dim2=1)))
    mask = torch.ones(T, T).tril().unsqueeze(-1)
    grad = (grad * mask).exp() * mask
    gn = (v[:-1] + q[1:] + noise - logz).exp()
    return logz, grad, gn


def oploop_viterbi_backward(score, noise, forcedStartPos=None):
    """Row-by-row torch restatement of viterbiBackward (NSCI:13-104), host backtrack."""
    import torch

    T, _, B = score.shape
    q = torch.zeros(T, B)
    ptr = []
    st = score.transpose(0, 1).contiguous()
    q[T - 1] = score[T - 1, T - 1] * (score[T - 1, T - 1] > 0)
    for i in range(1, T):
        tmp = torch.cat([q[T - i:T - i + 1] + noise[T - i - 1], q[T - i:] + st[T - i - 1, T - i:]], dim=0)
        cur, sel = tmp.max(dim=0)
        ptr.append(sel - 1)
        d = score[T - i - 1, T - i - 1]
        q[T - i - 1] = cur + d * (d > 0)
    ptr_l = torch.stack(ptr, 0).flip(0).t().contiguous().tolist()   # [B][T-1], ptr_l[c][t]
    diag = (torch.diagonal(score, dim1=0, dim2=1) > 0).tolist()      # [B][T]
    if forcedStartPos is None:
        forcedStartPos = [0] * B
    out = []
    for c in range(B):
        j = forcedStartPos[c]
        cur = []
        pc, dc = ptr_l[c], diag[c]
        while j < T - 1:
            if dc[j]:
                cur.append((j, j))
            s = pc[j]
            if s < 0:
                j += 1
            else:
                cur.append((j, s + j + 1))
                j = s + j + 1
        if dc[T - 1]:
            cur.append((T - 1, T - 1))
        out.append(cur)
    return out


def fetch_interval_features(ctx, intervals_batch):
    """CPU restatement of TransKun.fetchIntervalFeaturesBatch (ModelTransformer.py:501-532) + listToIdx (Util.py:173-176).
    ctx: numpy [N, SYM, T, D]; intervals_batch: per segment, per symbol, list of (begin, end).
    Returns (ctx_a_all [K,D], ctx_b_all [K,D], symIdx_all [K] int64, scatterIdx_all [K] int64) in the reference's order
    (segments in order, symbols in order, intervals in list order).  Test infrastructure only.  Pinned against the
    reference's own method by tools/make_golden.py (tests/golden/attr_*.npz)."""
    import numpy as np
    N, SYM, T, D = ctx.shape
    a, b, sym, sc = [], [], [], []
    for idx, cur in enumerate(intervals_batch):                  # :509
        for s_i, lst in enumerate(cur):                          # listToIdx: symbol index repeated per interval
            for (bg, en) in lst:
                a.append(ctx[idx, s_i, bg]); b.append(ctx[idx, s_i, en])      # :522-523: index begin/end + symIdx*T
                sym.append(s_i); sc.append(idx * SYM + s_i)                   # :514, :516
    return (np.stack(a).astype(np.float32), np.stack(b).astype(np.float32),
            np.asarray(sym, np.int64), np.asarray(sc, np.int64))


# --------------------------------------------------------------------------------------
# transcription segment loop -- test infrastructure only
# --------------------------------------------------------------------------------------

def segment_events(lists, nSym, ofValue, ofPresence, lastFrameIdx, frameDur, beginTime, stepFrames):
    """CPU restatement of the per-interval event assembly of TransKun.transcribeFrames (ModelTransformer.py:672-718) and of
    the hand-off of TransKun.transcribe (:789-800), in Python floats like the reference.
    lists: decoded intervals per chain (chain = segment * nSym + symbol); ofValue [K][2] floats and ofPresence [K][2] bools
    in list order; beginTime: one float per segment.  Returns (events, lastP, nextStart): events = per chain a list of
    (start, end, hasOnset, hasOffset) already shifted by the segment's begin time and clamped (:794-800).
    Pinned against the reference's own loop by tools/make_golden.py (tests/golden/transcribe_*.npz)."""
    events, lastP, nextStart = [], [], []
    n = 0
    for c, cur in enumerate(lists):
        bt = beginTime[c // nSym]
        lastEnd = 0                                            # :678
        curLastP = 0                                           # :679
        out = []
        for (b, e) in cur:
            start = (b + float(ofValue[n][0])) * frameDur      # :684
            end = (e + float(ofValue[n][1])) * frameDur        # :685
            hasOnset = (b > 0) or bool(ofPresence[n][0])       # :689
            hasOffset = (e < lastFrameIdx) or bool(ofPresence[n][1])   # :690
            start = max(start, lastEnd)                        # :694
            end = max(end, start + 1e-8)                       # :695
            lastEnd = end                                      # :696
            if hasOffset:                                      # :708-709
                curLastP = e
            s2 = start + bt; e2 = end + bt                     # transcribe :795-796
            s2 = max(s2, 0); e2 = max(e2, s2)                  # :798-799
            out.append((s2, e2, hasOnset, hasOffset))
            n += 1
        events.append(out)
        lastP.append(curLastP)                                 # :718
        nextStart.append(max(curLastP - stepFrames, 0))        # transcribe :789-791
    return events, lastP, nextStart


def merge_segments(per_segment_events, mergeIncompleteEvent=True):
    """CPU restatement of the cross-segment merge of TransKun.transcribe (:803-843, without resolveOverlapping).
    per_segment_events: for every segment a list of (start, end, pitch, velocity, hasOnset, hasOffset) in the reference's
    order (sorted by (start, end, pitch) within the segment).  Returns the surviving events as tuples."""
    by_type = {}
    for seg in per_segment_events:
        for e in seg:
            e = list(e)
            lst = by_type.setdefault(e[2], [])
            if mergeIncompleteEvent and lst:
                last = lst[-1]
                if e[0] < last[1]:
                    if e[4]:
                        lst[-1] = e
                    else:
                        last[5] = e[5]
                        last[1] = max(e[1], last[1])
                    continue
            if e[4]:
                lst.append(e)
    for lst in by_type.values():
        if lst:
            lst[-1][5] = True
    return [tuple(e) for lst in by_type.values() for e in lst if e[5]]
