"""MI355X-native Neural Semi-CRF interval layer -- host-side mirror of the reference class.

Drop-in for /root/reference/transkun/CRF/NeuralSemiCRFInterval.py (class at :553-588): same
constructor, method names, defaults, argument meaning and result types.  All arithmetic runs
in hand-written gfx950 HIP kernels behind the C ABI of include/semicrf_hip.h; this file only
validates shapes, marshals the Python interval lists to/from packed int32 buffers and wires the
kernels into autograd.  CPU tensors are dispatched to the shim's own host kernels (csrc/cpu_ops.cpp; config #1,
crfMinimalExample on the CPU); a GPU tensor is never computed anywhere but in the HIP kernels.

Arithmetic is fp32 (the reference's callers are fp32 throughout); other floating dtypes are computed in fp32 and the
results and gradients are cast back to the input dtype, as the reference preserves it.

What differs from the reference, invisibly at the API:
  * computeLogZ saves alpha (v [T,B]) and logZ instead of the dense marginals [T,T,B]
    (reference :463-464) and recomputes them in backward, fused with the beta sweep.
  * logProb() is ONE autograd node: its backward writes gout*(onehot(path) - marginal) in a single
    pass instead of summing two dense [T,T,B] gradients.
  * decode() backtracks on the device; only the packed (begin,end) pairs cross PCIe.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib

Intervals = List[List[Tuple[int, int]]]


# --------------------------------------------------------------------------------------
# marshalling
# --------------------------------------------------------------------------------------

def _check_inputs(score: torch.Tensor, noiseScore: torch.Tensor):
    # same asserts as the reference (:209-215, :377-382, :510-511) -> AssertionError
    assert len(score.shape) == 3
    assert score.shape[0] == score.shape[1]
    assert len(noiseScore.shape) == 2
    T, B = score.shape[0], score.shape[2]
    assert noiseScore.shape[0] == T - 1
    assert noiseScore.shape[1] == B
    _lib.require_device(score, "score")
    _lib.require_device(noiseScore, "noiseScore")
    assert score.device == noiseScore.device
    return T, B


def _prep(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_SIDE = {}


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=device)
    return st


def _ready(pairs) -> None:
    """Intervals packed with overlap=True travel on a side stream: make the current stream wait for them (once)."""
    ev = getattr(pairs, "_semicrf_ready", None)
    if ev is not None:
        torch.cuda.current_stream(pairs.device).wait_event(ev)
        pairs._semicrf_ready = None


def pack_intervals(intervals: Sequence[Sequence[Tuple[int, int]]], T: int, B: int, device, overlap: bool = False):
    """List[List[(begin,end)]] (len B) -> (pairs int32 [K,2], offsets int32 [B+1]) on `device`.
    One pass in C (csrc/pymarshal.c) into pinned host buffers, then two asynchronous copies.
    overlap=True: the copies go to a side stream (they need nothing from the GPU and would otherwise sit in front of the
    sweep that the caller enqueues next: 20-100 us per call); the caller must pass `pairs` to _ready() before its first use."""
    assert len(intervals) == B, f"expected {B} interval lists, got {len(intervals)}"
    mm = _lib.marshal()
    K = int(mm.count(intervals))
    pin = torch.device(device).type == "cuda"
    pairs_h = torch.empty(max(K, 1), 2, dtype=torch.int32, pin_memory=pin)
    offsets_h = torch.empty(B + 1, dtype=torch.int32, pin_memory=pin)
    if K == 0:
        pairs_h.zero_()
    k = mm.pack_into(intervals, pairs_h.data_ptr(), max(K, 1), offsets_h.data_ptr(), T)   # IndexError when out of range
    assert k == K
    if overlap and pin:
        main = torch.cuda.current_stream(device)
        side = _side_stream(device)
        with torch.cuda.stream(side):
            pairs_d = pairs_h.to(device, non_blocking=True)
            offsets_d = offsets_h.to(device, non_blocking=True)
            ev = side.record_event()
        pairs_d.record_stream(main); offsets_d.record_stream(main)      # allocated in the side stream's pool, used on `main`
        pairs_d._semicrf_ready = ev
    else:
        pairs_d = pairs_h.to(device, non_blocking=True)
        offsets_d = offsets_h.to(device, non_blocking=True)
    pairs_d._semicrf_K = K          # number of real intervals (the tensor holds one dummy row when K == 0)
    return pairs_d, offsets_d


def unpack_intervals(pairs_host: torch.Tensor, offsets_host: torch.Tensor, T: int = 1 << 30) -> Intervals:
    """packed int32 [K,2] + offsets [B+1] (host) -> List[List[Tuple[int,int]]] (the reference's result type), built in C
    with one shared int object per frame index (csrc/pymarshal.c); the cyclic GC is paused meanwhile (hundreds of
    thousands of fresh tuples would otherwise trigger several full collections)."""
    import gc
    pairs_host = pairs_host.contiguous()
    offsets_host = offsets_host.contiguous()
    assert pairs_host.dtype == torch.int32 and offsets_host.dtype == torch.int32
    B = offsets_host.numel() - 1
    total = int(offsets_host[-1]) if B >= 0 else 0
    assert pairs_host.numel() >= 2 * total
    if T >= 1 << 30:
        T = int(pairs_host[:total].max()) + 1 if total else 1
    was = gc.isenabled()
    gc.disable()
    try:
        return _lib.marshal().unpack(pairs_host.data_ptr(), offsets_host.data_ptr(), B, T)
    finally:
        if was:
            gc.enable()


# --------------------------------------------------------------------------------------
# raw kernel calls (no autograd)
# --------------------------------------------------------------------------------------

_DEBUG_WS = []      # test/diagnostic hook: when non-None-appendable and env SEMICRF_DEBUG_KEEP_WS is set, keeps workspaces


def _odd_pad(score) -> bool:
    """The persistent kernels take any NBatch >= 2 (an odd one natively since round 2: 4-byte aligned 16-byte accesses).
    A single chain is run with one all-zero ghost chain appended instead of on the ~100x slower row-sequential kernels;
    the ghost chain's outputs are dropped."""
    return score.is_cuda and score.shape[2] == 1 and score.shape[0] >= 2 and _lib.get_impl() == 0


def _pad1(t):
    return torch.nn.functional.pad(t, (0, 1))


def _logz_fwd_raw(score, noise, want_v: bool):
    if _odd_pad(score):
        logz, v = _logz_fwd_raw(_pad1(score), _pad1(noise), want_v)
        return logz[:-1].contiguous(), (v[:, :-1].contiguous() if v is not None else None)
    T, B = score.shape[0], score.shape[2]
    logz = torch.empty(B, dtype=torch.float32, device=score.device)
    v = torch.empty((T, B) if want_v else (0,), dtype=torch.float32, device=score.device)
    ws = _lib.leased_workspace(_lib.OP_LOGZ_FWD, T, B, score.device)
    _lib.ops().logz_fwd(score, noise, logz, v, want_v, ws)
    if os.environ.get("SEMICRF_DEBUG_KEEP_WS"):
        _DEBUG_WS[:] = [ws]
    return logz, (v if want_v else None)


# --------------------------------------------------------------------------------------
# the dense gradient's buffers
# --------------------------------------------------------------------------------------
# The reference's contract is a DENSE [T,T,B] gradient whose upper triangle (begin > end) is exactly zero
# (NeuralSemiCRFInterval.py:436-440, :469-472).  Those zeros are a third of the bytes the gradient sweep moves (0.74 of
# 2.22 GB at T=1024, NBatch=352) and they never change.  The pool below keeps the MEMORY of gradient tensors this library
# has fully written once (zeros included) and hands it out again -- with SEMICRF_GRAD_UPPER_IS_ZERO, so that the sweep skips
# the zeros -- only while both hold:
#   * nothing else references the storage (every tensor that aliased it -- score.grad, views, what autograd kept -- is gone);
#   * the version counter the handed-out tensor shares with all of its aliases is where the library's last write left it:
#     any in-place operation on the gradient by anybody (grad.add_(..), zero_(), an optimizer, AccumulateGrad adding a second
#     gradient into it) moves the counter, and the buffer is then written in full (zeros included) on its next use.
# The pool keeps `P.detach()` (same storage, same version counter, its own TensorImpl) and not P itself: autograd takes a
# gradient as .grad without a copy only when nobody else holds the tensor object.
GRAD_UPPER_IS_ZERO = 1          # include/semicrf_hip.h: SEMICRF_GRAD_UPPER_IS_ZERO


# torch._C._storage_Use_Count is a private binding (present in every torch 2.x this was run on): without it the pool cannot know
# that nobody else holds a buffer, so it stays off -- one warning, every gradient is then written in full like before round 4.
_USE_COUNT = getattr(torch._C, "_storage_Use_Count", None)
_WARNED = set()


def _warn_once(tag: str, msg: str) -> None:
    if tag not in _WARNED:
        _WARNED.add(tag)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _storage_users(t: torch.Tensor) -> int:
    st = t.untyped_storage()
    return _USE_COUNT(st._cdata) - 1           # minus the temporary `st`


def _empty_or_trim(shape, device):
    """torch.empty; on an out-of-memory error the pools (which hold memory outside the caching allocator's reach) are released
    and the allocation is tried once more."""
    try:
        return torch.empty(shape, dtype=torch.float32, device=device)
    except getattr(torch, "OutOfMemoryError", torch.cuda.OutOfMemoryError):      # (older torch 2.x: only torch.cuda.OutOfMemoryError)
        grad_pool_clear()
        if device.type == "cuda":
            torch.cuda.empty_cache()
        return torch.empty(shape, dtype=torch.float32, device=device)


class _GradPool:
    PER_KEY = 2                      # buffers kept per (device, stream, T, B): what one training step can have in flight

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.entries = []            # [key, keeper, version] -- most recently used last
        self.enabled = not os.environ.get("SEMICRF_NO_GRAD_POOL")
        if self.enabled and _USE_COUNT is None:
            self.enabled = False
            _warn_once("use_count", "transkun_amd: torch._C._storage_Use_Count is not available in this torch build; the pool of "
                                    "gradient / score buffers is off (every dense gradient is written in full, zeros included)")
        # budget: SEMICRF_GRAD_POOL_BYTES, else 4 GiB (two [1024,1024,352] buffers are 2.75 GiB); never more than PER_KEY per shape
        self.max_bytes = int(os.environ.get("SEMICRF_GRAD_POOL_BYTES", str(4 << 30)))
        self.min_bytes = 16 << 20    # smaller gradients: the zeros cost microseconds
        self.hits = self.misses = 0  # statistics (tests, bench)

    def held_bytes(self) -> int:
        """Bytes of device / host memory the pool keeps alive right now (outside the caching allocator's reach)."""
        with self.lock:
            return sum(e[1].numel() * 4 for e in self.entries)

    def take(self, T: int, B: int, device):
        """(dscore [T,T,B] fp32, flags): a pooled buffer with flags = GRAD_UPPER_IS_ZERO, or a fresh one with flags 0."""
        nbytes = 4 * T * T * B
        # inference tensors have no version counter (and nothing computes gradients under inference_mode): never pooled
        if not self.enabled or nbytes < self.min_bytes or torch.is_inference_mode_enabled():
            return _empty_or_trim((T, T, B), device), 0, None
        if device.type == "cpu":
            # host tensors too: the host kernels write the zeros anyway, but a fresh 1.4 GB allocation is 350 000 first-touch page
            # faults that many threads take at once (T=1024 x 352: the backward took 1.4 s on 8 threads and 8 - 20 s on 22 - 64)
            key = ("cpu", 0, T, B)
        else:
            if device.type != "cuda" or torch.cuda.is_current_stream_capturing():
                return _empty_or_trim((T, T, B), device), 0, None
            key = (device.index if device.index is not None else torch.cuda.current_device(),
                   torch.cuda.current_stream(device).cuda_stream, T, B)
        with self.lock:
            for i in range(len(self.entries) - 1, -1, -1):
                k, keeper, ver = self.entries[i]
                if k == key and _storage_users(keeper) == 1:
                    del self.entries[i]
                    if keeper._version == ver:
                        self.hits += 1
                        return keeper, GRAD_UPPER_IS_ZERO, key           # keeper becomes the handed-out tensor, see give()
                    self.misses += 1
                    return keeper, 0, key                                # written in place since: everything is written again
            self.misses += 1
        return _empty_or_trim((T, T, B), device), 0, key

    def give(self, key, dscore: torch.Tensor) -> None:
        """After the library's last write into `dscore` (its upper triangle holds exact zeros now): remember the memory."""
        if key is None or dscore.is_inference():
            return
        keeper = dscore.detach()
        nbytes = keeper.numel() * 4
        with self.lock:
            self.entries.append([key, keeper, keeper._version])
            same = [i for i, e in enumerate(self.entries) if e[0] == key]
            for i in same[:-self.PER_KEY]:                               # the oldest of this shape beyond PER_KEY
                self.entries[i] = None
            self.entries = [e for e in self.entries if e is not None]
            total = 0
            for i in range(len(self.entries) - 1, -1, -1):               # newest first; drop what exceeds the budget
                total += self.entries[i][1].numel() * 4
                if total > max(self.max_bytes, nbytes):
                    del self.entries[:i + 1]
                    break

    def rerecord(self, dscore: torch.Tensor) -> None:
        """The library itself has written cells with begin <= end into a pooled buffer again (the path cells of evalPath's
        gradient): that write is not an edit of the upper triangle."""
        if dscore is None:
            return
        ptr = dscore.data_ptr()
        with self.lock:
            for e in self.entries:
                if e[1].data_ptr() == ptr:
                    e[2] = e[1]._version

    def clear(self) -> None:
        with self.lock:
            self.entries.clear()


_GRAD_POOL = _GradPool()


def grad_pool_bytes() -> int:
    """Bytes the gradient pool and the scorer's score pool hold right now."""
    from .. import scorer as _sc
    return _GRAD_POOL.held_bytes() + (_sc._SCORE_POOL.held_bytes() if _sc._SCORE_POOL is not None else 0)


def grad_pool_clear() -> None:
    """Release the gradient buffers the pool holds (at most two per (device, stream, T, B) and SEMICRF_GRAD_POOL_BYTES -- 4 GiB by
    default -- in total; SEMICRF_NO_GRAD_POOL=1 disables the pool; an out-of-memory error inside the library releases them by
    itself) -- and the interval scorer's pool of score tensors, which works the same way (transkun_amd/scorer.py)."""
    _GRAD_POOL.clear()
    from .. import scorer as _sc
    if _sc._SCORE_POOL is not None:
        _sc._SCORE_POOL.clear()


def _logz_bwd_raw(score, noise, v, logz, gout, want_q: bool = False):
    if _odd_pad(score):
        ds, dn, q = _logz_bwd_raw(_pad1(score), _pad1(noise), _pad1(v), _pad1(logz), _pad1(gout), want_q)
        return ds[:, :, :-1].contiguous(), dn[:, :-1].contiguous(), (q[:, :-1].contiguous() if q is not None else None)
    T, B = score.shape[0], score.shape[2]
    dscore, flags, pkey = _GRAD_POOL.take(T, B, score.device)
    dnoise = torch.empty_like(noise)
    q = torch.empty((T, B) if want_q else (0,), dtype=torch.float32, device=score.device)
    ws = _lib.leased_workspace(_lib.OP_LOGZ_BWD, T, B, score.device)
    _lib.ops().logz_bwd(score, noise, v, logz, gout, dscore, dnoise, q, want_q, flags, ws)
    _GRAD_POOL.give(pkey, dscore)
    return dscore, dnoise, (q if want_q else None)


def _logprob_fwd_raw(score, noise, pairs, offsets, want_v: bool):
    """(logProb [B], logZ [B], alpha [T,B] or None) from ONE launch: spare waves of the forward sweep compute the path scores, the
    ring wave that finalises logZ subtracts (semicrf_logprob_fwd; bit-identical to _eval_path_raw(...) - logZ).  `pairs` must be
    ready on the current stream (_ready)."""
    K = getattr(pairs, "_semicrf_K", pairs.shape[0])
    if _odd_pad(score):
        logz, v = _logz_fwd_raw(score, noise, want_v)
        return _eval_path_raw(score, noise, pairs, offsets) - logz, logz, v
    T, B = score.shape[0], score.shape[2]
    dev = score.device
    lp = torch.empty(B, dtype=torch.float32, device=dev)
    logz = torch.empty(B, dtype=torch.float32, device=dev)
    v = torch.empty((T, B) if want_v else (0,), dtype=torch.float32, device=dev)
    ws = _lib.leased_workspace(_lib.OP_LOGZ_FWD, T, B, dev)
    _lib.ops().logprob_fwd(score, noise, pairs, int(K), offsets, lp, logz, v, want_v, ws)
    return lp, logz, (v if want_v else None)


def _eval_path_raw(score, noise, pairs, offsets):
    T, B = score.shape[0], score.shape[2]
    out = torch.empty(B, dtype=torch.float32, device=score.device)
    ws = _lib.workspace(_lib.OP_EVAL_PATH, T, B, score.device)
    K = getattr(pairs, "_semicrf_K", pairs.shape[0])
    _lib.ops().eval_path(score, noise, pairs, int(K), offsets, out, ws)
    return out


_EMPTY = {}


def _empty(device):
    e = _EMPTY.get(device)
    if e is None:
        e = _EMPTY[device] = torch.empty(0, dtype=torch.float32, device=device)
    return e


def _eval_path_bwd_raw(gout, T, B, pairs, offsets, dscore, dnoise, K: int, pooled: bool = False):
    """pooled: dscore is a gradient buffer this pass got from _logz_bwd_raw and nobody but this library has written to since
    (the scatter touches cells begin <= end only: the pool's record of the buffer stays valid)."""
    e = _empty(gout.device)
    _lib.ops().eval_path_bwd(gout, T, B, pairs, int(K), offsets, dscore if dscore is not None else e, dscore is not None,
                             dnoise if dnoise is not None else e, dnoise is not None)
    if pooled:
        _GRAD_POOL.rerecord(dscore)


def _gout(grad_output: torch.Tensor, B: int) -> torch.Tensor:
    assert grad_output.shape[-1] == B      # reference :471
    g = grad_output.reshape(B)
    if g.dtype != torch.float32:
        g = g.float()
    return g.contiguous()


# --------------------------------------------------------------------------------------
# autograd nodes
# --------------------------------------------------------------------------------------

# The reference's own call pattern is crf.evalPath(...) - crf.computeLogZ() as TWO autograd nodes (ModelTransformer.py:263-265).
# Differentiated naively, the evalPath node returns a dense zero [T,T,B] tensor with a few thousand cells set and autograd
# adds it to the dense gradient of the logZ node: two more passes over 1.48 GB at T=1024, NBatch=352.
#
# The methods of one NeuralSemiCRFInterval object therefore hang their nodes on a private HUB node (_Hub: an identity on
# (score, noiseScore) that only this object knows).  In backward the children do not hand dense gradients to the engine: they
# DEPOSIT them in the hub's accumulator of the running graph task -- the first dense gradient becomes the buffer, later dense
# ones are added in place, an evalPath node scatters its +-gout cells straight into it (or is parked until a dense gradient
# arrives) -- and return None; the hub's own backward, which the engine runs after every child of the pass, hands the
# buffer on.  The accumulator OWNS the tensor (a strong reference, no raw addresses), so any number of consumers of `score`
# in any order is safe: consumers outside this object meet the hub's result one level up, in the engine's own buffers.
# The module-level functions (no object, no hub) differentiate the plain way.


def _graph_task_id() -> int:
    fn = getattr(torch._C, "_current_graph_task_id", None)
    return int(fn()) if fn is not None else -1


class _HubState:
    """Gradient accumulators of one CRF object's hub, one per graph task in flight."""

    def __init__(self):
        self.acc = {}

    def _slot(self, tid):
        a = self.acc.get(tid)
        if a is None:
            a = self.acc[tid] = {"ds": None, "dn": None, "paths": [], "shape": None, "dev": None}
            if len(self.acc) > 8:                       # passes that never reached the hub (an error mid-backward)
                for k in list(self.acc)[:-8]:
                    if self.acc[k]["ds"] is not None or self.acc[k]["paths"]:
                        import warnings
                        warnings.warn("transkun_amd.CRF: more than 8 backward passes through one NeuralSemiCRFInterval object are in "
                                      "flight; the oldest one's deposited gradient is dropped (re-entrant / multi-threaded backward "
                                      "sharing one CRF object?) -- its score gradient will be incomplete")
                    del self.acc[k]
        return a

    @staticmethod
    def _scatter(a, path):
        g, pairs, offsets, K, T, B = path
        _eval_path_bwd_raw(g, T, B, pairs, offsets, a["ds"], a["dn"], K, pooled=a.get("pristine", False))

    def deposit_dense(self, tid, dscore, dnoise):
        a = self._slot(tid)
        if a["ds"] is None:
            a["ds"], a["dn"] = dscore, dnoise
            a["pristine"] = True                    # straight from _logz_bwd_raw: only this library has written to it
            for p in a["paths"]:
                self._scatter(a, p)
            a["paths"] = []
        else:
            a["pristine"] = False
            a["ds"].add_(dscore)
            if dnoise is not None and a["dn"] is not None:
                a["dn"].add_(dnoise)

    def deposit_path(self, tid, g, pairs, offsets, K, T, B):
        a = self._slot(tid)
        a["shape"], a["dev"] = (T, B), g.device
        if a["ds"] is not None:
            self._scatter(a, (g, pairs, offsets, K, T, B))
        else:
            a["paths"].append((g, pairs, offsets, K, T, B))

    def collect(self, tid):
        a = self.acc.pop(tid, None)
        if a is None:
            return None, None
        if a["ds"] is None and a["paths"]:              # only evalPath nodes were reached: the dense zero tensor after all
            T, B = a["shape"]
            a["ds"] = torch.zeros(T, T, B, dtype=torch.float32, device=a["dev"])
            a["dn"] = torch.zeros(max(T - 1, 0), B, dtype=torch.float32, device=a["dev"])
            for p in a["paths"]:
                self._scatter(a, p)
        return a["ds"], a["dn"]


class _Hub(torch.autograd.Function):
    """Identity on (score, noiseScore), private to one NeuralSemiCRFInterval object: collects its children's gradients."""

    @staticmethod
    def forward(ctx, score, noiseScore, state):
        ctx.state = state
        ctx.set_materialize_grads(False)
        ctx.in_dtypes = (score.dtype, noiseScore.dtype)
        ctx.shapes = (score.shape, noiseScore.shape)
        return score.detach(), noiseScore.detach()

    @staticmethod
    def backward(ctx, gs, gn):
        ds, dn = ctx.state.collect(_graph_task_id())
        if gs is not None:                              # a child that returned its gradient the plain way
            ds = gs.float() if ds is None else ds.add_(gs)
        if gn is not None:
            dn = gn.float() if dn is None else dn.add_(gn)
        if ds is not None:
            ds = ds.reshape(ctx.shapes[0]).to(ctx.in_dtypes[0])
        if dn is not None:
            dn = dn.reshape(ctx.shapes[1]).to(ctx.in_dtypes[1])
        return ds, dn, None


def _hub_of(state):
    """The state to deposit in during this backward pass, or None (no hub / no graph task id: return gradients plainly)."""
    if state is None or torch.is_grad_enabled() or _graph_task_id() < 0:
        return None
    return state


class ComputeLogZFasterGrad(torch.autograd.Function):
    """logZ with a hand-written gradient (reference :459-475), recompute-in-backward flavour."""

    @staticmethod
    def forward(ctx, score, noiseScore, hub=None):
        score_c, noise_c = _prep(score), _prep(noiseScore)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        logz, v = _logz_fwd_raw(score_c, noise_c, want_v=need)
        if need:
            ctx.save_for_backward(score_c, noise_c, v, logz)
        ctx.in_dtypes = (score.dtype, noiseScore.dtype)
        ctx.hub = hub
        return logz.to(score.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        score, noise, v, logz = ctx.saved_tensors
        B = score.shape[2]
        dscore, dnoise, _ = _logz_bwd_raw(score, noise, v, logz, _gout(grad_output, B))
        hub = _hub_of(ctx.hub)
        if hub is not None:
            hub.deposit_dense(_graph_task_id(), dscore, dnoise)
            return None, None, None
        return dscore.to(ctx.in_dtypes[0]), dnoise.to(ctx.in_dtypes[1]), None


def computeLogZFasterGrad(score, noiseScore):
    return ComputeLogZFasterGrad.apply(score, noiseScore)


class _EvalPath(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, noiseScore, pairs, offsets, hub=None):
        score_c, noise_c = _prep(score), _prep(noiseScore)
        ctx.save_for_backward(pairs, offsets)
        ctx.shape = (score_c.shape[0], score_c.shape[2])
        ctx.K = getattr(pairs, "_semicrf_K", pairs.shape[0])
        ctx.in_dtypes = (score.dtype, noiseScore.dtype)
        ctx.hub = hub
        return _eval_path_raw(score_c, noise_c, pairs, offsets).to(score.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        pairs, offsets = ctx.saved_tensors
        T, B = ctx.shape
        g = _gout(grad_output, B)
        hub = _hub_of(ctx.hub)
        if hub is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # the path cells go straight into the pass's dense gradient (owned by the hub); nothing of its own
            hub.deposit_path(_graph_task_id(), g, pairs, offsets, ctx.K, T, B)
            return None, None, None, None, None
        dscore = torch.zeros(T, T, B, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        dnoise = torch.zeros(max(T - 1, 0), B, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        _eval_path_bwd_raw(g, T, B, pairs, offsets, dscore, dnoise, ctx.K)
        return (dscore.to(ctx.in_dtypes[0]) if dscore is not None else None,
                dnoise.to(ctx.in_dtypes[1]) if dnoise is not None else None, None, None, None)


def _gout_strided(grad_output: torch.Tensor, B: int):
    """(fp32 tensor, stride): the cotangent as the kernels take it -- B values (stride 1) or, when autograd hands down an
    expanded scalar (the usual loss, -logProb.sum() / n: train.py:187), that ONE value (stride 0): no [B] copy, no kernel."""
    assert grad_output.shape[-1] == B      # reference :471
    g = grad_output.reshape(B) if grad_output.dim() != 1 else grad_output
    if g.dtype != torch.float32:
        g = g.float()
    if B > 1 and g.stride(0) == 0:
        return g.as_strided((1,), (1,)), 0
    return g.contiguous(), 1


class _LogProb(torch.autograd.Function):
    """evalPath - logZ as one node (semicrf_logprob_fwd / semicrf_logprob_bwd): the forward subtracts inside the path kernel,
    the backward writes gout * (onehot(path) - marginals) in one dense pass."""

    @staticmethod
    def forward(ctx, score, noiseScore, pairs, offsets):
        score_c, noise_c = _prep(score), _prep(noiseScore)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        T, B = score_c.shape[0], score_c.shape[2]
        K = getattr(pairs, "_semicrf_K", pairs.shape[0])
        _ready(pairs)                                   # the intervals' copy ran on a side stream
        lp, logz, v = _logprob_fwd_raw(score_c, noise_c, pairs, offsets, need)
        if need:
            ctx.save_for_backward(score_c, noise_c, v, logz, pairs, offsets)
            ctx.K = K
        ctx.in_dtypes = (score.dtype, noiseScore.dtype)
        return lp.to(score.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        score, noise, v, logz, pairs, offsets = ctx.saved_tensors
        T, B = score.shape[0], score.shape[2]
        if _odd_pad(score):
            g = _gout(grad_output, B)
            dscore, dnoise, _ = _logz_bwd_raw(score, noise, v, logz, -g)
            _eval_path_bwd_raw(g, T, B, pairs, offsets, dscore, dnoise, ctx.K)          # (a padded copy: sliced below, never pooled)
        else:
            g, gstride = _gout_strided(grad_output, B)
            dscore, flags, pkey = _GRAD_POOL.take(T, B, score.device)
            dnoise = torch.empty_like(noise)
            ws = _lib.leased_workspace(_lib.OP_LOGZ_BWD, T, B, score.device)
            _lib.ops().logprob_bwd(score, noise, v, logz, g, gstride, pairs, int(ctx.K), offsets, dscore, dnoise, flags, ws)
            _GRAD_POOL.give(pkey, dscore)
        return dscore.to(ctx.in_dtypes[0]), dnoise.to(ctx.in_dtypes[1]), None, None


# --------------------------------------------------------------------------------------
# module-level functions with the reference's names
# --------------------------------------------------------------------------------------

def _viterbi_raw(score_c, noise_c, start, forward: bool):
    """Enqueue the Viterbi sweep + on-device backtrack; returns device tensors (pairs [cap,2], offsets [B+1])."""
    if _odd_pad(score_c):
        st = _pad1(start) if start is not None else None
        pairs, offsets = _viterbi_raw(_pad1(score_c), _pad1(noise_c), st, forward)
        return pairs, offsets[:-1]        # the ghost chain is last: its intervals lie behind offsets[B]
    T, B = score_c.shape[0], score_c.shape[2]
    dev = score_c.device
    cap = B * 2 * T
    pairs = torch.empty(cap, 2, dtype=torch.int32, device=dev)
    offsets = torch.empty(B + 1, dtype=torch.int32, device=dev)
    ws = _lib.leased_workspace(_lib.OP_VITERBI, T, B, dev)
    has = start is not None
    _lib.ops().viterbi(score_c, noise_c, start if has else offsets, has, bool(forward), pairs, offsets, ws)
    return pairs, offsets


def _decode(score, noiseScore, forcedStartPos: Optional[Sequence[int]], forward: bool, packed: bool = False):
    assert len(score.shape) == 3
    assert score.shape[0] == score.shape[1]
    T, B = _check_inputs(score, noiseScore)
    with torch.no_grad():
        score_c, noise_c = _prep(score.detach()), _prep(noiseScore.detach())
        dev = score_c.device
        start = None
        if forcedStartPos is not None:
            assert len(forcedStartPos) == B
            st = np.asarray(forcedStartPos, dtype=np.int64)
            if (st < 0).any() or (st > T - 1).any():
                raise IndexError(f"forcedStartPos out of range for T={T}")
            start = torch.from_numpy(st.astype(np.int32)).to(dev, non_blocking=True)
        pairs, offsets = _viterbi_raw(score_c, noise_c, start, forward)
        off_h = offsets.cpu()                      # the one host sync of decode
        total = int(off_h[-1])
        if total < 0:
            _lib.async_error()                     # consumed here: the next call must not report this time-out again
            raise RuntimeError("semicrf_viterbi: a bounded hand-off wait timed out on the device (GPU shared with work that "
                               "kept part of the persistent kernel from running?); the decode result is invalid")
        pairs_h = pairs[:total].cpu()
    if packed:
        return pairs_h.numpy(), off_h.numpy()
    return unpack_intervals(pairs_h, off_h, T)


def viterbiBackward(score, noiseScore, forcedStartPos: Optional[List[int]] = None) -> Intervals:
    """Right-to-left Viterbi, the default decode (reference :13-104)."""
    return _decode(score, noiseScore, forcedStartPos, forward=False)


def viterbi(score, noiseScore, forcedStartPos: Optional[List[int]] = None) -> Intervals:
    """Left-to-right Viterbi (reference :107-202); forcedStartPos is the END position here."""
    return _decode(score, noiseScore, forcedStartPos, forward=True)


def computeLogZ(score, noiseScore):
    """Reference :207-246 (the autograd-traceable variant).  Same kernel as computeLogZFasterGrad."""
    _check_inputs(score, noiseScore)
    return ComputeLogZFasterGrad.apply(score, noiseScore)


def forward_backward(score, noiseScore):
    """Reference :375-456: returns (logZ [B], grad [T,T,B], gradNoise [T-1,B])."""
    T, B = _check_inputs(score, noiseScore)
    with torch.no_grad():
        s, n = _prep(score.detach()), _prep(noiseScore.detach())
        logz, v = _logz_fwd_raw(s, n, want_v=True)
        ones = torch.ones(B, dtype=torch.float32, device=s.device)
        grad, grad_noise, _ = _logz_bwd_raw(s, n, v, logz, ones)
    return logz, grad, grad_noise


def evalPath(intervals: Intervals, score, noiseScore):
    """Unnormalised path score (reference :508-550)."""
    T, B = _check_inputs(score, noiseScore)
    pairs, offsets = pack_intervals(intervals, T, B, score.device)
    return _EvalPath.apply(score, noiseScore, pairs, offsets)


class NeuralSemiCRFInterval:
    def __init__(self, score, noiseScore):
        """The output layer for multiple tracks of non-overlapping intervals (reference :553-564).

        score      -- [T, T, nBatch]: score of every closed interval [begin, end], indexed
                      [end, begin, track]; only end >= begin is read.
        noiseScore -- [T-1, nBatch]: score of "no event" between frames t and t+1.
        """
        self.score = score
        self.noiseScore = noiseScore
        self._hub = None            # (score, noiseScore, score alias, noise alias, _HubState): strong references, compared by identity

    def _hubbed(self):
        """(score, noiseScore, hub state) for the differentiable methods: aliases behind this object's private hub node when a
        gradient can flow, the tensors themselves (hub None) otherwise."""
        s, n = self.score, self.noiseScore
        if not (torch.is_grad_enabled() and (s.requires_grad or n.requires_grad)) :
            return s, n, None
        h = self._hub
        if h is None or h[0] is not s or h[1] is not n:
            state = _HubState()
            hs, hn = _Hub.apply(s, n, state)
            h = self._hub = (s, n, hs, hn, state)
        return h[2], h[3], h[4]

    def decode(self, forcedStartPos=None, forward=False):
        if forward:
            return viterbi(self.score, self.noiseScore, forcedStartPos)
        else:
            return viterbiBackward(self.score, self.noiseScore, forcedStartPos)

    def decode_packed(self, forcedStartPos=None, forward=False):
        """An EXTENSION of the reference's surface: the decoded path of `decode` as two int32 arrays -- pairs [K, 2] of
        (begin, end), chain after chain and ascending within a chain, and offsets [nBatch + 1] (chain c owns
        pairs[offsets[c]:offsets[c + 1]]) -- i.e. what the device produced, before the Python lists are built.  `decode` spends
        ~25 ns of CPython object creation per interval on top of it (657 k intervals at T=2048, nBatch=352: 17 ms against 1 ms
        here); callers that go on with arrays anyway should take this one."""
        return _decode(self.score, self.noiseScore, forcedStartPos, bool(forward), packed=True)

    def evalPath(self, intervals):
        """compute the unnormalized score"""
        T, B = _check_inputs(self.score, self.noiseScore)
        pairs, offsets = pack_intervals(intervals, T, B, self.score.device)
        s, n, hub = self._hubbed()
        return _EvalPath.apply(s, n, pairs, offsets, hub)

    def computeLogZ(self, noBackward=False):
        """compute the log normalization factor"""
        _check_inputs(self.score, self.noiseScore)
        s, n, hub = self._hubbed()
        return ComputeLogZFasterGrad.apply(s, n, hub)

    def logProb(self, intervals, noBackward=False):
        T, B = _check_inputs(self.score, self.noiseScore)
        pairs, offsets = pack_intervals(intervals, T, B, self.score.device, overlap=not os.environ.get("SEMICRF_NO_COPY_OVERLAP"))
        return _LogProb.apply(self.score, self.noiseScore, pairs, offsets)
