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
    l32 = loss.detach().float().reshape(())
    # (new_full is a fill on the device; torch.tensor(x, device=...) would be a copy from pageable memory: a host synchronisation)
    stats = torch.stack([l32, l32.new_full((), float(total_len)), l32.new_full((), float(n_batch))])
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


class FlatGradBucket:
    """The gradients of a model as views into ONE persistent flat fp32 buffer, exchanged in place.

    The reference sums gradients with one all_reduce PER PARAMETER (TrainUtil.py:36-48, called at train.py:229: 182 calls
    for the 13.61 M parameters of 2.0.conf, no divide).  Here
      * every trainable parameter's .grad is a VIEW of `flat` for the life of the bucket (autograd accumulates into it in
        place): no torch.cat into a fresh buffer and no copy back per step -- `zero()` replaces `p.grad = None`;
      * over RCCL the SUM is a reduce-scatter followed by an all-gather on the same buffer (rank r owns the r-th slice:
        both phases are in place): every rank talks to all 7 xGMI neighbours at once with 1/W of the bytes per link and
        phase, where a ring all-reduce is bound by one link (2 x 45 us against ~0.62 ms for 54.5 MB on 8 GPUs);
        backends without reduce-scatter (gloo, the CPU tests) take one all_reduce;
      * `arm()` hangs a post-accumulate hook on every parameter: when the last gradient of the bucket has been written, the
        exchange starts by itself on a side stream -- behind an event of the stream that produced the gradients -- and
        overlaps whatever the caller enqueues next (the loss bookkeeping, the next step's host work); `wait()` makes the
        current stream wait for it.  The persistent sweeps of this library need every one of their workgroups resident
        (bounded spins, then NaN): tests/test_gpu_parity.py::test_sweep_while_collective_in_flight runs both at once.
    """

    def __init__(self, parameters, group=None, always: bool = False):
        import torch.distributed as dist
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket: no trainable parameter")
        self.group = group
        self.always = always                 # issue the collectives even in a group of one (hardware tests of the RCCL path)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 or (dist.is_available() and dist.is_initialized()) else 0
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        unit = self.world * 64                                   # every rank's slice starts on a 256-byte boundary
        self.numel = n
        self.flat = torch.zeros((n + unit - 1) // unit * unit, dtype=torch.float32, device=dev)
        self._views = []
        off = 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatGradBucket: parameters must be fp32 on one device")
            v = self.flat[off:off + p.numel()].view_as(p)
            self._views.append(v)
            p.grad = v
            off += p.numel()
        self._hooks = []
        self._pending = 0
        self._armed = False                  # arm() has been called and no exchange has happened since
        self._exchanged = False              # an exchange has run since the last arm() / zero()
        self._closed = True                  # wait() has closed the step (gradients arriving now belong to the next one)
        self._late = 0                       # gradients that arrived between this step's exchange and its wait()
        self._event = None
        self._side = None
        self.rebound = 0                     # gradients found outside the flat buffer and moved back in (statistics / tests)
        self.collectives = 0                 # issued by the last exchange
        self.bytes_per_rank = 0

    # ---- the .grad <-> flat aliasing is checked, not assumed ------------------------------------------------------------------
    # optimizer.zero_grad(set_to_none=True) (the default) and the `p.grad = None` idiom rebind .grad: autograd then writes the
    # next gradient into a FRESH tensor and the flat buffer would be exchanged stale.  Every entry point below therefore
    # verifies that each p.grad is the view made in __init__ (same storage address) and repairs what is not.
    def _rebind(self, after_backward: bool) -> int:
        """after_backward=False (arm / zero): a gradient that is None becomes the (zeroed) view again, a foreign tensor is
        copied into its slice.  after_backward=True (before an exchange): a foreign tensor holds THIS step's gradient -- what
        the slice holds is stale -- and replaces the slice's content; None means 'no gradient this step': the slice is zeroed."""
        n = 0
        for p, v in zip(self.params, self._views):
            g = p.grad
            if g is not None and g.data_ptr() == v.data_ptr() and g.numel() == v.numel():
                continue
            with torch.no_grad():
                if g is None:
                    v.zero_()
                else:
                    if g.shape != v.shape:
                        raise RuntimeError("FlatGradBucket: a parameter's .grad changed its shape")
                    v.copy_(g)
            p.grad = v
            n += 1
        self.rebound += n
        return n

    def zero(self) -> None:
        self._rebind(after_backward=False)
        self.flat.zero_()
        self._exchanged = False
        self._closed = False
        self._late = 0

    def _backend(self) -> str:
        import torch.distributed as dist
        return dist.get_backend(self.group) if dist.is_available() and dist.is_initialized() else ""

    def _active(self) -> bool:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and (self.world > 1 or self.always)

    def exchange(self) -> int:
        """SUM over the ranks, in place, on the current stream.  Returns the number of collectives issued."""
        import torch.distributed as dist
        self._rebind(after_backward=True)
        self.collectives = 0
        self.bytes_per_rank = 0
        self._exchanged = True
        self._armed = False
        if not self._active():
            return 0
        if self._backend() == "nccl":
            shard = self.flat.view(self.world, -1)[self.rank]
            dist.reduce_scatter_tensor(shard, self.flat, op=dist.ReduceOp.SUM, group=self.group)      # in place: my slice of the sum
            dist.all_gather_into_tensor(self.flat, shard, group=self.group)                            # in place: everybody's slices
            self.collectives = 2
            self.bytes_per_rank = 2 * (self.world - 1) * shard.numel() * 4
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.collectives = 1
            self.bytes_per_rank = 2 * (self.world - 1) * (self.flat.numel() // max(self.world, 1)) * 4
        return self.collectives

    # ---- hook-started exchange on a side stream ---------------------------------------------------------------------------
    def arm(self) -> None:
        """The next backward pass starts the exchange itself, as soon as every parameter of the bucket has its gradient.
        ONE backward pass per arm(): with gradient accumulation over micro-batches, arm() before the LAST one only (the hooks of
        an unarmed bucket count nothing) -- a second backward after the armed one is reported by wait()."""
        self._rebind(after_backward=False)
        if not self._hooks:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._pending = len(self.params)
        self._armed = True
        self._exchanged = False
        self._closed = False
        self._late = 0
        self._event = None

    def disarm(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._armed = False

    def _on_grad(self, p) -> None:
        if self._armed:
            self._pending -= 1
            if self._pending == 0:
                self._start()
        elif self._exchanged and not self._closed:
            self._late += 1                                        # behind the exchange of a step that is still open
        # (not armed, step closed or not begun: plain accumulation into the flat buffer, nothing is counted)

    def _start(self) -> None:
        if not self._active():
            self.exchange()                                        # (rebinds and marks the step as exchanged; no collective)
            return
        if not self.flat.is_cuda:
            self.exchange()
            return
        cur = torch.cuda.current_stream(self.flat.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.flat.device)
        self._rebind(after_backward=True)                          # copies, if any, belong on the producing stream
        self._side.wait_stream(cur)                                # the gradients are complete on the producing stream
        with torch.cuda.stream(self._side):
            self.exchange()
            self._event = self._side.record_event()

    def wait(self) -> int:
        """Make the current stream wait for the hook-started exchange, or run the exchange now if none has happened since the
        last arm() / zero() (not every parameter received a gradient, the bucket was not armed, ...).  Returns the number of
        collectives of this step.  Raises when gradients arrived AFTER the exchange of this step had started (more backward
        passes than arm() calls): those gradients were not summed over the ranks."""
        late, self._late = self._late, 0
        if self._event is not None:
            torch.cuda.current_stream(self.flat.device).wait_event(self._event)
            self._event = None
        elif not self._exchanged:
            self.exchange()
        self._armed = False
        self._closed = True
        if late:
            raise RuntimeError(f"FlatGradBucket: {late} gradient(s) arrived after this step's exchange had started (a second "
                               "backward pass behind one arm()?): arm() before the LAST backward pass only, or call exchange() "
                               "yourself after all of them -- those gradients were not summed over the ranks")
        return self.collectives
