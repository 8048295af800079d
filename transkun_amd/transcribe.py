"""The transcription segment loop on top of the HIP path -- host-side mirror of
TransKun.transcribeFrames (/root/reference/transkun/ModelTransformer.py:537-725) and TransKun.transcribe (:729-848).

The reference transcribes a recording segment by segment (16 s windows, 8 s hop).  Per segment it decodes the semi-CRF
with a forced start position per symbol, walks the decoded Python lists on the host to build one Note per interval
(onset/offset refined by a small head), remembers per symbol the last confirmed offset (`lastP`), derives the next
segment's forced start from it (:789-791) and merges events that were cut by the segment boundary (:803-825).  One file
gives NBatch = 90 chains per decode and a host round trip per segment.

Here the per-segment work stays on the device: scorer -> Viterbi (packed intervals in HBM) -> optional onset-bound filter
-> attribute-head inputs gathered from the packed intervals -> the two heads (stock torch MLPs, as in the reference) ->
`segment_events` (event times in double with the reference's operation order, hasOnset/hasOffset, lastP and the NEXT forced
start as an int32 vector the next decode consumes directly).  Segments of DIFFERENT recordings run in lock step as one batch
(`transcribe_many`: F files -> NBatch = 90 F chains per launch), which is what keeps the decode kernels busy -- consecutive
segments of one file depend on each other.  Only the finished events of a step cross PCIe (asynchronously); the
incomplete-event merge is a short per-pitch walk on the host.

Out of scope: the audio front-end and the backbone.  What they produce -- `ctx` [segments, 90, T, D] --
is the input, through a callable that plays the role of makeFrame + processFramesBatch's backbone part.
"""
from __future__ import annotations

import importlib
import math
from collections import defaultdict
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, attributes
from . import fused
from .scorer import ScaledInnerProductIntervalScorer, _interval_score_raw, proj_forward, slot_maps, slot_pitch

_nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")


class Note:
    """Same fields as the reference's event object (Data.py:20-30)."""
    __slots__ = ("start", "end", "pitch", "velocity", "hasOnset", "hasOffset")

    def __init__(self, start, end, pitch, velocity, hasOnset=True, hasOffset=True):
        self.start = start
        self.end = end
        self.pitch = pitch
        self.velocity = velocity
        self.hasOnset = hasOnset
        self.hasOffset = hasOffset

    def __repr__(self):
        return str({k: getattr(self, k) for k in self.__slots__})

    def astuple(self):
        return (self.start, self.end, self.pitch, self.velocity, bool(self.hasOnset), bool(self.hasOffset))


def resolveOverlapping(note_events: List[Note]) -> List[Note]:
    """Data.py:170-214: in (start, end, pitch) order, an event that starts before the previous event of its pitch has ended
    cuts that event short; events left without duration are dropped."""
    note_events.sort(key=lambda x: (x.start, x.end, x.pitch))
    last_of_pitch = {}
    for i, ev in enumerate(note_events):
        j = last_of_pitch.get(ev.pitch)
        if j is not None and note_events[j].end > ev.start:
            note_events[j].end = ev.start
        last_of_pitch[ev.pitch] = i
    out = [n for n in note_events if n.start < n.end]
    out.sort(key=lambda x: (x.start, x.end, x.pitch))
    return out


class EventMerger:
    """The cross-segment bookkeeping of TransKun.transcribe for ONE recording (:745-746, :803-843): per pitch the list of
    events so far; a new event that starts before the last one of its pitch ended either replaces it (it has its own
    onset) or extends it (it is the continuation of an event cut by the previous segment's end)."""

    def __init__(self, mergeIncompleteEvent: bool = True):
        self.byType = defaultdict(list)
        self.merge = mergeIncompleteEvent

    def add_segment(self, events: Sequence[Note]) -> None:
        for e in events:                                             # :803-825
            lst = self.byType[e.pitch]
            if self.merge and lst:
                last_e = lst[-1]
                if e.start < last_e.end:
                    if e.hasOnset:
                        lst[-1] = e
                    else:
                        last_e.hasOffset = e.hasOffset
                        last_e.end = max(e.end, last_e.end)
                    continue
            if e.hasOnset:
                lst.append(e)

    def finish(self, resolve: bool = True) -> List[Note]:
        for lst in self.byType.values():                             # :831-834
            if lst:
                lst[-1].hasOffset = True
        events = [n for lst in self.byType.values() for n in lst if n.hasOffset]      # :837-841
        return resolveOverlapping(events) if resolve else events


class PackedEventMerger:
    """EventMerger + resolveOverlapping for ALL recordings of a transcribe_many call, on the packed rows of a step (start, end,
    hasOnset, hasOffset, velocity, symbol index, chain index as float64) instead of Note objects: the merge state lives in C
    (csrc/pymarshal.c: tm_*), one pass over the rows per step, and Note objects are made once, for the events that survive.  Same
    results as the Python classes above (tests/test_transcribe_merge.py); at the event density of a batched transcription the
    per-event Python walk, not the device, used to set the pace (8 ms of host work per 0.7 ms step)."""

    def __init__(self, n_files: int, pitches: Sequence[int], mergeIncompleteEvent: bool = True, eager_resolve: Optional[bool] = None):
        """eager_resolve (True / False = the `resolve` that finish() will be called with; needs the incomplete-event merge): Note
        objects are made inside add_step, as soon as nothing can change an event any more (it is not its track's last event and,
        with resolveOverlapping, neither is the event that may cut it short) -- i.e. while the device works on the next step --
        and finish() only merges the tracks' lists: 10 ms of object construction after the last step of a four-recording batch
        become < 1 ms.  None: all Notes at finish()."""
        self.pitches = [int(p) for p in pitches]
        self.n_files = n_files
        self._m = _lib.marshal()
        self._h = self._m.tm_new(n_files, len(self.pitches), bool(mergeIncompleteEvent))
        self.vel_float = False
        self.eager = eager_resolve is not None and bool(mergeIncompleteEvent)
        if self.eager:
            self._m.tm_eager(self._h, Note, self.pitches, bool(eager_resolve))

    def add_step(self, step_index: int, rows, K: int, active: Sequence[int], later_events_from: Optional[Sequence[float]] = None) -> None:
        """rows: a contiguous float64 [>= K, 7] host array (numpy or a CPU torch tensor), in chain order as the device wrote them.
        later_events_from (eager mode with resolve): per active recording, a time below which no event of a LATER step starts (the
        next segment's begin time): what lets events settle before finish(); without it they all wait."""
        if K <= 0:
            return
        addr = rows.data_ptr() if hasattr(rows, "data_ptr") else rows.ctypes.data
        if not self.eager:
            self._m.tm_add(self._h, int(step_index), int(addr), int(K), list(active))
            return
        import gc
        was = gc.isenabled()
        gc.disable()          # (fresh cycle-free objects by the thousand: see finish)
        try:
            self._m.tm_add(self._h, int(step_index), int(addr), int(K), list(active), bool(self.vel_float),
                           None if later_events_from is None else [float(b) for b in later_events_from])
        finally:
            if was:
                gc.enable()

    def finish(self, file: int, resolve: bool = True) -> List[Note]:
        import gc
        was = gc.isenabled()
        gc.disable()          # tens of thousands of fresh (cycle-free) objects would otherwise trigger several full collections
        try:
            return self._m.tm_finish(self._h, int(file), self.pitches, Note, bool(self.vel_float), bool(resolve))
        finally:
            if was:
                gc.enable()


def _head(n_in: int, hidden: int, n_out: int, dropout: float) -> nn.Sequential:
    return nn.Sequential(nn.Linear(n_in, hidden), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden, n_out))


class SegmentTranscriber(nn.Module):
    """The modules of TransKun that sit behind the backbone (ModelTransformer.py:97-124) with the reference's parameter
    names -- `scorer.map.0.*`, `velocityPredictor.{0,3}.*`, `refinedOFPredictor.{0,3}.*` load from its checkpoints -- and
    its segment loop."""

    def __init__(self, size: int = 256, velocityPredictorHiddenSize: int = 512, refinedOFPredictorHiddenSize: int = 512,
                 hopSize: int = 1024, windowSize: int = 4096, fs: int = 44100, segmentHopSizeInSecond: float = 8,
                 segmentSizeInSecond: float = 16, velocityDropoutProb: float = 0.1, refinedOFDropoutProb: float = 0.1,
                 targetMIDIPitch: Optional[Sequence[int]] = None):
        super().__init__()
        self.hopSize, self.windowSize, self.fs = hopSize, windowSize, fs
        self.segmentHopSizeInSecond, self.segmentSizeInSecond = segmentHopSizeInSecond, segmentSizeInSecond
        self.targetMIDIPitch = list(targetMIDIPitch) if targetMIDIPitch is not None else [-64, -67] + list(range(21, 108 + 1))   # :97
        self.scorer = ScaledInnerProductIntervalScorer(size, 1)
        self.scorer.fullSquare = 2      # S goes straight into this package's decode, which never reads begin > end: no zero fill
        self.capFactor, self.capFloor = 1.5, 4096     # transcribe_many: rows for the heads = capFactor x the largest count seen, at least capFloor
        self.projection = "separate"    # "merged": the scorer's two projections as one (see decode_step); scores then differ from
                                        # the reference's by fp32 reassociation, so the default keeps its operation order
        self._merged = None
        self.velocityPredictor = _head(size * 3, velocityPredictorHiddenSize, 128, velocityDropoutProb)           # :109-115
        self.refinedOFPredictor = _head(size * 3, refinedOFPredictorHiddenSize, 4, refinedOFDropoutProb)          # :119-125

    def _merged_weights(self):
        lin = self.scorer.map[0]
        key = (lin.weight._version, lin.bias._version, lin.weight.data_ptr(), lin.bias.data_ptr())
        if self._merged is None or self._merged[0] != key:
            with torch.no_grad():
                self._merged = (key, fused.merged_weights(lin.weight.float(), lin.bias.float(), self.scorer.size * self.scorer.expansionFactor))
        return self._merged[1]

    # ------------------------------------------------------------------------------------------------------------------
    # one step on the device
    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode_step(self, ctxBatch: torch.Tensor, start: Optional[torch.Tensor], beginTime: torch.Tensor, lastFrameIdx: int,
                    stepFrames: int, onsetBound: Optional[int] = None, velocityCriteron: str = "hamming", k_cap: Optional[int] = None):
        """ctxBatch [F, P, T, D] (one segment of each of F recordings); start: int32 [F*P] forced start positions on the device
        or None; beginTime: float64 [F] on the device.  Returns a dict of DEVICE tensors: pairs [K,2], offsets [F*P+1],
        symIdx [K], scatterIdx [K], velocity [K], times [K,2] f64, flags [K,2] u8, lastP [F*P], nextStart [F*P], and K.

        k_cap: run WITHOUT the host synchronisation for K: everything behind the decode is sized for k_cap intervals (the rows
        behind the real count hold harmless values), `K` in the result is None and `Kdev` is the count on the device; the caller
        checks it later (it must not exceed k_cap: beyond it the chains' events -- and `nextStart` -- are cut off)."""
        assert ctxBatch.dim() == 4
        Fn, P, T, D = ctxBatch.shape
        assert P == len(self.targetMIDIPitch)
        B = Fn * P
        dev = ctxBatch.device
        ops = _lib.ops()
        # processFramesBatch :199-222.  S stays inside this step, so its chain axis uses the slot layout (include/semicrf_hip.h):
        # the P symbols of a segment in `pitch` slots (96 for 90) -- whole 128-byte lines for the CRF kernels -- with all-zero
        # ghost chains that decode to nothing; the packed result is chain-indexed again before anything else sees it
        pitch = slot_pitch(P, T, D, Fn)
        if self.projection not in ("separate", "merged"):
            raise ValueError(f"projection must be 'separate' or 'merged', not {self.projection!r}")
        if self.projection == "merged" and fused.merged_eligible(D, T) and self.scorer.expansionFactor == 1:
            # opt-in: ONE size -> size GEMM instead of the reference's size -> 2 size + 1 (fused.merged_weights): the same scores up
            # to fp32 reassociation -- a decoded path can differ from the reference's only where two paths tie to ~1e-6
            Wm, bm = self._merged_weights()
            x3 = ctxBatch.float().contiguous().view(B, T, D)
            zc = proj_forward(x3.view(-1, D), Wm, bm, D, Wt=getattr(Wm, "_semicrf_T", None)).view(B, T, -1)
            score, noise = _interval_score_raw(zc[..., :D], x3, zc[..., D + 1], T, B, D, 1.0 / math.sqrt(D),
                                               _lib.LEN_MODES[self.scorer.lengthScaling], 2, P, pitch, rowc=zc[..., D])
        else:
            self.scorer.slotPitch = pitch if pitch != P else None
            try:
                S, b = self.scorer(ctxBatch)
            finally:
                self.scorer.slotPitch = None
            score, noise = S.flatten(-2, -1), b.flatten(-2, -1)
        if pitch != P:
            real, _ = slot_maps(Fn, P, pitch, dev)
            start_s = None
            if start is not None:
                start_s = torch.zeros(Fn * pitch, dtype=torch.int32, device=dev).index_copy_(0, real, start)
            pairs, offsets_s = _nsci._viterbi_raw(score, noise, start_s, False)          # transcribeFrames :549
            # offsets by chain: chain c starts where its slot starts; the total (and the time-out marker) is the last entry
            offsets = torch.cat([offsets_s.index_select(0, real), offsets_s[-1:]])
        else:
            pairs, offsets = _nsci._viterbi_raw(score, noise, start, False)              # transcribeFrames :549
        if onsetBound is not None:                                                       # :554-555
            pairs2 = torch.empty_like(pairs)
            offsets2 = torch.empty_like(offsets)
            counts = torch.empty(B, dtype=torch.int32, device=dev)
            ops.segment_onset_filter(pairs, offsets, B, int(onsetBound), pairs2, offsets2, counts)
            pairs, offsets = pairs2, offsets2
        Kdev = offsets[-1:]
        if k_cap is not None:
            # no host trip: k_cap rows, the chains' ranges cut at k_cap, the (begin, end) of the unused rows made valid frame indices
            K = int(k_cap)
            # (a decode whose hand-off wait timed out leaves offsets[-1] = -1: every chain is then EMPTY for the kernels below --
            # they must not index with ranges the decode never validated; the count, checked one step late, raises)
            offsets = torch.clamp(offsets, min=0, max=K) * (offsets[-1:] >= 0).to(offsets.dtype)
            pairs = pairs[:K].clamp(0, T - 1)
        else:
            K = int(offsets[-1])                                                         # the step's one host sync
            if K < 0:
                _lib.async_error()               # consumed here: the next library call must not report this time-out again
                raise RuntimeError("semicrf_viterbi: a bounded hand-off wait timed out on the device; the decode result is invalid")
        lastP = torch.empty(B, dtype=torch.int32, device=dev)
        nextStart = torch.empty(B, dtype=torch.int32, device=dev)
        if K == 0:                                                                       # :570-572: nothing detected
            lastP.zero_(); nextStart.zero_()
            e = torch.empty(0, device=dev)
            return dict(K=0, pairs=pairs[:0], offsets=offsets, symIdx=e.long(), scatterIdx=e.long(), velocity=e.long(),
                        times=torch.empty(0, 2, dtype=torch.float64, device=dev), flags=torch.empty(0, 2, dtype=torch.uint8, device=dev),
                        lastP=lastP, nextStart=nextStart)
        attributeInput, sym, sc = attributes.attribute_input_packed(ctxBatch, pairs, offsets, K)       # :578-586
        logitsVelocity = self.velocityPredictor(attributeInput)
        velocity = self._velocity(logitsVelocity, velocityCriteron)
        ofValue, ofPresence = self.refinedOFPredictor(attributeInput).chunk(2, dim=-1)               # :646-655
        ofDist = torch.distributions.ContinuousBernoulli(logits=ofValue, validate_args=False)   # (the argument check is a host sync)
        ofValue = torch.clamp((ofDist.mean - 0.5) / 0.99, -0.5, 0.5).float().contiguous()
        ofPresence = (ofPresence > 0).contiguous()
        times = torch.empty(K, 2, dtype=torch.float64, device=dev)
        flags = torch.empty(K, 2, dtype=torch.uint8, device=dev)
        ops.segment_events(pairs, K, offsets, B, P, ofValue, ofPresence.view(torch.uint8), int(lastFrameIdx), self.hopSize / self.fs,
                           beginTime, int(stepFrames), times, flags, lastP, nextStart)
        return dict(K=K if k_cap is None else None, Kdev=Kdev, k_cap=k_cap, pairs=pairs[:K], offsets=offsets, symIdx=sym, scatterIdx=sc,
                    velocity=velocity, times=times, flags=flags, lastP=lastP, nextStart=nextStart, ofValue=ofValue, ofPresence=ofPresence)

    @staticmethod
    def _velocity(logitsVelocity: torch.Tensor, criterion: str) -> torch.Tensor:
        pVelocity = F.softmax(logitsVelocity, dim=-1)                                    # :590-637
        dev = pVelocity.device
        if criterion == "hamming":
            return torch.argmax(pVelocity, dim=-1)
        if criterion == "mse":
            return (pVelocity * torch.arange(128, device=dev)).sum(-1)
        if criterion == "match":
            w = torch.arange(128, device=dev)
            utility = ((w.unsqueeze(1) - w.unsqueeze(0)).abs() < 0.1 * 128).float()
            return torch.argmax(pVelocity @ utility, dim=-1)
        if criterion == "mae":
            tmp = (pVelocity.cumsum(-1) - 0.5) > 0
            return torch.argmax(tmp * torch.arange(128, 0., -1, device=dev), dim=-1)
        raise Exception("Unrecognized criterion: {}".format(criterion))

    @staticmethod
    def _packed_rows(step: dict) -> torch.Tensor:
        """The step's events as ONE float64 [K, 7] device tensor (every field is exactly representable): start, end, hasOnset,
        hasOffset, velocity, symbol index, chain index."""
        return torch.stack([step["times"][:, 0], step["times"][:, 1], step["flags"][:, 0].to(torch.float64),
                            step["flags"][:, 1].to(torch.float64), step["velocity"].to(torch.float64),
                            step["symIdx"].to(torch.float64), step["scatterIdx"].to(torch.float64)], dim=1)

    def _notes_of_step(self, step: dict, n_files: int) -> List[List[Note]]:
        """Host objects of a step: per recording the Notes in the reference's order (sorted by (start, end, pitch), :722)."""
        K = step["K"]
        out: List[List[Note]] = [[] for _ in range(n_files)]
        if K == 0:
            return out
        P = len(self.targetMIDIPitch)
        # ONE copy to the host
        packed = self._packed_rows(step).cpu().numpy()
        seg = packed[:, 6].astype(np.int64) // P
        pitch = np.asarray(self.targetMIDIPitch, dtype=np.int64)[packed[:, 5].astype(np.int64)]
        order = np.lexsort((pitch, packed[:, 1], packed[:, 0], seg))             # by recording, then (start, end, pitch)
        # velocity: class indices for 'hamming' / 'match' / 'mae', the float expectation for 'mse' (ModelTransformer.py:597: its
        # .tolist() yields floats there)
        vel = packed[order, 4] if step["velocity"].is_floating_point() else packed[order, 4].astype(np.int64)
        cols = (seg[order].tolist(), packed[order, 0].tolist(), packed[order, 1].tolist(), pitch[order].tolist(),
                vel.tolist(), (packed[order, 2] != 0).tolist(), (packed[order, 3] != 0).tolist())
        for sg, a, b, p, v, f0, f1 in zip(*cols):
            out[sg].append(Note(a, b, p, v, f0, f1))
        return out

    # ------------------------------------------------------------------------------------------------------------------
    # the reference's two entry points
    # ------------------------------------------------------------------------------------------------------------------
    def transcribeFrames(self, ctxBatch, forcedStartPos=None, velocityCriteron="hamming", onsetBound=None, lastFrameIdx=None):
        """ModelTransformer.py:537-725 with the backbone's output in the place of the frames: returns (notes per segment of
        the batch, lastP per chain) as host objects.  Times are relative to the segment, as in the reference."""
        Fn, P, T, D = ctxBatch.shape
        dev = ctxBatch.device
        if lastFrameIdx is None:
            lastFrameIdx = T - 1
        start = None
        if forcedStartPos is not None:
            assert len(forcedStartPos) == Fn * P
            start = torch.tensor(list(forcedStartPos), dtype=torch.int32, device=dev)
        step = self.decode_step(ctxBatch, start, torch.zeros(Fn, dtype=torch.float64, device=dev), lastFrameIdx, 0, onsetBound,
                                velocityCriteron)
        if step["K"] == 0:
            return [[] for _ in range(Fn)], [0 for _ in range(Fn * P)]                  # :570-572
        return self._notes_of_step(step, Fn), step["lastP"].cpu().tolist()

    def transcribe(self, ctx_of_segment: Callable[[int, int], torch.Tensor], nSample: int, stepInSecond=None, segmentSizeInSecond=None,
                   discardSecondHalf=False, mergeIncompleteEvent=True, resolve=True) -> List[Note]:
        """ModelTransformer.py:729-848 for one recording of nSample samples (before padding); ctx_of_segment(i, T) returns the
        backbone output [1, 90, T, D] of segment i.  See transcribe_many for the batched form."""
        return self.transcribe_many([ctx_of_segment], [nSample], stepInSecond, segmentSizeInSecond, discardSecondHalf,
                                    mergeIncompleteEvent, resolve)[0]

    def segment_plan(self, nSample: int, stepInSecond=None, segmentSizeInSecond=None):
        """The segment geometry of TransKun.transcribe (:731-757, :778): padding, step and segment sizes, frames per segment."""
        if stepInSecond is None and segmentSizeInSecond is None:
            stepInSecond = self.segmentHopSizeInSecond
            segmentSizeInSecond = self.segmentSizeInSecond
        padTimeBegin = segmentSizeInSecond - stepInSecond
        pad = math.ceil(padTimeBegin * self.fs)
        total = nSample + 2 * pad
        startFrameIdx = math.floor(padTimeBegin * self.fs / self.hopSize)
        stepSize = math.ceil(stepInSecond * self.fs / self.hopSize) * self.hopSize
        segmentSize = math.ceil(segmentSizeInSecond * self.fs)
        nFrame = math.ceil(segmentSize / self.hopSize) + 1                              # makeFrame, Util.py:24
        return dict(padTimeBegin=padTimeBegin, nTotal=total, startFrameIdx=startFrameIdx, stepSize=stepSize, segmentSize=segmentSize,
                    nFrame=nFrame, lastFrameIdx=round(segmentSize / self.hopSize), begins=list(range(0, total, stepSize)))

    @torch.no_grad()
    def transcribe_many(self, ctx_fns: Sequence[Callable[[int, int], torch.Tensor]], nSamples: Sequence[int], stepInSecond=None,
                        segmentSizeInSecond=None, discardSecondHalf=False, mergeIncompleteEvent=True, resolve=True,
                        synchronous: bool = False) -> List[List[Note]]:
        """Several recordings in lock step: step s decodes segment s of every recording that still has one as ONE batch
        (NBatch = 90 x #recordings).  The forced start positions of step s+1 never leave the device.

        After the first step nothing waits for the device inside a step: the attribute heads run on a CAPPED number of rows (1.5 x
        the largest interval count seen so far per recording in the batch), the real count travels to the host with the step's rows
        and is checked one step late, when the rows are merged.  A count above the cap (the heads saw a truncated list) restarts
        the whole call with `synchronous=True`: every step then waits for its count, as in round 3.  Same Notes either way.
        `ctx_fns[f](s, T)` must therefore be RE-CALLABLE (a pure function of the step): a restart asks for step 0 again."""
        plans = [self.segment_plan(n, stepInSecond, segmentSizeInSecond) for n in nSamples]
        P = len(self.targetMIDIPitch)
        dev = next(self.parameters()).device
        merger = PackedEventMerger(len(plans), self.targetMIDIPitch, mergeIncompleteEvent, eager_resolve=bool(resolve))
        nsteps = max(len(p["begins"]) for p in plans)
        T = plans[0]["nFrame"]
        stepFrames = int(plans[0]["stepSize"] / self.hopSize)                           # :791
        onsetBound = plans[0]["stepSize"] if discardSecondHalf else None                # :779-782 (the reference passes samples)
        start = torch.full((len(plans) * P,), plans[0]["startFrameIdx"], dtype=torch.int32, device=dev)      # :751-752
        active_prev = list(range(len(plans)))
        pending = None                  # (step index, K, pinned rows, copy event, active files): the host part runs one step late
        bufs = [None, None]             # two pinned row buffers, alternating: step s fills one while step s-1's is merged

        use_cap = dev.type == "cuda" and not synchronous
        kpin = torch.empty(2, dtype=torch.int32, pin_memory=True) if use_cap else None
        kmax_per_file = [0.0]           # largest verified interval count per recording of a batch

        class _Overflow(Exception):
            pass

        def to_host(s, step):
            K = step["K"]
            cap = step.get("k_cap")
            n = K if cap is None else cap
            if n == 0:
                return None
            merger.vel_float = step["velocity"].is_floating_point()
            rows = self._packed_rows(step)
            if dev.type != "cuda":
                return (s, K, rows.contiguous(), None, None)
            b = bufs[s & 1]
            if b is None or b.shape[0] < n:
                b = bufs[s & 1] = torch.empty(max(2 * n, 4096), 7, dtype=torch.float64, pin_memory=True)
            b[:n].copy_(rows, non_blocking=True)
            if cap is not None:
                kpin[(s & 1):(s & 1) + 1].copy_(step["Kdev"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return (s, K, b, ev, cap)

        def merge(p):
            if p is None or p[0] is None:
                return
            (s, K, rows, ev, cap), active = p
            if ev is not None:
                ev.synchronize()
            if cap is not None:
                K = int(kpin[s & 1])
                if K < 0:
                    _lib.async_error()           # consumed here: the next library call must not report this time-out again
                    raise RuntimeError("semicrf_viterbi: a bounded hand-off wait timed out on the device; the decode result is invalid")
                if K > cap:
                    raise _Overflow()
            kmax_per_file[0] = max(kmax_per_file[0], K / max(len(active), 1))
            if K > 0:
                # no event of step s + 1 starts before that step's segment begins (minus the refinement's half frame: one frame here)
                frame = self.hopSize / self.fs
                later = [plans[f]["begins"][s + 1] / self.fs - plans[f]["padTimeBegin"] - frame if s + 1 < len(plans[f]["begins"]) else 1e300
                         for f in active]
                merger.add_step(s, rows, K, active, later_events_from=later)

        # every step's segment begin times in ONE upload, packed by step (a per-step torch.tensor(..., device=...) is a copy from
        # pageable memory: the host waits for the device each time -- tools/sync_audit.py)
        bt_rows, bt_off = [], []
        for s in range(nsteps):
            bt_off.append(len(bt_rows))
            bt_rows.extend(plans[f]["begins"][s] / self.fs - plans[f]["padTimeBegin"] for f in range(len(plans)) if s < len(plans[f]["begins"]))
        bt_host = torch.tensor(bt_rows, dtype=torch.float64)
        bt_all = bt_host.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else bt_host

        for s in range(nsteps):
            active = [f for f, p in enumerate(plans) if s < len(p["begins"])]
            if active != active_prev:                                                   # recordings that ended drop out of the batch
                keep = torch.tensor([active_prev.index(f) for f in active], device=dev)
                start = start.view(len(active_prev), P)[keep].reshape(-1).contiguous()
                active_prev = active
            ctxBatch = torch.cat([ctx_fns[f](s, T) for f in active], dim=0) if len(active) > 1 else ctx_fns[active[0]](s, T)
            beginTime = bt_all[bt_off[s]:bt_off[s] + len(active)]                        # :766
            cap = None
            if use_cap and s > 0:
                cap = max(int(self.capFloor), (int(self.capFactor * kmax_per_file[0] * len(active)) + 1023) // 1024 * 1024)
            step = self.decode_step(ctxBatch, start, beginTime, plans[0]["lastFrameIdx"], stepFrames, onsetBound, k_cap=cap)
            start = step["nextStart"]                                                   # :789-791, stays on the device
            if step["K"] is not None:                                                   # (a synchronous step knows its count at once)
                kmax_per_file[0] = max(kmax_per_file[0], step["K"] / max(len(active), 1))
            host = to_host(s, step)                                                     # the rows' copy runs behind the step's kernels
            try:
                merge(pending)                                                          # ... while the host merges the previous step
            except _Overflow:
                return self.transcribe_many(ctx_fns, nSamples, stepInSecond, segmentSizeInSecond, discardSecondHalf, mergeIncompleteEvent,
                                            resolve, synchronous=True)
            pending = (host, active)
        try:
            merge(pending)
        except _Overflow:
            return self.transcribe_many(ctx_fns, nSamples, stepInSecond, segmentSizeInSecond, discardSecondHalf, mergeIncompleteEvent,
                                        resolve, synchronous=True)
        return [merger.finish(f, resolve) for f in range(len(plans))]
