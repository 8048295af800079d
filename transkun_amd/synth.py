"""Deterministic synthetic inputs for the semi-CRF path.

Every value is produced by an exact 64-bit integer hash of (seed, linear index) and mapped
to a dyadic rational that fp32 represents exactly, so the *same bits* come out of numpy on
a CPU box and of torch on an MI355X (no libm, no dependence on torch.randn streams).
SURVEY.md section 8(c)/(d): large golden cases commit only outputs; inputs are regenerated.

Distribution "randn": Irwin-Hall(4) of 16-bit uniforms, centred, / 2**15  -> mean 0,
std 1.1547, range (-4, 4), resolution 2**-15 (stand-in for crfMinimalExample.py:11-15).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

_M64 = (1 << 64) - 1
_C0 = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def _s64(x: int) -> int:
    """Two's-complement view of a uint64 constant as a Python int that fits int64."""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(z: torch.Tensor, r: int) -> torch.Tensor:
    return (z >> r) & ((1 << (64 - r)) - 1)


def hash_u64_torch(idx: torch.Tensor, seed: int) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (wrapping arithmetic)."""
    z = idx + _s64(seed * _C0)
    z = (z ^ _lsr(z, 30)) * _s64(_C1)
    z = (z ^ _lsr(z, 27)) * _s64(_C2)
    return z ^ _lsr(z, 31)


def hash_u64_numpy(idx: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((seed * _C0) & _M64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        return z ^ (z >> np.uint64(31))


def _ih4_torch(z: torch.Tensor) -> torch.Tensor:
    s = (z & 0xFFFF) + (_lsr(z, 16) & 0xFFFF) + (_lsr(z, 32) & 0xFFFF) + (_lsr(z, 48) & 0xFFFF)
    return (s - 131070).to(torch.float32) * (1.0 / 32768.0)


def _ih4_numpy(z: np.ndarray) -> np.ndarray:
    m = np.uint64(0xFFFF)
    s = (z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + ((z >> np.uint64(48)) & m)
    return (s.astype(np.int64) - 131070).astype(np.float32) * np.float32(1.0 / 32768.0)


def hash_normal(numel: int, seed: int, device="cpu", offset: int = 0, chunk: int = 1 << 26) -> torch.Tensor:
    """Flat fp32 tensor of `numel` randn-like values for linear indices offset..offset+numel."""
    out = torch.empty(numel, dtype=torch.float32, device=device)
    for s in range(0, numel, chunk):
        n = min(chunk, numel - s)
        idx = torch.arange(offset + s, offset + s + n, dtype=torch.int64, device=device)
        out[s:s + n] = _ih4_torch(hash_u64_torch(idx, seed))
    return out


def hash_normal_numpy(numel: int, seed: int, offset: int = 0) -> np.ndarray:
    idx = np.arange(offset, offset + numel, dtype=np.uint64)
    return _ih4_numpy(hash_u64_numpy(idx, seed))


def crf_inputs(T: int, B: int, seed: int = 1234, device="cpu", kind: str = "randn") -> Tuple[torch.Tensor, torch.Tensor]:
    """(score [T,T,B], noise [T-1,B]) fp32.

    kind "randn": both randn-like (crfMinimalExample.py:11-15).
    kind "model": scaled-inner-product-like scores: g * |e-b| * 0.25 off the diagonal, randn-like
    diagonal, noise == 0 (LayersTransformer.py:416-437: linear length scaling, zero noise).
    kind "ties": small-integer scores (exercises the first-maximum tie-break of Viterbi).
    """
    score = hash_normal(T * T * B, seed, device).view(T, T, B)
    noise = hash_normal(max(T - 1, 0) * B, seed + 1, device).view(max(T - 1, 0), B)
    if kind == "randn":
        return score, noise
    if kind == "model":
        t = torch.arange(T, device=device)
        length = (t[:, None] - t[None, :]).abs().to(torch.float32)
        diag = torch.diagonal(score, dim1=0, dim2=1).clone()          # [B,T]
        score = score * (0.25 * length)[:, :, None]
        torch.diagonal(score, dim1=0, dim2=1).copy_(diag)
        return score.contiguous(), torch.zeros_like(noise)
    if kind == "ties":
        return torch.round(score * 1.5), torch.round(noise * 1.5)
    raise ValueError(kind)


def crf_inputs_numpy(T: int, B: int, seed: int = 1234, kind: str = "randn"):
    s, n = crf_inputs(T, B, seed, "cpu", kind)
    return s.numpy(), n.numpy()


def synthetic_intervals(T: int, B: int, seed: int = 7, every: int = 16, active_every: int = 3) -> List[List[Tuple[int, int]]]:
    """Deterministic non-overlapping ascending interval lists (begin,end) per chain.

    Roughly music-like density: one chain in `active_every` is active; an active chain gets one
    event every `every` frames of length 1..8, some singletons and some touching pairs; other
    chains are empty (Data.py:1064-1096 guarantees the same preconditions for real targets).
    """
    out: List[List[Tuple[int, int]]] = []
    h = hash_u64_numpy(np.arange(B * (T // every + 1), dtype=np.uint64), seed).reshape(B, -1)
    for c in range(B):
        cur: List[Tuple[int, int]] = []
        if c % active_every == 0:
            t = int(h[c, 0] % np.uint64(every))
            k = 1
            while t < T:
                r = int(h[c, k % h.shape[1]]); k += 1
                ln = r % 9                      # 0 => singleton
                e = min(t + ln, T - 1)
                cur.append((t, e))
                if (r >> 8) % 4 == 0 and e > t and e < T - 1:   # touching follow-up interval
                    e2 = min(e + 1 + (r >> 12) % 4, T - 1)
                    cur.append((e, e2))
                    e = e2
                t = e + 1 + (r >> 16) % every
        out.append(cur)
    return out


class StandInBackbone(torch.nn.Module):
    """A stand-in for the reference's Backbone (LayersTransformer.py:444-660; out of scope here) with its
    interface: log-mel features [N, 1, T, n_mels, nWin] -> ctx [N, P, T, size].  One Linear over a frame's features plus a
    learned symbol embedding -- just enough arithmetic for BASELINE configs[3]'s data flow: mel front-end (fp32) -> backbone
    in bf16 -> fp32 hand-off into the interval scorer.  Benchmarks and tests only; nothing of the hot path depends on it."""

    def __init__(self, n_mels: int = 229, n_win: int = 6, n_sym: int = 90, size: int = 256):
        super().__init__()
        self.proj = torch.nn.Linear(n_mels * n_win, size)
        self.sym = torch.nn.Parameter(torch.zeros(n_sym, size))
        with torch.no_grad():
            self.sym.copy_(hash_normal(n_sym * size, 4242).view(n_sym, size) * 0.5)
            self.proj.weight.copy_(hash_normal(size * n_mels * n_win, 4243).view(size, n_mels * n_win) * (4.0 / (n_mels * n_win) ** 0.5))

    def forward(self, feat: torch.Tensor) -> torch.Tensor:
        N, _, T = feat.shape[:3]
        h = self.proj(feat.reshape(N, T, -1) - 0.5)                  # [N, T, size]
        return h.unsqueeze(1) + self.sym.view(1, -1, 1, h.shape[-1])   # [N, P, T, size]
