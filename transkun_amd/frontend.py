"""Audio front-end of the reference model as stock torch ops: framing, the six-window spectrum and the
log-mel features that feed the backbone (/root/reference/transkun/Util.py:21-170, ModelTransformer.py:159-164).

Nothing here is a hand-written kernel: rFFT, a [2049 x 229] matmul and elementwise ops are what torch.fft / hipBLASLt are
for, and the backbone between these features and the interval scorer is out of scope.  The module exists
so that a caller who replaces the reference end to end finds the same pieces with the same parameter names
(`spectrogramExtractor.win`, `spectrogramExtractor.winGen.sigma|center`, `freq2mels`).

Parity: framing, the Gaussian windows and the spectrum are pinned against the reference's own classes (importable in the
build container; tests/golden/frontend.npz).  The mel filterbank is PARITY UNPINNED: the reference takes it from
torchaudio.functional.melscale_fbanks (Util.py:134-141), which is not installed here; `melscale_fbanks` below restates
torchaudio's documented definition (HTK mel scale, triangular filters, no area normalisation) and is checked for its
structural properties only.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def makeFrame(x: torch.Tensor, hopSize: int, windowSize: int, leftPaddingHalfFrame: bool = True) -> torch.Tensor:
    """[..., nSample] -> [..., nFrame, windowSize] with nFrame = ceil(nSample / hop) + 1 (Util.py:21-43): frame t is centred
    on sample t * hop when the signal is padded by half a window on the left."""
    assert hopSize < windowSize
    n = x.shape[-1]
    nFrame = math.ceil(n / hopSize) + 1
    left = windowSize // 2 if leftPaddingHalfFrame else 0
    covered = (nFrame - 1) * hopSize + windowSize            # samples the nFrame windows span
    right = covered - left - n
    frames = F.pad(x, (left, right)).unfold(-1, windowSize, hopSize)
    assert frames.shape[-2] == nFrame, (frames.shape[-2], nFrame)
    return frames


class GaussianWindows(nn.Module):
    """n learnable Gaussian analysis windows of length nWin (Util.py:47-71): centre and width are sigmoids of free parameters,
    initialised to evenly spaced centres and a common width."""

    def __init__(self, n: int, nWin: int):
        super().__init__()
        self.n, self.nWin = n, nWin
        self.sigma = nn.Parameter(torch.full((n,), -1.0))
        self.center = nn.Parameter(torch.logit(torch.arange(1, n + 1) / (n + 1)))

    def get(self) -> torch.Tensor:
        width = torch.sigmoid(self.sigma) * self.nWin / 2
        mid = torch.sigmoid(self.center) * self.nWin
        t = torch.arange(self.nWin, device=self.sigma.device).unsqueeze(1)
        return torch.exp(-0.5 * ((t - mid) / width) ** 2)               # [nWin, n]


class Spectrum(nn.Module):
    """Orthonormal rFFT of every frame under a Hann window and nExtraWins Gaussian windows (Util.py:78-124):
    [..., nFrame, windowSize] -> complex [..., nFrame, windowSize//2+1, 1+nExtraWins]."""

    def __init__(self, windowSize: int, nExtraWins: int = 0, log: bool = False):
        super().__init__()
        self.outputDim = windowSize // 2 + 1
        self.nChannel = nExtraWins + 1
        self.log = log
        self.nExtraWins = nExtraWins
        self.register_buffer("win", torch.hann_window(windowSize))
        if nExtraWins > 0:
            self.winGen = GaussianWindows(nExtraWins, windowSize)

    def windows(self) -> torch.Tensor:
        wins = self.win.unsqueeze(0)
        if self.nExtraWins > 0:
            wins = torch.cat([wins, self.winGen.get().t()], dim=0)
        return wins                                                          # [1+nExtraWins, windowSize]

    def forward(self, frames: torch.Tensor) -> torch.Tensor:
        spec = torch.fft.rfft(frames.unsqueeze(-2) * self.windows(), norm="ortho")
        if self.log:
            spec = torch.complex(spec.abs(), spec.angle())
        return spec.transpose(-1, -2)


def _hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """[n_freqs, n_mels] triangular filterbank: n_mels + 2 points equally spaced on the HTK mel scale between f_min and
    f_max; filter m rises from point m to point m+1 and falls to point m+2; linear frequency bins 0 .. sample_rate/2; no
    area normalisation (torchaudio.functional.melscale_fbanks with its defaults -- restated, parity unpinned)."""
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel_htk(f_min), _hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]                                          # [n_mels + 1]
    slopes = f_pts.unsqueeze(0) - freqs.unsqueeze(1)                         # [n_freqs, n_mels + 2]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


class MelSpectrum(nn.Module):
    """Power spectrum of every window -> mel bands -> optional log compression mapped to [0, 1] (Util.py:126-170):
    [..., nFrame, windowSize] -> [..., nFrame, n_mels, 1+nExtraWins]."""

    def __init__(self, windowSize, f_min, f_max, n_mels, fs, nExtraWins=0, log=False, eps=1e-5, toMono=False):
        super().__init__()
        self.outputDim = n_mels
        self.nChannel = nExtraWins + 1
        self.register_buffer("freq2mels", melscale_fbanks(windowSize // 2 + 1, f_min, f_max, n_mels, fs))
        self.log, self.eps, self.toMono = log, eps, toMono
        self.spectrogramExtractor = Spectrum(windowSize, nExtraWins)

    def forward(self, frames: torch.Tensor) -> torch.Tensor:
        power = self.spectrogramExtractor(frames).abs().pow(2)              # [..., nFrame, nFreq, nWin]
        if self.toMono and power.dim() >= 4:
            power = power.mean(dim=-4, keepdim=True)                         # over the audio channels
        mel = (power.transpose(-1, -2) @ self.freq2mels).transpose(-1, -2)
        if self.log:
            mel = ((mel + self.eps).log() - math.log(self.eps)) / (-math.log(self.eps))
        return mel


def normalize_gain(framesBatch: torch.Tensor) -> torch.Tensor:
    """Per-recording gain normalisation in front of the feature extractor (ModelTransformer.py:159-161)."""
    mean = torch.mean(framesBatch, dim=[1, 2, 3], keepdim=True)
    std = torch.std(framesBatch, dim=[1, 2, 3], keepdim=True)
    return (framesBatch - mean) / (std + 1e-8)
