#!/usr/bin/env python3
"""Which calls of the hot paths synchronise with the device?  Runs each path once warm under
torch.cuda.set_sync_debug_mode("warn") and prints the warnings (file:line of the synchronising torch call) per path.
Found this way in round 4: the argument check of torch.distributions.ContinuousBernoulli in the transcription loop."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transkun_amd import CRF, synth
from transkun_amd.fused import scorer_crf_logprob
from transkun_amd.scorer import ScaledInnerProductIntervalScorer
from transkun_amd.transcribe import SegmentTranscriber
from transkun_amd.trainstep import SegmentModel, train_step

dev = torch.device("cuda:0")


def audit(name, fn, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        fn()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    hits = {}
    for x in w:
        if "synchroniz" in str(x.message):
            key = f"{os.path.relpath(x.filename)}:{x.lineno}"
            hits[key] = hits.get(key, 0) + 1
    print(f"{name}: {sum(hits.values())} synchronising call(s)" + ("" if not hits else ": " + ", ".join(f"{k} x{v}" for k, v in hits.items())))


T, B = 1024, 352
score, noise = synth.crf_inputs(T, B, 1234, dev, "randn")
iv = synth.synthetic_intervals(T, B, seed=1234)
score.requires_grad_(); noise.requires_grad_()


def headline():
    score.grad = None; noise.grad = None
    lp = CRF.NeuralSemiCRFInterval(score, noise).logProb(iv)
    (lp.sum() * -0.25).backward()


audit("logProb forward + backward (T=1024 x 352)", headline)
audit("decode, Python lists", lambda: CRF.NeuralSemiCRFInterval(score.detach(), noise.detach()).decode())
del score, noise
Ts, P, D, N = 691, 90, 256, 2
m = ScaledInnerProductIntervalScorer(D, 1).to(dev)
ctx = (synth.hash_normal(N * P * Ts * D, 11, dev).view(N, P, Ts, D) * 0.5).requires_grad_()
iv2 = synth.synthetic_intervals(Ts, N * P, seed=11)


def fused():
    m.zero_grad(); ctx.grad = None
    (-scorer_crf_logprob(m, ctx, iv2).view(N, -1).sum(-1).mean() / 50).backward()


def unfused():
    m.zero_grad(); ctx.grad = None
    S, b = m(ctx)
    lp = CRF.NeuralSemiCRFInterval(S.flatten(-2, -1), b.flatten(-2, -1)).logProb(iv2)
    (-lp.view(N, -1).sum(-1).mean() / 50).backward()


audit("scorer + CRF logProb step, fused", fused)
audit("scorer + CRF logProb step, module + CRF API", unfused)
model = SegmentModel(D).to(dev)
audit("train_step", lambda: (ctx.__setattr__("grad", None), train_step(model, ctx, iv2)))
tr = SegmentTranscriber(D).to(dev).eval()
n_audio = int(40.0 * tr.fs)
plan = tr.segment_plan(n_audio)
cs = [(synth.hash_normal(P * plan["nFrame"] * D, 31 + i, dev).view(1, P, plan["nFrame"], D) * 0.5) for i in range(2)]
fns = [(lambda i, T_, f=f: cs[(i + f) % 2]) for f in range(2)]
audit(f"transcribe_many, 2 recordings x {len(plan['begins'])} steps", lambda: tr.transcribe_many(fns, [n_audio] * 2))
