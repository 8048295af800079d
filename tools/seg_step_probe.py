#!/usr/bin/env python3
"""One segment-shaped scorer + CRF logProb step (T=691, 90 symbols, N segments) on its own: wall time per step, device time per
step (events), and -- under `rocprofv3 --kernel-trace --stats` -- which kernels it is made of.
   python tools/seg_step_probe.py [N] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transkun_amd import synth
from transkun_amd.fused import scorer_crf_logprob
from transkun_amd.scorer import ScaledInnerProductIntervalScorer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
Ts, P, D = 691, 90, 256
m = ScaledInnerProductIntervalScorer(D, 1).to(dev)
ctx = (synth.hash_normal(N * P * Ts * D, 11, dev).view(N, P, Ts, D) * 0.5).requires_grad_()
iv = synth.synthetic_intervals(Ts, N * P, seed=11)


def step():
    m.zero_grad()
    ctx.grad = None
    lp = scorer_crf_logprob(m, ctx, iv)
    (-lp.view(N, -1).sum(-1).mean() / 50).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
for rep in range(3):                 # (the first repetition still grows the allocators' pools)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter()
    a.record()
    for _ in range(steps):
        step()
    b.record()
    cpu_done = time.perf_counter() - t
    torch.cuda.synchronize()
    wall = time.perf_counter() - t
    print(f"N={N} repetition {rep}: device {a.elapsed_time(b) / steps:.3f} ms per step, wall {wall / steps * 1e3:.3f}, host-side enqueue {cpu_done / steps * 1e3:.3f}")
if os.environ.get("SEG_PROBE_PROFILE"):
    import cProfile
    import pstats
    ts = []
    for _ in range(10):
        t = time.perf_counter()
        step()
        ts.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize()
    print("host ms per step:", " ".join(f"{v:.2f}" for v in ts))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
