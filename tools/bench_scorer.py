"""Timing of the interval-score kernel (HIP events).  GPU box only."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
from transkun_amd.scorer import _interval_score_raw
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--C", type=int, default=352); ap.add_argument("--D", type=int, default=256)
a = ap.parse_args()
dev = torch.device("cuda:0")
y = synth.hash_normal(a.C * a.T * (2 * a.D + 1), 5, dev).view(a.C, a.T, 2 * a.D + 1)
q, k, dg = y[..., :a.D].contiguous(), y[..., a.D:2 * a.D].contiguous(), y[..., 2 * a.D].contiguous()
for impl in (0, 1):
    if impl == 1 and a.T * a.C > 200 * 400: continue
    _lib.set_impl(impl)
    for _ in range(2): _interval_score_raw(q, k, dg, a.T, a.C, a.D, 1.0 / 16, 0, False)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 5
    for _ in range(n): S, _ = _interval_score_raw(q, k, dg, a.T, a.C, a.D, 1.0 / 16, 0, False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * a.C * (a.T * (a.T + 1) / 2) * a.D
    print(f"impl={impl} T={a.T} C={a.C} D={a.D}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s (lower triangle)  out {4*a.C*a.T*(a.T+1)/2/ms/1e6:.0f} GB/s", flush=True)
