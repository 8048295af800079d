"""Sweep table of BASELINE.md's grid: T in {200,512,1024,2048} x NBatch in {4,88,176,352}: log-partition forward,
fused backward (gradient sweep) and device-level decode, kernel-level timing (HIP events).  GPU box only."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")

def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

print("| T | NBatch | logZ fwd us | GB/s (algorithmic) | frac of 8 TB/s | grad sweep us | decode us (device) | chains/s decode |")
print("|---|---|---|---|---|---|---|---|")
for T in (200, 512, 1024, 2048):
    for B in (4, 88, 176, 352):
        s, n = synth.crf_inputs(T, B, 1234, dev)
        ab = 4 * B * (T * (T + 1) // 2 + T - 1)
        f = timeit(lambda: nsci._logz_fwd_raw(s, n, True))
        lz, v = nsci._logz_fwd_raw(s, n, True); g = torch.ones(B, device=dev)
        b = timeit(lambda: nsci._logz_bwd_raw(s, n, v, lz, g)) if T * T * B * 4 * 2 < 12e9 else float("nan")
        d = timeit(lambda: nsci._viterbi_raw(s, n, None, False), n=5)
        print(f"| {T} | {B} | {f:.1f} | {ab / f / 1e3:.0f} | {ab / f / 1e3 / 8000:.3f} | {b:.1f} | {d:.1f} | {B / d * 1e6:.0f} |", flush=True)
        del s, n, lz, v
        torch.cuda.empty_cache()
