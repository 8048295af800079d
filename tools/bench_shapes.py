"""The shapes the model runs (T=691, 90 symbols x 1 / 4 segments), their 96-slot paddings and BASELINE's grid corners: forward
sweep, gradient sweep, device decode.  Kernel-level timing (HIP events).  GPU box only."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(691, 90), (691, 96), (691, 360), (691, 384), (1024, 88), (1024, 96), (1024, 352), (2048, 352)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
print("| T | NBatch | logZ fwd us | frac of 8 TB/s | grad sweep us | decode us (device) |")
print("|---|---|---|---|---|---|")
for T, B in shapes:
    for kind in ("randn",):
        s, n = synth.crf_inputs(T, B, 1234, dev, kind)
        ab = 4 * B * (T * (T + 1) // 2 + T - 1)
        f = timeit(lambda: nsci._logz_fwd_raw(s, n, True))
        lz, v = nsci._logz_fwd_raw(s, n, True); g = torch.ones(B, device=dev)
        b = timeit(lambda: nsci._logz_bwd_raw(s, n, v, lz, g)) if T * T * B * 4 * 2 < 12e9 else float("nan")
        d = timeit(lambda: nsci._viterbi_raw(s, n, None, False), n=5)
        print(f"| {T} | {B} | {f:.1f} | {ab / f / 1e3 / 8000:.3f} | {b:.1f} | {d:.1f} |", flush=True)
        del s, n, lz, v
        torch.cuda.empty_cache()
