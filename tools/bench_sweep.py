"""Kernel-level timing of the sweeps (HIP events on torch's stream).  Usage on the GPU box:
   python tools/bench_sweep.py [--T 1024 --B 352] [--flags 0,1,3,12] [--ops fwd,bwd,vit]"""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")

def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--B", type=int, default=352)
ap.add_argument("--flags", default="0"); ap.add_argument("--ops", default="fwd")
ap.add_argument("--impl", type=int, default=0); ap.add_argument("--n", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
_lib.set_impl(a.impl)
s, n = synth.crf_inputs(a.T, a.B, 1234, dev)
L = a.T * (a.T + 1) // 2
ab = 4 * a.B * (L + a.T - 1)
for fl in a.flags.split(","):
    os.environ["SEMICRF_DEBUG_FLAGS"] = fl
    for op in a.ops.split(","):
        if op == "fwd":
            us = timeit(lambda: nsci._logz_fwd_raw(s, n, True), a.n)
        elif op == "bwd":
            lz, v = nsci._logz_fwd_raw(s, n, True); g = torch.ones(a.B, device=dev)
            us = timeit(lambda: nsci._logz_bwd_raw(s, n, v, lz, g), a.n)
        elif op == "beta":
            from transkun_amd.fused import _beta_raw
            us = timeit(lambda: _beta_raw(s, n), a.n)
        elif op == "vit":
            crf = nsci.NeuralSemiCRFInterval(s, n)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(3): crf.decode()
            torch.cuda.synchronize(); us = (time.perf_counter() - t) / 3 * 1e6
        print(f"T={a.T} B={a.B} impl={a.impl} flags={fl} op={op}: {us:.1f} us  ({ab / us / 1e3:.1f} GB/s algorithmic per sweep)  status={_lib.device_status()}", flush=True)
