"""Interval-scorer kernels over the BASELINE grid (D = 256): forward, backward (packed and direct), HIP events.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
from transkun_amd.scorer import _interval_score_raw
lib = _lib.load(); dev = torch.device("cuda:0"); D = 256
def timeit(fn, n=5):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
print("| T | chains | fwd ms | fwd TFLOP/s | bwd (packed) ms | bwd TFLOP/s | bwd (direct) ms | workspace GB |")
print("|---|---|---|---|---|---|---|---|")
for T in (200, 512, 691, 1024, 2048):
    for C in (88, 352):
        if T == 2048 and C == 352: continue            # 5.9 GB score + 5.9 GB gradient: skipped to keep the run short
        q = synth.hash_normal(C * T * D, 5, dev).view(C, T, D); k = synth.hash_normal(C * T * D, 6, dev).view(C, T, D)
        dg = synth.hash_normal(C * T, 7, dev).view(C, T)
        S = [None]
        def fwd(): S[0] = _interval_score_raw(q, k, dg, T, C, D, 1.0 / 16, 0, False)[0]
        tf = timeit(fwd)
        dq = torch.empty_like(q); dk = torch.empty_like(k); dd = torch.empty_like(dg)
        nws = int(lib.interval_score_bwd_workspace_bytes(C, T, D)); ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
        def bwd(use):
            _lib.check(lib.interval_score_bwd_ws(_lib.ptr(S[0]), _lib.ptr(q), _lib.ptr(k), C, T, D, D, D, 1.0 / 16, 0, _lib.ptr(dq),
                                                 _lib.ptr(dk), _lib.ptr(dd), D, D, 1, _lib.ptr(ws) if use else None, nws if use else 0,
                                                 _lib.stream_of(q)), "bwd")
        tb = timeit(lambda: bwd(True)); td = timeit(lambda: bwd(False))
        fl = 2.0 * C * (T * (T + 1) / 2) * D
        print(f"| {T} | {C} | {tf:.3f} | {fl/tf/1e9:.1f} | {tb:.3f} | {2*fl/tb/1e9:.1f} | {td:.3f} | {nws/1e9:.2f} |", flush=True)
        del q, k, dg, dq, dk, dd, ws; S[0] = None
