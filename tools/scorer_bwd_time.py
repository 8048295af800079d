import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from transkun_amd.scorer import ScaledInnerProductIntervalScorer
dev = torch.device("cuda:0")
for (N, P, T) in ((4, 88, 1024), (4, 90, 691), (1, 90, 691)):
    m = ScaledInnerProductIntervalScorer(256).to(dev)
    ctx = torch.randn(N, P, T, 256, device=dev, requires_grad=True)
    def step():
        S, b = m(ctx)
        g = torch.ones_like(S)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S.backward(g)
        torch.cuda.synchronize(); return time.perf_counter() - t0
    def fwd():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S, b = m(ctx)
        torch.cuda.synchronize(); return time.perf_counter() - t0
    for _ in range(2): step()
    tb = min(step() for _ in range(3)); tf = min(fwd() for _ in range(3))
    print(f"N={N} P={P} T={T}: scorer module fwd {tf*1e3:.2f} ms, bwd {tb*1e3:.2f} ms")
# kernel-only timing of interval_score_bwd
from transkun_amd import _lib
lib = _lib.load()
for (C, T, D) in ((352, 1024, 256), (360, 691, 256), (90, 691, 256)):
    dS = torch.randn(T, T, C, device=dev); q = torch.randn(C, T, D, device=dev); k = torch.randn(C, T, D, device=dev)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dd = torch.empty(C, T, device=dev)
    f = lambda: _lib.check(lib.interval_score_bwd(_lib.ptr(dS), _lib.ptr(q), _lib.ptr(k), C, T, D, D, D, 1.0 / 16, 0, _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dd), D, D, 1, _lib.stream_of(dS)), "bwd")
    for _ in range(2): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2 * 2.0 * C * (T * (T + 1) / 2) * D
    print(f"interval_score_bwd C={C} T={T} D={D}: {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s")
    nws = int(lib.interval_score_bwd_workspace_bytes(C, T, D))
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    f = lambda: _lib.check(lib.interval_score_bwd_ws(_lib.ptr(dS), _lib.ptr(q), _lib.ptr(k), C, T, D, D, D, 1.0 / 16, 0, _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dd), D, D, 1, _lib.ptr(ws), nws, _lib.stream_of(dS)), "bwd_ws")
    for _ in range(2): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"interval_score_bwd_ws (repack + 2 GEMMs) C={C} T={T} D={D}: {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s")
