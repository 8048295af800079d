"""The scorer kernel families in one process, a few launches each (for rocprofv3 passes): interval_score_tiled_kernel (exact fp32,
the default) and interval_score_tile_kernel<128> beside it, interval_score_tile3_kernel (three-limb bf16) and the packed backward (score_bwd_pack_kernel + score_bwd_gemm_kernel), at
T=1024 x 352 chains x D=256 and at the model's shape in the slot layout (T=691, 4 x 90 symbols at pitch 96).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
from transkun_amd.scorer import _interval_score_raw, bwd_workspace
dev = torch.device("cuda:0")
ops = _lib.ops()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only_big = len(sys.argv) > 2 and sys.argv[2] == "big"       # the PMC passes: one shape, so that a kernel's counters are not a mix


def timeit(fn, n=n, warm=2):
    for _ in range(warm): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, P, pitch, T, D) in ((1, 352, 352, 1024, 256), (4, 90, 96, 691, 256))[:1 if only_big else 2]:
    C = N * P
    q = synth.hash_normal(C * T * D, 5, dev).view(C, T, D)
    k = synth.hash_normal(C * T * D, 6, dev).view(C, T, D)
    dg = synth.hash_normal(C * T, 7, dev).view(C, T)
    fl = 2.0 * C * (T * (T + 1) / 2) * D
    # interval_score_tiled_kernel (the default; the 64- / 128-row reference tiles live in the debug library since round 4)
    f32 = timeit(lambda: _interval_score_raw(q, k, dg, T, C, D, 1.0 / 16, 0, 2, P, pitch), n, 1)
    b3 = timeit(lambda: _interval_score_raw(q, k, dg, T, C, D, 1.0 / 16, 0, 2 | 4, P, pitch))
    S, _ = _interval_score_raw(q, k, dg, T, C, D, 1.0 / 16, 0, 0, P, pitch)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dd = torch.empty_like(dg)
    ws = bwd_workspace(C, T, D, dev)
    bw = timeit(lambda: ops.interval_score_bwd_ws(S, q, k, C, T, D, D, D, 1.0 / 16, 0, P, pitch, dq, dk, dd, dd, D, D, 1, 0, ws))
    bw3 = timeit(lambda: ops.interval_score_bwd_ws(S, q, k, C, T, D, D, D, 1.0 / 16, 16, P, pitch, dq, dk, dd, dd, D, D, 1, 0, ws))
    print(f"T={T} chains={C} (groups of {P} at pitch {pitch}) D={D}: fwd fp32 {f32:.3f} ms = {fl / f32 / 1e9:.1f} TFLOP/s ({fl / f32 / 1e9 / 157.3:.3f} of 157.3), "
          f"fwd bf16x3 {b3:.3f} ms = {fl / b3 / 1e9:.1f} TFLOP/s fp32-equivalent, bwd (pack + 2 GEMMs) {bw:.3f} ms = {2 * fl / bw / 1e9:.1f} TFLOP/s, bwd bf16x3 {bw3:.3f} ms = {2 * fl / bw3 / 1e9:.1f} TFLOP/s fp32-equivalent", flush=True)
    del q, k, dg, S, dq, dk, dd, ws
    torch.cuda.empty_cache()
