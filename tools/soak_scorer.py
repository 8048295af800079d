"""Randomised shape sweep of the interval-scorer kernels: every forward variant against the register-load kernel, the packed
backward against the direct one (both exact-fp32 MFMA: differences are summation order only).  GPU box only."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
from transkun_amd.scorer import _interval_score_raw
lib = _lib.load(); dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst_f = worst_b = 0.0
for it in range(n_cases):
    T = rng.choice([64, 65, 96, 127, 128, 129, 160, 255, 256, 257, 300, 383, 384, 385, 512, 640, 691, 700])
    C = rng.choice([1, 2, 3, 4, 7, 8, 31, 32, 33, 44, 64, 90, 97])
    D = rng.choice([64, 128, 256])
    mode = rng.choice([0, 1, 2]); full = rng.random() < 0.3
    q = synth.hash_normal(C * T * D, 1000 + it, dev).view(C, T, D)
    k = synth.hash_normal(C * T * D, 2000 + it, dev).view(C, T, D)
    dg = synth.hash_normal(C * T, 3000 + it, dev).view(C, T)
    qs = 1.0 / D ** 0.5
    lib.semicrf_debug_score_variant(0)
    ref, _ = _interval_score_raw(q, k, dg, T, C, D, qs, mode, full)
    for v in (32, 64, 128, 2):
        lib.semicrf_debug_score_variant(v)
        got, _ = _interval_score_raw(q, k, dg, T, C, D, qs, mode, full)
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        worst_f = max(worst_f, err)
        assert err < 3e-6, (T, C, D, mode, full, v, err)
    lib.semicrf_debug_score_variant(-1)
    dS = synth.hash_normal(T * T * C, 4000 + it, dev).view(T, T, C)
    nws = int(lib.interval_score_bwd_workspace_bytes(C, T, D)); assert nws > 0
    ws = torch.full((nws,), 0xFF, dtype=torch.uint8, device=dev)
    outs = []
    for use in (True, False):
        dq = torch.full((C, T, D), float("nan"), device=dev); dk = torch.full((C, T, D), float("nan"), device=dev)
        dd = torch.full((C, T), float("nan"), device=dev)
        _lib.check(lib.interval_score_bwd_ws(_lib.ptr(dS), _lib.ptr(q), _lib.ptr(k), C, T, D, D, D, qs, mode, _lib.ptr(dq), _lib.ptr(dk),
                                             _lib.ptr(dd), D, D, 1, _lib.ptr(ws) if use else None, nws if use else 0, _lib.stream_of(dS)), "bwd")
        outs.append((dq, dk, dd))
    for a, b in zip(outs[0], outs[1]):
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        worst_b = max(worst_b, err)
        assert err < 5e-6 and not torch.isnan(a).any(), (T, C, D, mode, err)
    assert _lib.device_status() == 0
print(f"{n_cases} random shapes OK; worst relative difference forward {worst_f:.2e}, backward {worst_b:.2e}")
