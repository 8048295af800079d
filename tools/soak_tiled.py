"""Randomised bit-equality soak of interval_score_tiled_kernel (variant 2, the default) against the 128-row tile kernel: random
T, segment counts, symbols per segment, slot pitches, D, length scaling, triangle / full square, with and without the merged
projection's row constant, strided q rows.  GPU box only.  Usage: python tools/soak_tiled.py [seed] [cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import interval_score_variant        # the 128-row reference kernel lives in libsemicrf_hip_debug.so since round 4
from transkun_amd import _lib, synth
from transkun_amd.scorer import _interval_score_raw
dev = torch.device("cuda:0")
lib = _lib.load()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for it in range(n_cases):
    T = rng.choice([128, 129, 160, 191, 192, 193, 255, 256, 257, 300, 383, 384, 385, 512, 640, 691, 700, 1024])
    N = rng.choice([1, 1, 2, 3, 4])
    P = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 31, 32, 33, 44, 90])
    if N * P * T * T > 2.5e8: P = max(1, int(2.5e8 / (N * T * T)))
    pitch = P if rng.random() < 0.4 else (P + 3) // 4 * 4 + 4 * rng.choice([0, 1, 2])
    if pitch == P and N > 1 and rng.random() < 0.5: pitch = P     # contiguous
    D = rng.choice([64, 128, 192, 256])
    mode = rng.choice([0, 1, 2]); full = rng.choice([0, 0, 1, 2])
    use_rc = rng.random() < 0.5
    C = N * P
    pad = rng.choice([0, 4, 8])
    qb = synth.hash_normal(C * T * (D + pad), 1000 + it, dev).view(C, T, D + pad)
    q = qb[..., :D]
    k = synth.hash_normal(C * T * D, 2000 + it, dev).view(C, T, D)
    dg = synth.hash_normal(C * T, 3000 + it, dev).view(C, T)
    rc = synth.hash_normal(C * T, 4000 + it, dev).view(C, T) if use_rc else None
    qs = 1.0 / D ** 0.5
    outs = []
    for v in (128, 2):
        S, _ = interval_score_variant(v, q, k, dg, T, C, D, qs, mode, full, P, pitch, rowc=rc)
        if full == 2: S = torch.tril(S.permute(2, 0, 1)).contiguous()     # cells above the diagonal are not written
        elif full == 0: S = torch.tril(S.permute(2, 0, 1)).contiguous()   # (the debug entry point leaves the zero fill to its caller)
        outs.append(S)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), (it, T, N, P, pitch, D, mode, full, use_rc, pad, float((outs[0] - outs[1]).abs().max()))
    assert _lib.device_status() == 0
    del outs, q, qb, k, dg, rc
print(f"{n_cases} random cases: interval_score_tiled_kernel == 128-row tile kernel, bit for bit")
