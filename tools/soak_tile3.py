"""Randomised soak of the opt-in three-limb forward (interval_score_tile3_kernel) against the exact kernel: random T, segment counts,
symbols per segment, slot pitches (incl. ones with PADDING quads -- the round-6 race), D, length scaling, triangle / full square, with and
without the merged projection's row constant.  Checks: every real slot within the three-limb bound of the exact scores, ghost slots
exactly zero, two runs of the same call bit-identical (a race shows up as either).  GPU box only.   python tools/soak_tile3.py [seed] [cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
from transkun_amd.scorer import BF16X3, _interval_score_raw
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
worst = 0.0
for it in range(n_cases):
    T = rng.choice([256, 257, 300, 383, 384, 385, 512, 640, 691, 700, 1024])
    N = rng.choice([1, 1, 2, 3, 4])
    P = rng.choice([1, 2, 3, 5, 7, 9, 10, 12, 26, 31, 33, 44, 90])
    if N * P * T * T > 2.0e8: P = max(1, int(2.0e8 / (N * T * T)))
    pitch = P if rng.random() < 0.25 else (P + 3) // 4 * 4 + 4 * rng.choice([0, 1, 1, 2, 3])
    D = rng.choice([64, 128, 256])
    mode = rng.choice([0, 1, 2]); full = rng.choice([0, 1, 2, 2])
    use_rc = rng.random() < 0.6
    C = N * P
    q = synth.hash_normal(C * T * (D + 4), 1000 + it, dev).view(C, T, D + 4)[..., :D]
    k = synth.hash_normal(C * T * D, 2000 + it, dev).view(C, T, D)
    dg = synth.hash_normal(C * T, 3000 + it, dev).view(C, T)
    rc = synth.hash_normal(C * T, 4000 + it, dev).view(C, T) * 4.0 if use_rc else None
    qs = 1.0 / D ** 0.5
    tri = torch.tril(torch.ones(T, T, dtype=torch.bool, device=dev)).unsqueeze(-1)
    zero = torch.zeros((), device=dev)
    def run(fs):
        S, _ = _interval_score_raw(q, k, dg, T, C, D, qs, mode, fs, P, pitch, rowc=rc)
        return S.clone() if full == 1 else torch.where(tri, S, zero)
    ex, a, b = run(full), run(full | BF16X3), run(full | BF16X3)
    torch.cuda.synchronize()
    real = torch.zeros(N * pitch, dtype=torch.bool, device=dev)
    for n in range(N): real[n * pitch:n * pitch + P] = True
    scale = float(ex.abs().max()) + 1e-30
    err = float((ex - a)[:, :, real].abs().max()) / scale
    worst = max(worst, err)
    info = (it, T, N, P, pitch, D, mode, full, use_rc)
    assert torch.equal(a, b), ("not reproducible", info)
    assert err <= 2e-5, ("beyond the three-limb bound", info, err)
    if pitch > P: assert float(a[:, :, ~real].abs().max()) == 0.0, ("ghost slots", info)
    assert _lib.device_status() == 0
    del ex, a, b, q, k, dg, rc
print(f"{n_cases} random cases: three-limb forward within {worst:.2e} of the exact scores' largest value, reproducible, ghost slots zero")
