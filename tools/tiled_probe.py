"""Cycle accounting of interval_score_tiled_kernel (probe build: SEMICRF_TILED_PROBE=1): per wave, the cycles between the stamps
around the vmcnt wait, the barrier, the requests, the two halves of the contraction and the stores.  GPU box only."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
from transkun_amd.scorer import _interval_score_raw
dev = torch.device("cuda:0")
lib = _lib.load()
lib.semicrf_debug_score_variant(2)
names = ["vmcnt wait", "stores", "requests", "barrier after S", "contraction", "barrier after M", "-", "loop/other"]
for (N, P, pitch, T, D) in [(1, 352, 352, 1024, 256), (4, 90, 96, 691, 256)]:
    C = N * P
    q = synth.hash_normal(C * T * D, 5, dev).view(C, T, D)
    k = synth.hash_normal(C * T * D, 6, dev).view(C, T, D)
    dg = synth.hash_normal(C * T, 7, dev).view(C, T)
    for _ in range(3):
        S, _ = _interval_score_raw(q, k, dg, T, C, D, 1.0 / 16, 0, 2, P, pitch)
    torch.cuda.synchronize()
    Cs = N * pitch
    pr = S.view(-1)[Cs:Cs + 256 * 8 * 8].view(256, 8, 8).double().cpu()
    tot = pr[:, :, [0, 1, 2, 3, 4, 5, 7]].sum(-1)
    print(f"T={T} C={C}: cycles per wave {tot.mean():.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        m = pr[:, sl].mean((0, 1))
        print("  " + grp + ": " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, m.tolist())))
lib.semicrf_debug_score_variant(-1)
