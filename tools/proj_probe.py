#!/usr/bin/env python3
"""Time the projection kernels (csrc/proj_gemm.hip) beside torch's GEMMs at the scorer's shapes: forward, input gradient, weight
gradient.  python tools/proj_probe.py [--rows 62190,248760] [--n 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from transkun_amd.scorer import QPAD, proj_forward, proj_input_grad, proj_weight_grad


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="62190,248760")
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--nmain", type=int, default=256)
    args = ap.parse_args()
    gpu = torch.device("cuda:0")
    for M in [int(v) for v in args.rows.split(",")]:
        K, n_main = args.k, args.nmain
        Nout = n_main + QPAD
        x = torch.randn(M, K, device=gpu)
        W = torch.randn(Nout, K, device=gpu)
        b = torch.randn(Nout, device=gpu)
        W[n_main + 2:] = 0
        b[n_main + 2:] = 0
        dy = torch.randn(M, Nout, device=gpu)
        flop = 2.0 * M * K * (n_main + 2)
        rows = [("forward", lambda: proj_forward(x, W, b, n_main), lambda: F.linear(x, W, b)),
                ("input gradient", lambda: proj_input_grad(dy, W), lambda: dy.mm(W)),
                ("weight gradient", lambda: proj_weight_grad(dy, x, n_main), lambda: (dy.t().mm(x), dy.sum(0)))]
        for name, ours, ref in rows:
            t1, t2 = timed(ours, args.n), timed(ref, args.n)
            print(f"M={M} K={K} N={n_main}+2  {name:16s} library {t1:8.1f} us ({flop / t1 * 1e-6:6.1f} TFLOP/s)   torch {t2:8.1f} us ({flop / t2 * 1e-6:6.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
