"""Probe: how does the torch-CPU op-loop port scale with threads on this host? (bench cpu_baseline sizing)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle as o
from transkun_amd import synth
T, B = 512, 88
s, n = synth.crf_inputs(T, B, 1, "cpu")
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    t = time.perf_counter(); o.oploop_forward_backward(s, n); dt = time.perf_counter() - t
    print(f"threads={th} T={T} B={B} fwd_bwd={dt:.3f}s", flush=True)
