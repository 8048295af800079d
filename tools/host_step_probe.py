"""Is the headline step host-bound?  Enqueue time per step (no sync) vs the time including the device work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import CRF, synth
dev = torch.device("cuda:0")
T, B = 1024, 352
score, noise = synth.crf_inputs(T, B, 1234, dev, "randn")
intervals = synth.synthetic_intervals(T, B, seed=1234)
score.requires_grad_(); noise.requires_grad_()
def step():
    score.grad = None; noise.grad = None
    lp = CRF.NeuralSemiCRFInterval(score, noise).logProb(intervals)
    (-lp.sum() / 4).backward()
for _ in range(5): step()
torch.cuda.synchronize()
for n in (20, 20, 100):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e3*(t1-t0)/n:.3f} ms/step, with device {1e3*(t2-t0)/n:.3f} ms/step, drain after the loop {1e3*(t2-t1):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
