import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import synth
from transkun_amd.trainstep import SegmentModel, train_step
dev = torch.device("cuda:0")
Ts, P, D, N = 691, 90, 256, 4
torch.manual_seed(0)
model = SegmentModel(D).to(dev)
if len(sys.argv) > 1: model.scorer.contraction = sys.argv[1]
ctx = synth.hash_normal(N * P * Ts * D, 21, dev).view(N, P, Ts, D) * 0.5
iv = synth.synthetic_intervals(Ts, N * P, seed=21)
"""The train.py-shaped step (4 segments x 90 symbols x T=691) on its own: the workload of `rocprofv3 --kernel-trace --stats` in
profiles/ (which kernels a step is made of).  GPU box only."""
ctx.requires_grad_()
def step():
    ctx.grad = None                      # ctx is an intermediate tensor in the model: no accumulation into a leaf's .grad
    train_step(model, ctx, iv)
for _ in range(3): step()
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t) / 10 * 1e3)
