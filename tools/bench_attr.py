"""Interval features for the attribute heads: the reference's formulation (Python lists -> index tensors -> index_select ->
cat, per segment; ModelTransformer.py:501-532, :578-582) in plain torch on the GPU vs the HIP gather (from lists, and from
the packed decode output that is already on the device).  GPU box only."""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import attributes, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=4); ap.add_argument("--SYM", type=int, default=90)
ap.add_argument("--T", type=int, default=691); ap.add_argument("--D", type=int, default=256)
a = ap.parse_args()
dev = torch.device("cuda:0")
N, SYM, T, D = a.N, a.SYM, a.T, a.D
ctx = synth.hash_normal(N * SYM * T * D, 9, dev).view(N, SYM, T, D)
flat = synth.synthetic_intervals(T, N * SYM, seed=5)
batch = [flat[n * SYM:(n + 1) * SYM] for n in range(N)]
K = sum(len(x) for x in flat)

def torch_formulation():
    ca, cb = [], []
    for idx, cur in enumerate(batch):
        ints = sum(cur, [])
        if len(ints) > 0:
            symIdx = torch.tensor([i for i, l in enumerate(cur) for _ in l], dtype=torch.long, device=dev)
            ind = torch.tensor(ints, dtype=torch.long, device=dev)
            rows = ctx[idx].flatten(0, 1)
            ca.append(rows.index_select(0, ind[:, 0] + symIdx * T)); cb.append(rows.index_select(0, ind[:, 1] + symIdx * T))
    ca = torch.cat(ca); cb = torch.cat(cb)
    return torch.cat([ca, cb, ca * cb], dim=-1)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

pairs, offsets = nsci.pack_intervals(flat, T, N * SYM, dev)
ref = torch_formulation()
out, _, _ = attributes.attribute_input_packed(ctx, pairs, offsets, K)
assert torch.equal(out, ref)
t_ref = timeit(torch_formulation)
t_lists = timeit(lambda: attributes.fetchIntervalFeaturesBatch(ctx, batch))
t_packed = timeit(lambda: attributes.attribute_input_packed(ctx, pairs, offsets, K))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): attributes.attribute_input_packed(ctx, pairs, offsets, K)
e1.record(); torch.cuda.synchronize()
t_dev = e0.elapsed_time(e1) / 50
byt = K * 5 * D * 4
print(f"N={N} SYM={SYM} T={T} D={D}: {K} intervals")
print(f"  reference formulation in torch (lists -> index tensors -> index_select/cat per segment): {t_ref:.3f} ms")
print(f"  mirror from Python lists (one pack + one kernel):                                        {t_lists:.3f} ms")
print(f"  packed intervals already on the device (decode output):                                  {t_packed:.3f} ms wall, "
      f"{t_dev*1e3:.1f} us device = {byt / t_dev / 1e6:.1f} GB/s of the {byt/1e6:.2f} MB it must move")
