import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")
T, B = 333, 46
s, n = synth.crf_inputs(T, B, 77, dev)
g = synth.hash_normal(B, 5, dev)
lz, v = nsci._logz_fwd_raw(s, n, True)
outs = []
for i in range(6):
    ds, dn, q = nsci._logz_bwd_raw(s, n, v, lz, g, True)
    outs.append((ds.clone(), dn.clone(), q.clone()))
for i in range(1, 6):
    for name, a, b in zip(("ds", "dn", "q"), outs[0], outs[i]):
        if not torch.equal(a, b):
            d = (a - b).abs()
            idx = torch.nonzero(d > 0)
            print(i, name, "differs at", idx.shape[0], "places; max", float(d.max()), "first", idx[:5].tolist(), "vals", [ (float(a[tuple(j)]), float(b[tuple(j)])) for j in idx[:3].tolist()])
_lib.set_impl(1)
ds1, dn1, q1 = nsci._logz_bwd_raw(s, n, v, lz, g, True)
_lib.set_impl(0)
print("vs rowseq impl: max abs diff ds", float((outs[0][0] - ds1).abs().max()), "dn", float((outs[0][1] - dn1).abs().max()))
