"""Per-step timestamps of the diagonal chain of spine workgroup 0 (debug flag 16).  GPU box only."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--B", type=int, default=352)
ap.add_argument("--flags", type=int, default=16)
a = ap.parse_args()
os.environ["SEMICRF_DEBUG_FLAGS"] = str(a.flags | 16); os.environ["SEMICRF_DEBUG_KEEP_WS"] = "1"
dev = torch.device("cuda:0")
s, n = synth.crf_inputs(a.T, a.B, 1234, dev)
for _ in range(3):
    nsci._logz_fwd_raw(s, n, True)
torch.cuda.synchronize()
ws = nsci._DEBUG_WS[0]
CT = 16 * 256 * 4
ts = ws[CT:CT + a.T * 8].view(torch.int64).cpu().numpy()
d = np.diff(ts) if ts.any() else np.zeros(a.T - 1)
print(f"T={a.T} B={a.B} flags={a.flags}: total {ts[-1]-ts[0]} ticks over {a.T-1} steps; mean {d.mean():.1f} median {np.median(d):.1f}")
j = np.arange(1, a.T)
for mod, name in ((16, "j%16==0 (block start)"), (8, "j%8==0 (chunk start)")):
    m = (j % mod) == 0
    print(f"  {name}: mean {d[m].mean():.1f} median {np.median(d[m]):.1f} max {d[m].max()}")
m = (j % 8) != 0
print(f"  other steps: mean {d[m].mean():.1f} median {np.median(d[m]):.1f} max {d[m].max()}")
print("  first 40 deltas:", d[:40].tolist())
print("  deltas 512..552:", d[512:552].tolist())

K = (a.T + 15) // 16
ev = ws[CT + a.T * 8:CT + a.T * 8 + K * 64].view(torch.int64).cpu().numpy().reshape(K, 8)
t0 = ev[0, 0]
print("  per-block events (cycles since start): iter_start, consts_done, loads_issued, shadow0, shadow1, shadow2, diag_start, diag_end")
print('  diag_start by block (us at 2.4 GHz):', [round((ev[k,6]-t0)/2400,1) for k in range(0,K,4)])
for k in list(range(0, 12)) + list(range(32, 38)):
    print(f"   k={k:3d} wave={k%4}: " + " ".join(f"{(x - t0) if x not in (0, -1) else 0:9d}" for x in ev[k]))
