"""Where a ring block's period goes (probe build with debug flag 16, spine workgroup 0): per block k, in shader cycles,
   period            = diag_end(k) - diag_end(k-1)
   late              = start of the `last` shadow batch (block k-1's entries) - diag_end(k-1): > 0: the owner of k was still busy
   last shadow       = diag_start - that start (includes waiting for block k-1's entries if the owner was early)
   far wait          = far partial in hand - diag_start
   diagonal phase    = diag_end - far partial in hand."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["SEMICRF_DEBUG_FLAGS"] = "16"; os.environ["SEMICRF_DEBUG_KEEP_WS"] = "1"
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--B", type=int, default=88)
a = ap.parse_args()
dev = torch.device("cuda:0")
s, n = synth.crf_inputs(a.T, a.B, 1234, dev)
for _ in range(3): nsci._logz_fwd_raw(s, n, True)
torch.cuda.synchronize()
ws = nsci._DEBUG_WS[0]
CT = 16 * 256 * 4
K = (a.T + 15) // 16
ev = ws[CT + a.T * 8:CT + a.T * 8 + K * 64].view(torch.int64).cpu().numpy().reshape(K, 8).astype(np.float64)
rows = []
for k in range(5, K):
    rows.append((ev[k, 7] - ev[k - 1, 7], ev[k, 5] - ev[k - 1, 7], ev[k, 6] - ev[k, 5], ev[k, 2] - ev[k, 6], ev[k, 7] - ev[k, 2],
                 ev[k, 1] - ev[k, 0], ev[k, 0] - ev[k - 4, 7]))
r = np.array(rows)
print(f"T={a.T} B={a.B}: cycles per block (mean / median over blocks 5..{K-1}); total {(ev[K-1,7]-ev[0,0]):.0f} cycles")
for i, name in enumerate(("period", "late (last-shadow start - mate's publish)", "last shadow (+ wait for entries)", "far wait", "diagonal phase",
                          "tiles+consts (iter start -> consts done)", "iter start - own previous publish")):
    print(f"  {name:45s} {r[:, i].mean():8.0f} {np.median(r[:, i]):8.0f}   p90 {np.percentile(r[:, i], 90):8.0f}")
for k in (8, 9, 10, 11, 20, 21, 22, 23, 40, 41, 42, 43):
    if k < K:
        print(f"   k={k:3d}: " + " ".join(f"{x:7.0f}" for x in rows[k - 5]))
