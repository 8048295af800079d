"""Gradient-sweep timeline (probe build with debug flag 16): publish time of every 4th block of spine 0 and the ring's per-block
cycle split, like chain_trace.py / ring_trace.py for the forward sweep.  GPU box only."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["SEMICRF_DEBUG_FLAGS"] = "16"
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--B", type=int, default=352); ap.add_argument("--fwd", type=int, default=0)
a = ap.parse_args()
T, B = a.T, a.B
K = (T + 15) // 16
dev = torch.device("cuda:0")
s, n = synth.crf_inputs(T, B, 1234, dev)
lz, v = nsci._logz_fwd_raw(s, n, True); g = torch.ones(B, device=dev)
ws = _lib.leased_workspace(_lib.OP_LOGZ_BWD, T, B, dev)
dn = torch.empty_like(n); q = torch.empty(0, device=dev); ds = torch.zeros(T, T, B, device=dev)
if a.fwd:
    ws = _lib.leased_workspace(_lib.OP_LOGZ_FWD, T, B, dev)
    for _ in range(3): _lib.ops().logz_fwd(s, n, lz, v, True, ws)
else:
    for _ in range(3): _lib.ops().logz_bwd(s, n, v, lz, g, ds, dn, q, False, nsci.GRAD_UPPER_IS_ZERO, ws)
torch.cuda.synchronize()
CT = 16 * 256 * 4
ts = ws[CT:CT + 2 * T * 8].view(torch.int64).cpu().numpy().astype(np.float64) / 100.0
pub, got, far = ts[0:64], ts[64:128], ts[128:192]
t0 = pub[0]
print(f"GRAD T={T} B={B}: publish time of block k (us):", [round(float(pub[k] - t0), 1) for k in range(0, min(K, 64), 4)])
per = np.diff(pub[:min(K, 64)])
print(f"  block period: mean {per.mean():.2f} median {np.median(per):.2f} p90 {np.percentile(per, 90):.2f}; total {pub[min(K,64)-1]-t0:.1f} us")
ev = ws[CT + T * 8:CT + T * 8 + K * 64].view(torch.int64).cpu().numpy().reshape(K, 8).astype(np.float64)
rows = []
for k in range(5, K):
    rows.append((ev[k, 7] - ev[k - 1, 7], ev[k, 6] - ev[k, 5], ev[k, 2] - ev[k, 6], ev[k, 7] - ev[k, 2], ev[k, 1] - ev[k, 0]))
r = np.array(rows)
for i, name in enumerate(("period", "last shadow (+ wait)", "far wait", "diagonal phase", "tiles+consts")):
    print(f"  {name:25s} mean {r[:, i].mean():8.0f} median {np.median(r[:, i]):8.0f} p90 {np.percentile(r[:, i], 90):8.0f}")
seen, stored = ts[192:256], ts[256:320]
stored_q = np.stack([ts[256 + 64 * i:320 + 64 * i] for i in range(4)])
ks = np.arange(8, min(K, 64))
p = pub[ks - 4]
for name, arr in (("seen", seen), ("stored", stored), ("far", far), ("owner", got)):
    d = arr[ks] - p
    print(f"  {name:7s}: mean {d.mean():5.2f}  median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f}")
d = stored_q[:, ks] - p[None, :]
print("  stored per quarter: mean %s; slowest quarter mean %.2f p90 %.2f; far - slowest: mean %.2f p90 %.2f" % (
    np.round(d.mean(1), 2).tolist(), d.max(0).mean(), np.percentile(d.max(0), 90), (far[ks] - p - d.max(0)).mean(),
    np.percentile(far[ks] - p - d.max(0), 90)))
for k in list(range(8, 40, 3)) + [48, 56, 63]:
    if k >= K: break
    p0 = pub[k - 4]
    print(f"{k:3d}: {p0 - t0:8.2f}  seen {seen[k] - p0:6.2f}  stored {stored[k] - p0:6.2f} (q max {stored_q[:, k].max() - p0:6.2f})  far {far[k] - p0:6.2f}  owner {got[k] - p0:6.2f}   | {pub[k] - pub[k-1]:5.2f}")
if ts.shape[0] >= 960 and ts[640:960].any():
    st, po = ts[640:704], ts[896:960]
    ks2 = np.arange(8, min(K, 64))
    print("  far wave: start after publish(k-4): mean %.2f; partials-in: mean %.2f" % ((st[ks2] - pub[ks2 - 4]).mean(), (po[ks2] - pub[ks2 - 4]).mean()))
if T >= 1024:
    A, Bq, D, E, NP = ts[1536:1600], ts[1600:1664], ts[1664:1728], ts[1728:1792], ts[1792:1856] * 100.0
    ks = np.arange(8, min(K, 64))
    p = pub[ks - 4]
    print("  newest tile of the traced task, us after publish(k-4): iteration start %.2f, stage in %.2f, seen %.2f, math done %.2f, drained %.2f, stored %.2f; polls %.1f" % (
        (A[ks] - p).mean(), (Bq[ks] - p).mean(), (seen[ks] - p).mean(), (D[ks] - p).mean(), (E[ks] - p).mean(), (stored[ks] - p).mean(), NP[ks].mean()))
    for k in list(range(8, 40, 3)):
        p0 = pub[k - 4]
        print(f"   {k:3d}: start {A[k]-p0:6.2f} stage {Bq[k]-p0:6.2f} seen {seen[k]-p0:6.2f} math {D[k]-p0:6.2f} drained {E[k]-p0:6.2f} stored {stored[k]-p0:6.2f} polls {NP[k]:.0f}")
