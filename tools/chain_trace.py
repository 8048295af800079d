"""Hand-off chain probes (probe build, debug flag 16): publish u(k-4) -> panel sees it -> partial stored -> far wave -> owner."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["SEMICRF_DEBUG_FLAGS"] = str(16 | int(os.environ.get("CHAIN_TRACE_FLAGS", "0"))); os.environ["SEMICRF_DEBUG_KEEP_WS"] = "1"
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--B", type=int, default=352)
a = ap.parse_args()
T, B = a.T, a.B
K = (T + 15) // 16
dev = torch.device("cuda:0")
s, n = synth.crf_inputs(T, B, 1234, dev)
for _ in range(3): nsci._logz_fwd_raw(s, n, True)
torch.cuda.synchronize()
ws = nsci._DEBUG_WS[0]
CT = 16 * 256 * 4
ts = ws[CT:CT + T * 8].view(torch.int64).cpu().numpy().astype(np.float64) / 100.0     # s_memrealtime: 100 MHz -> us
pub, got, far, seen, stored = ts[0:64], ts[64:128], ts[128:192], ts[192:256], ts[256:320]
stored_q = np.stack([ts[256 + 64 * i:320 + 64 * i] for i in range(4)])          # per row quarter
t0 = pub[0]
print(f"T={T} B={B}: k: publish(k-4)  seen(+)  stored(+)  far(+)  owner(+)   | block period")
for k in list(range(4, 40, 3)) + [48, 56, 63]:
    if k >= K: break
    p = pub[k - 4]
    print(f"{k:3d}: {p - t0:8.2f}  {seen[k] - p:6.2f}  {stored[k] - p:6.2f}  {far[k] - p:6.2f}  {got[k] - p:6.2f}   | {pub[k] - pub[k-1]:5.2f}")
ks = np.arange(4, min(K, 64))
p = pub[ks - 4]
for name, arr in (("seen", seen), ("stored", stored), ("far", far), ("owner", got)):
    d = arr[ks] - p
    print(f"  {name:7s}: mean {d.mean():5.2f}  median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f}")
if stored_q[1:].any():
    d = stored_q[:, ks] - p[None, :]
    print("  stored per quarter: mean %s; slowest quarter mean %.2f p90 %.2f; far - slowest: mean %.2f p90 %.2f" % (
        np.round(d.mean(1), 2).tolist(), d.max(0).mean(), np.percentile(d.max(0), 90), (far[ks] - p - d.max(0)).mean(),
        np.percentile(far[ks] - p - d.max(0), 90)))
if ts.shape[0] >= 960 and ts[640:960].any():
    st, ce, ri, ne, po = ts[640:704], ts[704:768], ts[768:832], ts[832:896], ts[896:960]
    print("  far wave phases, us after publish(k-4): k: start  cells  ring-entries  near-done  partials-in | entry written")
    for k in list(range(5, 40, 3)) + [48, 56, 63]:
        if k >= K: break
        p0 = pub[k - 4]
        print(f"  {k:3d}: {st[k]-p0:7.2f} {ce[k]-p0:7.2f} {ri[k]-p0:7.2f} {ne[k]-p0:7.2f} {po[k]-p0:7.2f} | {far[k]-p0:7.2f}")
    ks2 = np.arange(6, min(K, 64))
    for name, arr in (("start", st), ("cells", ce), ("ring", ri), ("near", ne), ("partials", po)):
        d = arr[ks2] - pub[ks2 - 4]
        print(f"    {name:9s}: mean {d.mean():6.2f} median {np.median(d):6.2f}")
print("  publish time of block k (us):", [round(float(pub[k] - t0), 1) for k in range(0, min(K, 64), 4)])
per = np.diff(pub[:min(K, 64)])
ctl = ws[:1024].view(torch.int32).cpu().numpy().astype(np.int64)
print("  probes: idle peeks %d, tasks last-part %d, full-part %d; prog %d; heads %s" % (
    (ctl[80] + 1) % 2**32, (ctl[81] + 1) % 2**32, (ctl[82] + 1) % 2**32, ctl[128],
    [int((x + 1) % 2**32) for x in ctl[32:37]]))
print(f"  block period: mean {per.mean():.2f} median {np.median(per):.2f} p90 {np.percentile(per, 90):.2f}; total {pub[min(K,64)-1]-t0:.1f} us")
