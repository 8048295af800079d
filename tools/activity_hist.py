"""Activity histogram of the far-field waves (probe build, debug flag 1024): tiles taken per 4 us of the sweep, by the panel
workgroups' waves and by the spare waves of the spine workgroups -> the streaming rate over the course of the launch.
(Every XCD's tiles are bucketed against the start of its own first workgroup: s_memrealtime counts per XCD.)"""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--B", type=int, default=352)
ap.add_argument("--flags", type=int, default=0)
a = ap.parse_args()
os.environ["SEMICRF_DEBUG_FLAGS"] = str(a.flags | 1024 | 16); os.environ["SEMICRF_DEBUG_KEEP_WS"] = "1"
import numpy as np, torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
T, B = a.T, a.B
s, n = synth.crf_inputs(T, B, 1234, torch.device("cuda:0"))
for _ in range(3): nsci._logz_fwd_raw(s, n, True)
torch.cuda.synchronize()
ws = nsci._DEBUG_WS[0]
CT = 16 * 256 * 4
raw = ws[CT:CT + 2 * T * 8].view(torch.int64).cpu().numpy()
hb = raw[(3 * T) // 2:]
t0 = int(hb[0])
hp = hb[1:65] + 1; hh = hb[65:129] + 1      # counters start at all-ones
pub = raw[0:64].astype(np.float64)
print(f"T={T} B={B} flags={a.flags}: bucket(us)  panel-wave tiles  spare-wave tiles   TB/s (8 KB tiles)   ring block published by then")
tot = 0
for b in range(64):
    if hp[b] == 0 and hh[b] == 0 and b > 5 and tot > 0 and hp[b:].sum() + hh[b:].sum() == 0: break
    tiles = int(hp[b]) + int(hh[b]); tot += tiles
    kk = int(((pub - t0) / 100.0 < 4 * (b + 1)).sum()) if pub[0] > 0 else -1
    print(f"  {4*b:4d}-{4*b+4:<4d} {int(hp[b]):8d} {int(hh[b]):8d}   {tiles * 8192 / 4e-6 / 1e12:6.2f}   {kk}")
print(f"  total tiles {tot} = {tot * 8192 / 1e9:.3f} GB")
