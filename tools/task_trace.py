"""Task timeline of ONE panel wave (build with SEMICRF_PANEL_PROBES=1 SEMICRF_PROBE_TASKS=1, debug flag 256): per task, when it was drawn, when its first tile had landed,
when its last tile was done and when its partial was stored.  --flags adds other debug flags (12 = panels alone, 44 = stream only)."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--B", type=int, default=352)
ap.add_argument("--flags", type=int, default=0)
a = ap.parse_args()
os.environ["SEMICRF_DEBUG_FLAGS"] = str(a.flags | 256); os.environ["SEMICRF_DEBUG_KEEP_WS"] = "1"
import numpy as np, torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
T, B = a.T, a.B
dev = torch.device("cuda:0")
s, n = synth.crf_inputs(T, B, 1234, dev)
for _ in range(3): nsci._logz_fwd_raw(s, n, True)
torch.cuda.synchronize()
ws = nsci._DEBUG_WS[0]
CT = 16 * 256 * 4
raw = ws[CT:CT + 2 * T * 8].view(torch.int64).cpu().numpy()
rec = raw[(3 * T) // 2:].astype(np.int64)
rows = []
for i in range(len(rec) // 5):
    r = rec[5 * i:5 * i + 5]
    if r[0] <= 0 or r[3] <= 0 or r[0] == -1: break
    rows.append(r)
if not rows:
    print("no trace (probe build? SEMICRF_LIB)"); sys.exit(0)
t0 = rows[0][0]
print(f"T={T} B={B} flags={a.flags}: task: k tiles frontier part | drawn  +first  +stream  +reduce | gap to next draw   (us)")
tot = dict(fill=0.0, stream=0.0, red=0.0, gap=0.0, tiles=0)
for i, r in enumerate(rows):
    us = lambda x: x / 100.0
    tiles, k, fr, ea = int(r[4] & 255), int((r[4] >> 8) & 255), int((r[4] >> 16) & 1), int((r[4] >> 20) & 15)
    gap = us(rows[i + 1][0] - r[3]) if i + 1 < len(rows) else 0.0
    tot["fill"] += us(r[1] - r[0]); tot["stream"] += us(r[2] - r[1]); tot["red"] += us(r[3] - r[2]); tot["gap"] += gap; tot["tiles"] += tiles
    if i < 40 or i + 3 > len(rows):
        print(f"{i:3d}: {k:3d} {tiles:3d} {fr} {ea} | {us(r[0]-t0):8.2f} {us(r[1]-r[0]):6.2f} {us(r[2]-r[1]):7.2f} {us(r[3]-r[2]):6.2f} | {gap:6.2f}")
print(f"  {len(rows)} tasks, {tot['tiles']} tiles: fill {tot['fill']:.1f}  stream {tot['stream']:.1f} ({tot['stream']/max(tot['tiles'],1):.2f}/tile)  "
      f"reduce {tot['red']:.1f}  gaps {tot['gap']:.1f}  span {(rows[-1][3]-t0)/100.0:.1f} us")
