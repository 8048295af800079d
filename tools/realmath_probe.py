"""Panels alone on REAL u values vs on the fill pattern (probe build): is the panel math value dependent?
One normal launch publishes u into a workspace; further launches with debug flags 12 / 76 / 44 (+512: keep that u) reuse it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SEMICRF_HYBRID_PANEL_WAVES", "0")
import torch
from transkun_amd import _lib, synth
T, B = int(os.environ.get("PT", 1024)), int(os.environ.get("PB", 352))
dev = torch.device("cuda:0")
s, n = synth.crf_inputs(T, B, 1234, dev)
logz = torch.empty(B, device=dev); v = torch.empty(T, B, device=dev)
ws = _lib.workspace(_lib.OP_LOGZ_FWD, T, B, dev)
def run(flags, reps=10):
    os.environ["SEMICRF_DEBUG_FLAGS"] = str(flags)
    for _ in range(2): _lib.ops().logz_fwd(s, n, logz, v, True, ws)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): _lib.ops().logz_fwd(s, n, logz, v, True, ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("full sweep: %.1f us" % run(0))
for base, name in ((12, "panels alone, math + u"), (76, "panels alone, math, no u traffic"), (44, "panels alone, stream only")):
    run(0, 1)                                   # publish real u into ws
    real = run(base | 512)
    fillp = run(base)
    print(f"{name:36s}: real u {real:6.1f} us   fill pattern {fillp:6.1f} us")
