"""Effective shader clock of a panel workgroup over one forward sweep (probe build, debug flag 128): cycle counter vs the
100 MHz real-time counter, for the full kernel and for the streaming-only / no-math ablations."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["SEMICRF_DEBUG_KEEP_WS"] = "1"
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
T, B = 1024, 352
dev = torch.device("cuda:0")
s, n = synth.crf_inputs(T, B, 1234, dev)
CT = 16 * 256 * 4
for flags in (128, 128 + 12, 128 + 12 + 64, 128 + 44):
    os.environ["SEMICRF_DEBUG_FLAGS"] = str(flags)
    for _ in range(3): nsci._logz_fwd_raw(s, n, True)
    torch.cuda.synchronize()
    ts = nsci._DEBUG_WS[0][CT:CT + T * 16].view(torch.int64).cpu().numpy()
    cyc, rt = ts[602] - ts[600], (ts[603] - ts[601]) / 100.0
    print(f"flags={flags}: panel workgroup alive {rt:.1f} us, {cyc} cycles -> {cyc / rt / 1e3:.2f} GHz")
