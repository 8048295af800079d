"""Cell-stream cache policy (csrc/persist.hip: cell_policy_nt): the sweeps as a training / transcription step runs them -- forward
sweep, gradient sweep (pooled buffer) and decode one after the other on the SAME score tensor -- optionally with 1 GB of unrelated
traffic in front of every step (THRASH=1: what the projection GEMMs of a real step do to the memory-side cache).  GPU box only.
   python tools/nt_probe.py 691x90 1024x88 ..."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")
thrash = os.environ.get("THRASH", "0") == "1"
junk = torch.empty(256 << 20, dtype=torch.float32, device=dev) if thrash else None     # 1 GiB

def ev():
    return torch.cuda.Event(enable_timing=True)

rows = []
for a in sys.argv[1:]:
    T, B = (int(x) for x in a.split("x"))
    s, n = synth.crf_inputs(T, B, 1234, dev, "randn")
    g = torch.ones(B, device=dev)
    lz, v = nsci._logz_fwd_raw(s, n, True)
    acc = [0.0, 0.0, 0.0]
    N = 8
    for it in range(N + 2):
        if thrash:
            junk.add_(1.0)
        e = [ev() for _ in range(4)]
        e[0].record(); lz, v = nsci._logz_fwd_raw(s, n, True)
        e[1].record(); nsci._logz_bwd_raw(s, n, v, lz, g)
        e[2].record(); nsci._viterbi_raw(s, n, None, False)
        e[3].record(); torch.cuda.synchronize()
        if it >= 2:
            for i in range(3): acc[i] += e[i].elapsed_time(e[i + 1]) * 1e3 / N
    mb = 4 * B * (T * (T + 1) // 2) / 1e6
    rows.append(f"| {T} | {B} | {mb:.0f} | {acc[0]:.1f} | {acc[1]:.1f} | {acc[2]:.1f} |")
    del s, n, lz, v; torch.cuda.empty_cache()
print("| T | NBatch | lower triangle MB | fwd us | grad sweep us | decode us |" + (" (1 GiB of other traffic before every step)" if thrash else ""))
print("|---|---|---|---|---|---|")
print("\n".join(rows)); print("status", _lib.device_status(), flush=True)
