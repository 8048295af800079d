import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")
def run(T, B):
    s, n = synth.crf_inputs(T, B, 5 + T + B, dev)
    g = synth.hash_normal(B, 3, dev)
    lz, v = nsci._logz_fwd_raw(s, n, True)
    ds, dn, q = nsci._logz_bwd_raw(s, n, v, lz, g, True)
    pairs, offs = nsci._viterbi_raw(s, n, None, False)
    return lz, v, ds, dn, q, offs, pairs[:int(offs[-1])]
shapes = [(48, 7), (130, 45), (200, 33), (333, 91), (691, 89), (1024, 351), (70, 1101), (64, 3), (300, 5)]
ok = True
for T, B in shapes:
    os.environ.pop("SEMICRF_ODD_NATIVE", None)
    ref = run(T, B)
    os.environ["SEMICRF_ODD_NATIVE"] = "1"
    got = run(T, B)
    same = [torch.equal(a, b) for a, b in zip(ref, got)]
    close = [float((a.float() - b.float()).abs().max()) if a.numel() else 0.0 for a, b in zip(ref, got)]
    print(T, B, same, ["%.2e" % c for c in close], "status", _lib.device_status(), flush=True)
    ok = ok and all(same[5:]) and max(close[:5]) < 1e-2
def t(fn, n=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for T, B in [(1024, 351), (691, 89)]:
    s, n = synth.crf_inputs(T, B, 5, dev)
    os.environ.pop("SEMICRF_ODD_NATIVE", None); a = t(lambda: nsci._logz_fwd_raw(s, n, True))
    os.environ["SEMICRF_ODD_NATIVE"] = "1"; b = t(lambda: nsci._logz_fwd_raw(s, n, True))
    s2, n2 = synth.crf_inputs(T, B + 1, 5, dev); c = t(lambda: nsci._logz_fwd_raw(s2, n2, True))
    print(f"T={T} B={B}: ghost chain {a:.1f} us, native odd {b:.1f} us, even B+1 {c:.1f} us")
print("ODD OK" if ok else "ODD MISMATCH")
