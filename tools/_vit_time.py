import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")
T, B = int(sys.argv[1]), int(sys.argv[2])
s, n = synth.crf_inputs(T, B, 1234, dev)
def t(fn, k=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / k * 1e3
print(f"T={T} B={B} HS={os.environ.get('SEMICRF_HYBRID_START','default')}: viterbi (sweep + backtrack + pack) {t(lambda: nsci._viterbi_raw(s, n, None, False)):.1f} us")
