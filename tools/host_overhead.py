"""Per-call host overhead of the boundary: a tiny semicrf_logz_fwd (T=16, NBatch=4: the kernel itself is a few us) issued
2000 times through torch.ops.semicrf (stable-ABI shim, dispatcher + device guard + current stream) and through raw
ctypes on the same pre-allocated buffers.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
dev = torch.device("cuda:0")
T, B = 16, 4
s, n = synth.crf_inputs(T, B, 1, dev)
logz = torch.empty(B, device=dev); v = torch.empty(T, B, device=dev)
ws = _lib.workspace(_lib.OP_LOGZ_FWD, T, B, dev)
ops, lib = _lib.ops(), _lib.load()
N = 2000
def run(fn):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N): fn()
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    return t_issue / N * 1e6, (time.perf_counter() - t) / N * 1e6
a = run(lambda: ops.logz_fwd(s, n, logz, v, True, ws))
st = _lib.stream_of(s)
b = run(lambda: lib.semicrf_logz_fwd(_lib.ptr(s), _lib.ptr(n), T, B, _lib.ptr(logz), _lib.ptr(v), _lib.ptr(ws), ws.numel(), st))
c = run(lambda: lib.semicrf_logz_fwd(_lib.ptr(s), _lib.ptr(n), T, B, _lib.ptr(logz), _lib.ptr(v), _lib.ptr(ws), ws.numel(), _lib.stream_of(s)))
print(f"torch.ops.semicrf.logz_fwd : {a[0]:.2f} us/call to issue, {a[1]:.2f} us/call incl. drain")
print(f"ctypes (stream cached)     : {b[0]:.2f} us/call to issue, {b[1]:.2f} us/call incl. drain")
print(f"ctypes (stream per call)   : {c[0]:.2f} us/call to issue, {c[1]:.2f} us/call incl. drain")
