"""Gradient sweep (and forward sweep) timing that does not depend on where ONE gradient buffer happens to lie: the sweep is timed
into several dScore allocations (the physical placement of the 1.4 GB the sweep writes moves its time by +-5 %)
and the minimum / median are printed.  GPU box only.   python tools/bench_grad.py 1024x352 691x384 ..."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")

def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

NBUF = int(os.environ.get("NBUF", "4"))
out = []
for a in sys.argv[1:]:
    T, B = (int(x) for x in a.split("x"))
    s, n = synth.crf_inputs(T, B, 1234, dev, "randn")
    f = min(timeit(lambda: nsci._logz_fwd_raw(s, n, True)) for _ in range(2))
    lz, v = nsci._logz_fwd_raw(s, n, True); g = torch.ones(B, device=dev)
    ws = _lib.leased_workspace(_lib.OP_LOGZ_BWD, T, B, dev)
    dn = torch.empty_like(n); q = torch.empty(0, device=dev)
    ts, keep = [], []
    for i in range(NBUF):
        keep.append(torch.empty((37 + 61 * i) << 18, dtype=torch.uint8, device=dev))       # shift the next allocation
        ds = torch.zeros(T, T, B, device=dev)
        ts.append(timeit(lambda: _lib.ops().logz_bwd(s, n, v, lz, g, ds, dn, q, False, nsci.GRAD_UPPER_IS_ZERO, ws)))
        del ds
    ts.sort()
    out.append("%dx%d fwd %.1f grad min %.1f med %.1f max %.1f" % (T, B, f, ts[0], ts[len(ts) // 2], ts[-1]))
    del s, n, lz, v, keep; torch.cuda.empty_cache()
print(" | ".join(out), "| status", _lib.device_status(), flush=True)
