"""Soak / determinism check: repeated sweeps must give bit-identical results (every reduction order is fixed by the
task decomposition, not by timing) and never raise the device status word.  GPU box only."""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")
def dig(t): return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
bad = 0
for (T, B, n) in ((1024, 352, 60), (691, 90, 150), (691, 360, 60), (2048, 352, 10), (333, 46, 300), (130, 600, 60), (200, 33, 200), (512, 88, 100), (1024, 351, 20),
                  (1024, 88, 60), (2048, 88, 15), (1100, 190, 30)):        # two far waves per spine (forward / decode of few chains, T >= 1024)
    s, nz = synth.crf_inputs(T, B, 77, dev)
    g = synth.hash_normal(B, 5, dev)
    ref = None
    for it in range(n):
        lz, v = nsci._logz_fwd_raw(s, nz, True)
        ds, dn, q = nsci._logz_bwd_raw(s, nz, v, lz, g, True)
        pairs, offs = nsci._viterbi_raw(s, nz, None, False)
        tot = int(offs[-1])
        cur = (dig(lz), dig(v), dig(ds), dig(dn), dig(q), dig(offs), dig(pairs[:tot]))
        if ref is None: ref = cur
        elif cur != ref:
            bad += 1; print("MISMATCH", T, B, it, [a == b for a, b in zip(cur, ref)], flush=True)
    st = _lib.device_status()
    print(f"T={T} B={B}: {n} repeats, digests {'stable' if bad == 0 else 'UNSTABLE'}, device status {st}", flush=True)
    if st: bad += 1
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
