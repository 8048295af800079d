"""Where the host side of NeuralSemiCRFInterval.decode goes (T=2048 x 352, model-like and randn scores): device time, the two copies,
the list building, per call.  GPU box only."""
import gc, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transkun_amd import CRF, _lib, synth
nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
dev = torch.device("cuda:0")
T, B = 2048, 352
start = [4] * B
for kind in ("model", "randn"):
    s, n = synth.crf_inputs(T, B, 1234, dev, kind)
    crf = CRF.NeuralSemiCRFInterval(s, n)
    crf.decode(forcedStartPos=start); gc.collect(); torch.cuda.synchronize()
    def t(fn, reps=5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
    ms_all, res = t(lambda: crf.decode(forcedStartPos=start))
    ms_pk, (ph, oh) = t(lambda: crf.decode_packed(forcedStartPos=start))
    pt, ot = torch.from_numpy(ph), torch.from_numpy(oh)
    ms_un, _ = t(lambda: nsci.unpack_intervals(pt, ot, T))
    mm = _lib.marshal()
    def raw():
        return mm.unpack(pt.data_ptr(), ot.data_ptr(), B, T)
    ms_raw, _ = t(raw)
    st = np.asarray(start, dtype=np.int64)
    ms_st, _ = t(lambda: torch.from_numpy(st.astype(np.int32)).to(dev, non_blocking=True), 20)
    print(f"{kind}: {len(ph)} intervals; decode() {ms_all:.3f} ms, decode_packed() {ms_pk:.3f}, unpack_intervals alone {ms_un:.3f}, marshal.unpack alone {ms_raw:.3f}, "
          f"start upload {ms_st:.3f}; gc counts {gc.get_count()} thresholds {gc.get_threshold()} objects {len(gc.get_objects())}", flush=True)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): crf.decode(forcedStartPos=start)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
