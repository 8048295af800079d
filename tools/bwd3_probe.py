"""interval_score_bwd_ws: exact fp32 against the three-limb bf16 kernels (length_scaling | 16), event-timed; run under
rocprofv3 --kernel-trace --stats for the per-kernel split (pack / dq / dk)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transkun_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
shapes = ((352, 1024, 256), (360, 691, 256), (90, 691, 256), (88, 2048, 256), (352, 1024, 128))
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (C, T, D) in shapes:
    dS = torch.randn(T, T, C, device=dev); q = torch.randn(C, T, D, device=dev); k = torch.randn(C, T, D, device=dev)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dd = torch.empty(C, T, device=dev)
    nws = int(lib.interval_score_bwd_workspace_bytes(C, T, D))
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    res = {}
    for name, mode in (("fp32", 0), ("bf16x3", 16)):
        f = lambda: _lib.check(lib.interval_score_bwd_ws(_lib.ptr(dS), _lib.ptr(q), _lib.ptr(k), C, T, D, D, D, 1.0 / 16, mode, _lib.ptr(dq),
                                                         _lib.ptr(dk), _lib.ptr(dd), D, D, 1, _lib.ptr(ws), nws, _lib.stream_of(dS)), "bwd_ws")
        # (WARM / ITERS: the shader clock climbs for tens of milliseconds under load -- 2 + 5 calls measure the ramp, 20 + 50 the plateau)
        nwarm, niter = int(os.environ.get("WARM", "2")), int(os.environ.get("ITERS", "5"))
        for _ in range(nwarm): f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(niter): f()
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / niter, dq.clone(), dk.clone())
    fl = 2 * 2.0 * C * (T * (T + 1) / 2) * D
    a, b = res["fp32"], res["bf16x3"]
    print(f"C={C} T={T} D={D}: fp32 {a[0]:.3f} ms ({fl/a[0]/1e9:.1f} TF), bf16x3 {b[0]:.3f} ms ({fl/b[0]/1e9:.1f} TF fp32-eq); "
          f"max |d dq| {float((a[1]-b[1]).abs().max()):.3e} of {float(a[1].abs().max()):.3e}, |d dk| {float((a[2]-b[2]).abs().max()):.3e} of {float(a[2].abs().max()):.3e}", flush=True)

if "--probe" in sys.argv:
    # a library built with -DSEMICRF_G3_PROBE=1 (SEMICRF_LIB): cycle counters of every wave of the dq kernel behind the row sums
    import numpy as np
    C, T, D = 352, 1024, 256
    dS = torch.randn(T, T, C, device=dev); q = torch.randn(C, T, D, device=dev); k = torch.randn(C, T, D, device=dev)
    dq = torch.zeros_like(q); dk = torch.zeros_like(k); dd = torch.empty(C, T, device=dev)
    dc = torch.zeros(C * T + 256 * 64, device=dev)
    nws = int(lib.interval_score_bwd_workspace_bytes(C, T, D)); ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    for _ in range(int(os.environ.get('PROBE_CALLS', '2'))):
        _lib.check(lib.interval_score_bwd_ws_pc(_lib.ptr(dS), _lib.ptr(q), _lib.ptr(k), C, T, D, D, D, 1.0 / 16, 16, C, C, _lib.ptr(dq), _lib.ptr(dk),
                                                _lib.ptr(dd), _lib.ptr(dc), D, D, 1, 1, _lib.ptr(ws), nws, _lib.stream_of(dS)), "bwd_ws_pc")
    torch.cuda.synchronize()
    names = ["multiply", "epilogue", "barrier1", "wait loads", "split+stores", "requests", "barrier2", "-"]
    pc = dc[C * T:C * T + 256 * 64].view(256, 8, 8).cpu().numpy().astype(np.float64)
    tot = pc[:, :, :7].sum(axis=2)
    rt = pc[:, :, 7]
    print("dq cycles per wave: mean %.0f min %.0f max %.0f; wall (100 MHz ticks) mean %.0f = %.1f us -> clock %.2f GHz" % (
        tot.mean(), tot.min(), tot.max(), rt.mean(), rt.mean() / 100.0, tot.mean() / (rt.mean() * 10.0)))
    for g, sl in (("group 0", slice(0, 4)), ("group 1", slice(4, 8))):
        m = pc[:, sl, :7].mean(axis=(0, 1))
        print("  ", g, " ".join("%s %.1f%% (%.0f)" % (names[i], 100 * m[i] / m.sum(), m[i]) for i in range(7)))

if "--proj" in sys.argv:
    # the projection's NN forms at the training shape: exact fp32 against the three-limb kernels
    from transkun_amd.scorer import QPAD, proj_forward, proj_input_grad
    M, K = 4 * 90 * 691, 256
    x = torch.randn(M, K, device=dev); W = torch.randn(K + QPAD, K, device=dev) / 16; W[K + 2:] = 0; b = torch.randn(K + QPAD, device=dev); b[K + 2:] = 0
    dy = torch.randn(M, K + QPAD, device=dev); dy[:, K + 2:] = 0
    acc = torch.zeros(M, K, device=dev)
    def t(f, n=5):
        for _ in range(2): f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    fl = 2.0 * M * K * (K + 2)
    for name, p in (("fp32", 0), ("bf16x3", 1)):
        from transkun_amd.scorer import proj_weight_grad
        tf = t(lambda: proj_forward(x, W, b, K, prec=p)); tb = t(lambda: proj_input_grad(dy, W, out=acc, prec=p))
        tw = t(lambda: proj_weight_grad(dy, x, K, prec=p))
        print(f"projection M={M} K={K} {name}: forward {tf:.3f} ms ({fl / tf / 1e9:.1f} TF), input gradient (accumulating) {tb:.3f} ms ({fl / tb / 1e9:.1f} TF), "
              f"weight gradient (GEMM + extras + reduction) {tw:.3f} ms", flush=True)

if "--proj-probe" in sys.argv:
    # a library built with -DSEMICRF_P3_PROBE=1: cycle counters of proj_gemm3_kernel's waves over the first rows of the output
    import numpy as np
    from transkun_amd.scorer import proj_input_grad
    M, K = 4 * 90 * 691, 256
    dy = torch.randn(M, K, device=dev); W = torch.randn(K, K, device=dev) / 16
    for _ in range(2):
        out = proj_input_grad(dy, W, prec=1)
    torch.cuda.synchronize()
    pc = out.flatten()[:256 * 64].view(256, 8, 8).cpu().numpy().astype(np.float64)
    names = ["multiply", "epilogue", "barrier1", "wait loads", "split+stores", "requests", "barrier2"]
    tot = pc[:, :, :7].sum(axis=2); rt = pc[:, :, 7]
    print("proj cycles per wave: mean %.0f; wall %.1f us -> clock %.2f GHz; chunks per workgroup ~%.1f" % (tot.mean(), rt.mean() / 100.0, tot.mean() / (rt.mean() * 10.0), M / 128 / 256 * 8))
    for g, sl in (("group 0", slice(0, 4)), ("group 1", slice(4, 8))):
        m = pc[:, sl, :7].mean(axis=(0, 1))
        print("  ", g, " ".join("%s %.1f%% (%.0f)" % (names[i], 100 * m[i] / m.sum(), m[i]) for i in range(7)))
