"""Shared by the CPU (oracle) and GPU tests of the segment-shaped goldens (tests/golden/segment_*.npz, written by
tools/make_golden.py:case_segment from the reference's scorer + CRF + fetchIntervalFeaturesBatch)."""
import numpy as np

SEGMENT_CASES = {
    # name: (N, P, T, D, ctx scale, weight scale, seed)      -- must match tools/make_golden.py SEGMENTS
    "small": (2, 5, 48, 32, 0.5, 0.3, 3),
    "T691_P90": (1, 90, 691, 256, 1.0, 1.0, 5),
    "T691_N4": (4, 90, 691, 256, 0.5, 0.3, 6),
}


def ctx_weights(T, D):
    t = np.arange(T)[:, None]
    d = np.arange(D)[None, :]
    return (((t * 31 + d * 17) % 64).astype(np.float64) / 64.0)


def segment_inputs(name, device="cpu"):
    """Same construction as tools/make_golden.py:segment_inputs (inputs are not stored in the fixtures)."""
    from transkun_amd import synth
    N, P, T, D, cscale, wscale, seed = SEGMENT_CASES[name]
    ctx = synth.hash_normal(N * P * T * D, 200 + seed, device).view(N, P, T, D) * cscale
    W = synth.hash_normal((2 * D + 1) * D, 300 + seed, device).view(2 * D + 1, D) * (wscale / D ** 0.5)
    bias = synth.hash_normal(2 * D + 1, 400 + seed, device) * 0.1
    iv = synth.synthetic_intervals(T, N * P, seed=seed)
    gout = synth.hash_normal(N * P, 500 + seed, device)
    starts = [(c * 29 + 3) % (T // 2) for c in range(N * P)]
    return ctx, W, bias, iv, gout, starts


def check_segment_grads(g, dctx, dW, dbias, tol=2e-3):
    """dctx [N,P,T,D], dW [2D+1,D], dbias [2D+1] (numpy) against the reference's digests.  Sums are compared relative to
    the sum of absolute values the reference saw (cancellation), rows relative to their largest entry."""
    N, P, T, D = (int(x) for x in g["meta"][:4])
    d64 = np.asarray(dctx, np.float64)
    w = ctx_weights(T, D)
    scale = g["dctx_abs_sum"] + 1e-30
    e1 = np.max(np.abs(d64.sum(axis=(2, 3)) - g["dctx_sum"]) / scale)
    e2 = np.max(np.abs((d64 * w[None, None]).sum(axis=(2, 3)) - g["dctx_wsum"]) / scale)
    e3 = np.max(np.abs(np.abs(d64).sum(axis=(2, 3)) - g["dctx_abs_sum"]) / scale)
    rows = np.asarray(dctx)[0][np.ix_(g["psel"], g["tsel"])]
    e4 = np.max(np.abs(rows - g["dctx_rows"])) / (np.max(np.abs(g["dctx_rows"])) + 1e-30)
    dWs = np.asarray(dW)
    e5 = np.max(np.abs(dWs[[0, 1, D - 1, D, 2 * D - 1, 2 * D]] - g["dW_rows"])) / (np.max(np.abs(g["dW_rows"])) + 1e-30)
    e6 = np.max(np.abs(dWs.astype(np.float64).sum(axis=1) - g["dW_rowsum"])) / (np.max(np.abs(g["dW_rowsum"])) + 1e-30)
    e7 = np.max(np.abs(np.asarray(dbias) - g["dbias"])) / (np.max(np.abs(g["dbias"])) + 1e-30)
    errs = {"dctx_sum": e1, "dctx_wsum": e2, "dctx_abs_sum": e3, "dctx_rows": e4, "dW_rows": e5, "dW_rowsum": e6, "dbias": e7}
    bad = {k: float(v) for k, v in errs.items() if not v < tol}
    assert not bad, bad
    return errs


def check_segment_features(g, a, b, sym, sc):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert np.array_equal(np.asarray(sym).astype(np.int32), g["attr_symIdx"])
    assert np.array_equal(np.asarray(sc).astype(np.int32), g["attr_scatterIdx"])
    w = (np.arange(a.shape[0], dtype=np.float64) % 7 + 1)[:, None]
    for got, key in ((a.sum(0), "attr_a_sum"), (b.sum(0), "attr_b_sum"), ((a * b).sum(0), "attr_ab_sum"),
                     ((a * w).sum(0), "attr_a_wsum"), ((b * w).sum(0), "attr_b_wsum")):
        err = np.max(np.abs(got - g[key])) / (np.max(np.abs(g[key])) + 1e-30)
        assert err < 1e-6, (key, err)


TRANSCRIBE_CASES = {
    # name: (D, hidden, fs, hop, window, segment s, step s, audio s, ctx scale, weight scale, seed)   -- tools/make_golden.py TRANSCRIBE
    "small": (32, 48, 44100, 1024, 4096, 1.2, 0.6, 1.9, 0.5, 0.3, 11),
    "real": (256, 512, 44100, 1024, 4096, 16.0, 8.0, 37.0, 0.5, 0.3, 12),
}
TARGET_PITCH = [-64, -67] + list(range(21, 108 + 1))


def transcribe_inputs(name, device="cpu"):
    """Same construction as tools/make_golden.py:transcribe_inputs."""
    import math
    from transkun_amd import synth
    D, H, fs, hop, win, seg_s, step_s, audio_s, cscale, wscale, seed = TRANSCRIBE_CASES[name]
    P = 90
    pad_t = seg_s - step_s
    n_sample = int(audio_s * fs) + 2 * math.ceil(pad_t * fs)
    step = math.ceil(step_s * fs / hop) * hop
    seg = math.ceil(seg_s * fs)
    n_seg = len(range(0, n_sample, step))
    T = math.ceil(seg / hop) + 1
    ctxs = [synth.hash_normal(P * T * D, 700 + 10 * seed + i, device).view(1, P, T, D) * cscale for i in range(n_seg)]
    W = synth.hash_normal((2 * D + 1) * D, 800 + seed, device).view(2 * D + 1, D) * (wscale / D ** 0.5)
    bias = synth.hash_normal(2 * D + 1, 810 + seed, device) * 0.1
    heads = {}
    for nm, nout, sd in (("velocity", 128, 820), ("of", 4, 830)):
        heads[nm] = (synth.hash_normal(H * 3 * D, sd + seed, device).view(H, 3 * D) * (1.0 / (3 * D) ** 0.5),
                     synth.hash_normal(H, sd + 1 + seed, device) * 0.1,
                     synth.hash_normal(nout * H, sd + 2 + seed, device).view(nout, H) * (1.0 / H ** 0.5),
                     synth.hash_normal(nout, sd + 3 + seed, device) * 0.1)
    return dict(D=D, H=H, fs=fs, hop=hop, win=win, seg_s=seg_s, step_s=step_s, audio_s=audio_s, P=P, T=T, n_seg=n_seg,
                ctxs=ctxs, W=W, bias=bias, heads=heads, n_sample_unpadded=int(audio_s * fs), step=step, seg=seg, pad_t=pad_t)


def golden_of_heads(g, i):
    """(ofValue [K,2] float32, ofPresence [K,2] bool, velocity [K]) of segment-with-intervals number i, from the reference's raw head
    outputs, by the reference's own expressions (ModelTransformer.py:646-655) on the CPU."""
    import torch
    raw = torch.from_numpy(g[f"head{i}_of"])
    ofValue, ofPresence = raw.chunk(2, dim=-1)
    ofDist = torch.distributions.ContinuousBernoulli(logits=ofValue)
    ofValue = torch.clamp((ofDist.mean - 0.5) / 0.99, -0.5, 0.5)
    return ofValue.contiguous(), (ofPresence > 0).contiguous(), g[f"head{i}_velocity_argmax"]


def event_table(events):
    """sorted tuples (start, end, pitch, velocity, hasOnset, hasOffset) for comparisons"""
    return sorted((float(e[0]), float(e[1]), int(e[2]), int(e[3]), bool(e[4]), bool(e[5])) for e in events)


def golden_events(g, key):
    t, pv, fl = g[key + "_times"], g[key + "_pitch_velocity"], g[key + "_flags"]
    return sorted((float(t[i, 0]), float(t[i, 1]), int(pv[i, 0]), int(pv[i, 1]), bool(fl[i, 0]), bool(fl[i, 1])) for i in range(len(t)))
