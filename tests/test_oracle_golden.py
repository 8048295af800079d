"""Pins the C oracle (oracle/semicrf_oracle.c) against golden vectors produced by the
reference itself (tools/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import EDGE_CASES, edge_inputs, grad_weights, load_golden, rel_err, unpack_lists

FP_TOL = 2e-6      # fp32 logsumexp-order differences between torch's vectorised sums and the scalar C loops
GRAD_TOL = 1e-4    # marginals = exp(sum of O(1e2..1e3) terms): one fp32 ulp at 370 is 3e-5 (SURVEY hard part 5)


def scorer_close(S, ref):
    """fp32 dot-product round-off (~1e-6 of sum|q.k|) is multiplied by the length scale |e-b|."""
    T = ref.shape[0]
    ln = np.abs(np.arange(T)[:, None] - np.arange(T)[None, :]).astype(np.float64) + 1.0
    tol = 5e-6 * ln[:, :, None] + 1e-5 * np.abs(ref)
    return bool(np.all(np.abs(S.astype(np.float64) - ref) <= tol))


def grad_tol(logz):
    return max(1e-4, 2e-6 * float(np.max(np.abs(logz))))


def _starts(g, key):
    k = key + "_start"
    return None if k not in g else [int(x) for x in g[k]]


def _check_decodes(oracle, g, score, noise):
    keys = [k[:-6] for k in g if k.startswith("decode_") and k.endswith("_pairs")]
    assert keys
    for key in keys:
        fwd = key.endswith("_fwd")
        want = unpack_lists(g[key + "_pairs"], g[key + "_offsets"])
        got = oracle.viterbi(score, noise, _starts(g, key), forward=fwd)
        assert got == want, key


def _check_fp(oracle, g, score, noise, full_grad=True):
    T, B = score.shape[0], score.shape[2]
    logz, grad, gn, v, q = oracle.forward_backward(score, noise)
    # marginals are exp() of sums of O(|logZ|) fp32 numbers: their noise floor scales with ulp(|logZ|)
    GRAD_TOL = grad_tol(g["fb_logZ"])
    # the f64 instantiation of the same restatement must agree with the reference's fp32 results too
    lz64, grad64, gn64, _, _ = oracle.forward_backward_f64(score, noise)
    assert rel_err(lz64, g["fb_logZ"]) < FP_TOL
    assert rel_err(gn64, g["fb_gradNoise"]) < GRAD_TOL
    assert rel_err(logz, g["fb_logZ"]) < FP_TOL
    assert rel_err(logz, g["logZ_noBackward"]) < 1e-5
    assert rel_err(gn, g["fb_gradNoise"]) < GRAD_TOL
    if full_grad and "fb_grad" in g:
        assert rel_err(grad, g["fb_grad"]) < GRAD_TOL
        assert np.all(np.triu(grad.transpose(2, 0, 1), 1) == 0.0)
    w = grad_weights(T)
    assert rel_err(grad.astype(np.float64).sum(axis=(0, 1)), g["fb_grad_sum"]) < GRAD_TOL
    assert rel_err((grad.astype(np.float64) * w[:, :, None]).sum(axis=(0, 1)), g["fb_grad_wsum"]) < GRAD_TOL
    # v[T-1] == q[0] (reference comment :332-334)
    assert rel_err(v[-1], q[0]) < 1e-5
    if "evalPath" in g:
        iv = unpack_lists(g["intervals_pairs"], g["intervals_offsets"])
        path = oracle.eval_path(iv, score, noise)
        assert rel_err(path, g["evalPath"]) < FP_TOL
        assert rel_err(path - logz, g["logProb"]) < 1e-5


def test_minimal_example(oracle):
    g = load_golden("minimal_T200_B4")
    score, noise = g["score"], g["noise"]
    _check_decodes(oracle, g, score, noise)
    _check_fp(oracle, g, score, noise)
    logz, grad, gn, v, q = oracle.forward_backward(score, noise)
    rows = g["rows"]
    assert rel_err(grad[rows], g["fb_grad_rows"]) < GRAD_TOL
    # d logProb / d score = onehot(path) - marginals
    iv = unpack_lists(g["intervals_pairs"], g["intervals_offsets"])
    d = -grad.copy()
    for c, lst in enumerate(iv):
        for b, e in lst:
            d[e, b, c] += 1.0
    # golden holds d(-sum logProb): sign flip
    assert rel_err(-d[rows], g["dScore_logProb_rows"]) < GRAD_TOL


@pytest.mark.parametrize("case", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_cases(oracle, case):
    name, T, B, kind, seed, tr = case
    g = load_golden("edge_" + name)
    score, noise = edge_inputs(T, B, kind, seed, tr)
    score, noise = score.numpy(), noise.numpy()
    _check_decodes(oracle, g, score, noise)
    _check_fp(oracle, g, score, noise)


@pytest.mark.parametrize("kind", ["randn", "model"])
def test_medium(oracle, kind):
    from transkun_amd import synth
    g = load_golden(f"medium_T256_B90_{kind}")
    T, B, seed = (int(x) for x in g["meta"])
    score, noise = synth.crf_inputs_numpy(T, B, seed, kind)
    _check_decodes(oracle, g, score, noise)
    _check_fp(oracle, g, score, noise, full_grad=False)


@pytest.mark.parametrize("name", ["small", "sqrt", "none", "medium"])
def test_scorer(oracle, name):
    import torch
    from transkun_amd import synth
    g = load_golden("scorer_" + name)
    N, P, T, D = (int(x) for x in g["meta"])
    ls = str(g["ls"])
    W = synth.hash_normal((2 * D + 1) * D, 41, "cpu").view(2 * D + 1, D) * (1.0 / D ** 0.5)
    bvec = synth.hash_normal(2 * D + 1, 42, "cpu") * 0.1
    ctx = synth.hash_normal(N * P * T * D, 43, "cpu").view(N, P, T, D)
    y = torch.nn.functional.linear(ctx, W, bvec)
    q, k, diag = y.split([D, D, 1], dim=-1)
    S = oracle.interval_score(q.reshape(N * P, T, D).numpy(), k.reshape(N * P, T, D).numpy(),
                              diag.reshape(N * P, T).numpy(), ls)
    if "S" in g:
        ref = g["S"].reshape(T, T, N * P)
        assert scorer_close(S, ref)
    w = grad_weights(T)
    Sd = S.astype(np.float64)
    assert rel_err((Sd * np.tril(np.ones((T, T)))[:, :, None]).sum(axis=(0, 1)), g["S_tril_sum"]) < 1e-4
    assert rel_err((Sd * w[:, :, None]).sum(axis=(0, 1)), g["S_wsum"]) < 1e-4
    assert float(g["noise_absmax"]) == 0.0


def test_oploop_port_matches_oracle(oracle):
    """The torch op-loop port (timed as cpu_baseline) computes the same thing as the C oracle."""
    import torch
    from transkun_amd import synth
    score, noise = synth.crf_inputs(48, 5, 3, "cpu", "randn")
    logz, grad, gn = oracle.oploop_forward_backward(score, noise)
    lz, g2, gn2, _, _ = oracle.forward_backward(score.numpy(), noise.numpy())
    assert rel_err(logz.numpy(), lz) < FP_TOL
    assert rel_err(grad.numpy(), g2) < GRAD_TOL
    assert rel_err(gn.numpy(), gn2) < GRAD_TOL
    st = [(c * 5) % 48 for c in range(5)]
    assert oracle.oploop_viterbi_backward(score, noise, st) == oracle.viterbi(score.numpy(), noise.numpy(), st)
    assert oracle.oploop_viterbi_backward(score, noise) == oracle.viterbi(score.numpy(), noise.numpy())


def _attr_case(name):
    """Inputs of an attr_* fixture (regenerated from the integer hash) + the reference's outputs."""
    import torch
    from transkun_amd import synth
    g = load_golden("attr_" + name)
    N, SYM, T, D, seed = (int(x) for x in g["meta"])
    ctx = synth.hash_normal(N * SYM * T * D, 100 + seed, "cpu").view(N, SYM, T, D)
    flat = unpack_lists(g["pairs"], g["offsets"])
    batch = [flat[n * SYM:(n + 1) * SYM] for n in range(N)]
    return g, ctx, flat, batch, (N, SYM, T, D)


def check_attr_outputs(g, a, b, sym, sc):
    """a, b: float64-able arrays [K, D] in the reference's order; compares with what the reference's own
    fetchIntervalFeaturesBatch produced (full tensors for the small case, column sums / weighted sums for the large one)."""
    assert np.array_equal(np.asarray(sym), g["symIdx"]) and np.array_equal(np.asarray(sc), g["scatterIdx"])
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if "ctx_a" in g:
        assert np.array_equal(a.astype(np.float32), g["ctx_a"]) and np.array_equal(b.astype(np.float32), g["ctx_b"])
    else:
        w = (np.arange(a.shape[0], dtype=np.float64) % 7 + 1)[:, None]
        for got, key in ((a.sum(0), "ctx_a_sum"), (b.sum(0), "ctx_b_sum"), ((a * b).sum(0), "ab_sum"),
                         ((a * w).sum(0), "ctx_a_wsum"), ((b * w).sum(0), "ctx_b_wsum")):
            assert rel_err(got, g[key]) < 1e-6, key


@pytest.mark.parametrize("name", ["small", "model"])
def test_attribute_features_oracle(oracle, name):
    """oracle.fetch_interval_features against the reference's TransKun.fetchIntervalFeaturesBatch."""
    g, ctx, flat, batch, _ = _attr_case(name)
    a, b, sym, sc = oracle.fetch_interval_features(ctx.numpy(), batch)
    check_attr_outputs(g, a, b, sym, sc)


@pytest.mark.parametrize("name", ["small", "T691_P90"])
def test_segment_oracle(oracle, name):
    """The oracle's scorer + CRF composite (interval_score -> forward_backward / eval_path / viterbi -> the scorer's
    gradient in f64 numpy -> fetch_interval_features) against the reference's own modules run on the model glue
    (ModelTransformer.py:199-225, :256-266, :537-582): tests/golden/segment_*.npz."""
    import torch
    from segment_common import SEGMENT_CASES, check_segment_features, check_segment_grads, segment_inputs
    g = load_golden("segment_" + name)
    N, P, T, D = SEGMENT_CASES[name][:4]
    C = N * P
    ctx, W, bias, iv, gout, starts = segment_inputs(name)
    y = torch.nn.functional.linear(ctx, W, bias)
    q, k, diag = (t.reshape(C, T, -1).numpy() for t in y.split([D, D, 1], dim=-1))
    S = oracle.interval_score(q, k, diag[..., 0], "linear")
    Sd = S.astype(np.float64)
    assert rel_err((Sd * np.tril(np.ones((T, T)))[:, :, None]).sum(axis=(0, 1)), g["S_tril_sum"]) < 1e-4
    noise = np.zeros((T - 1, C), np.float32)
    logz, marg, _, _, _ = oracle.forward_backward(S, noise)
    path = oracle.eval_path(iv, S, noise)
    assert rel_err(logz, g["logZ"]) < 2e-5
    assert rel_err(path, g["evalPath"]) < 2e-5
    assert rel_err(path - logz, g["logProb"]) < 2e-5
    # decode on the oracle's scores == the reference's decode on the reference's scores
    assert oracle.viterbi(S, noise, starts) == unpack_lists(g["decode_pairs"], g["decode_offsets"])
    # gradient of sum(gout * logProb): dS = gout * (onehot(path) - marginals), pushed through the scorer in f64
    go = gout.numpy().astype(np.float64)
    dS = -marg.astype(np.float64) * go[None, None, :]
    for c, lst in enumerate(iv):
        for b0, e0 in lst:
            dS[e0, b0, c] += go[c]
    t = np.arange(T)
    ln = np.abs(t[:, None] - t[None, :]).astype(np.float64)
    G = np.tril(dS.transpose(2, 0, 1) * ln[None]) / np.sqrt(D)                # [C, e, b]
    dq = np.matmul(G, k.astype(np.float64))
    dk = np.matmul(G.transpose(0, 2, 1), q.astype(np.float64))
    dd = np.diagonal(dS, axis1=0, axis2=1)                                   # [C, T]
    dy = np.concatenate([dq, dk, dd[..., None]], axis=-1).reshape(C * T, 2 * D + 1)
    x = ctx.numpy().astype(np.float64).reshape(C * T, D)
    dctx = (dy @ W.numpy().astype(np.float64)).reshape(N, P, T, D)
    dW = dy.T @ x
    dbias = dy.sum(axis=0)
    check_segment_grads(g, dctx, dW, dbias)
    dec = unpack_lists(g["decode_pairs"], g["decode_offsets"])
    batch = [dec[n * P:(n + 1) * P] for n in range(N)]
    a, b, sym, sc = oracle.fetch_interval_features(ctx.numpy(), batch)
    check_segment_features(g, a, b, sym, sc)


@pytest.mark.parametrize("name", ["small", "real"])
def test_transcribe_loop_oracle(oracle, name):
    """The oracle's restatement of the segment loop (event assembly, lastP, next forced start, incomplete-event merge) and the
    product's host-side merge + resolveOverlapping against the reference's own TransKun.transcribe run on the same decoded
    paths and head outputs (tests/golden/transcribe_*.npz)."""
    import math
    from segment_common import TARGET_PITCH, event_table, golden_events, golden_of_heads, transcribe_inputs
    from transkun_amd.transcribe import EventMerger, Note
    g = load_golden("transcribe_" + name)
    I = transcribe_inputs(name)
    n_seg, P = I["n_seg"], I["P"]
    frameDur = I["hop"] / I["fs"]
    stepFrames = int(I["step"] / I["hop"])
    lastFrameIdx = round(I["seg"] / I["hop"])
    per_segment, merger = [], EventMerger()
    hi = 0                                                   # head outputs exist only for segments with intervals
    for i in range(n_seg):
        lists = unpack_lists(g[f"seg{i}_pairs"], g[f"seg{i}_offsets"])
        beginTime = (i * I["step"]) / I["fs"] - I["pad_t"]
        K = sum(len(x) for x in lists)
        if K == 0:
            per_segment.append([])
            continue
        ofValue, ofPresence, vel = golden_of_heads(g, hi); hi += 1
        ev, lastP, nextStart = oracle.segment_events(lists, P, ofValue.tolist(), ofPresence.tolist(), lastFrameIdx, frameDur, [beginTime],
                                                     stepFrames)
        assert lastP == [int(x) for x in g[f"seg{i}_lastP"]]
        if i + 1 < n_seg:
            assert nextStart == [int(x) for x in g[f"seg{i + 1}_start"]]        # the next segment's forcedStartPos (:789-791)
        seg, n = [], 0
        for c, cur in enumerate(ev):
            for (s, e, on, off) in cur:
                seg.append((s, e, TARGET_PITCH[c % P], int(vel[n]), on, off)); n += 1
        seg.sort(key=lambda x: (x[0], x[1], x[2]))
        per_segment.append(seg)
        merger.add_segment([Note(*x) for x in seg])
    assert event_table(oracle.merge_segments(per_segment)) == golden_events(g, "merged")
    assert event_table(n.astuple() for n in merger.finish(resolve=False)) == golden_events(g, "merged")
    merger2 = EventMerger()
    for seg in per_segment:
        merger2.add_segment([Note(*x) for x in seg])
    assert event_table(n.astuple() for n in merger2.finish(resolve=True)) == golden_events(g, "final")
