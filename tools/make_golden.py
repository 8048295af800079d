#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (torch CPU) in the build container.

Run here only (needs /root/reference): `PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--large]`.
The fixtures hold data only: inputs (when small and not regenerable) and the reference's
outputs.  Inputs of all but the crfMinimalExample case come from transkun_amd.synth (exact
integer hash, same bits everywhere) and are NOT stored.

Decoded interval lists are stored packed: pairs int32 [K,2] + offsets int64 [B+1].
Large cases store digests/checksums only.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/transkun")
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import CRF as REF  # noqa: E402  (the reference package, /root/reference/transkun/CRF)
from CRF import NeuralSemiCRFInterval as RM  # noqa: E402,F401
import importlib  # noqa: E402

REFMOD = importlib.import_module("CRF.NeuralSemiCRFInterval")

from transkun_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def pack(lists):
    counts = [len(x) for x in lists]
    off = np.zeros(len(lists) + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    pairs = np.asarray([p for l in lists for p in l], dtype=np.int32).reshape(-1, 2)
    return pairs, off


def digest(lists):
    pairs, off = pack(lists)
    h = hashlib.sha256()
    h.update(off.astype("<i8").tobytes())
    h.update(pairs.astype("<i4").tobytes())
    return h.hexdigest()


def grad_weights(T):
    e = np.arange(T)[:, None]
    b = np.arange(T)[None, :]
    return (((e * 31 + b * 17) % 64).astype(np.float32) / 64.0)


def run_case(score, noise, intervals, starts, want_grad="full"):
    """Run the reference on one (score, noise) and collect everything."""
    out = {}
    s = score.clone().requires_grad_()
    n = noise.clone().requires_grad_()
    crf = REF.NeuralSemiCRFInterval(s, n)
    T, _, B = score.shape
    if intervals is not None:
        path = crf.evalPath(intervals)
        logz = crf.computeLogZ()
        lp = path - logz
        out["evalPath"] = path.detach().numpy()
        out["logZ"] = logz.detach().numpy()
        out["logProb"] = lp.detach().numpy()
        (-lp.sum()).backward()
        out["dNoise_logProb"] = n.grad.numpy().copy()
        g = s.grad.numpy()
        if want_grad == "full":
            out["dScore_logProb"] = g.copy()
        w = grad_weights(T)
        out["dScore_logProb_sum"] = g.sum(axis=(0, 1)).astype(np.float64)
        out["dScore_logProb_wsum"] = (g.astype(np.float64) * w[:, :, None]).sum(axis=(0, 1))
        out["dScore_upper_absmax"] = np.float64(np.abs(np.triu(g.transpose(2, 0, 1), 1)).max()) if T > 1 else np.float64(0)
    with torch.no_grad():
        logz, grad, gn = REFMOD.forward_backward(score, noise)
        out["fb_logZ"] = logz.numpy()
        out["fb_gradNoise"] = gn.numpy()
        g = grad.numpy()
        if want_grad == "full":
            out["fb_grad"] = g.copy()
        w = grad_weights(T)
        out["fb_grad_sum"] = g.astype(np.float64).sum(axis=(0, 1))
        out["fb_grad_wsum"] = (g.astype(np.float64) * w[:, :, None]).sum(axis=(0, 1))
        out["logZ_noBackward"] = REFMOD.computeLogZ(score, noise).numpy()
        if T > 1:
            for name, st in starts.items():
                for fwd in (False, True):
                    key = f"decode_{name}_{'fwd' if fwd else 'bwd'}"
                    if fwd and st is not None:
                        stf = [T - 1 - x for x in st]      # forward variant: start = END position
                    else:
                        stf = st
                    res = crf.decode(forcedStartPos=stf, forward=fwd)
                    p, o = pack(res)
                    out[key + "_pairs"] = p
                    out[key + "_offsets"] = o
                    if st is not None:
                        out[key + "_start"] = np.asarray(stf, np.int32)
    return out


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def case_minimal():
    """crfMinimalExample.py on CPU: T=200, NBatch=4 (BASELINE.json configs[0])."""
    torch.manual_seed(1234)
    T, B = 200, 4
    score = torch.randn(T, T, B)
    noise = torch.randn(T - 1, B)
    intervals = [[(0, 2), (4, 6), (6, 6), (7, 8)], [(1, 2), (3, 5), (19, 19)], [(0, 0), (4, 7)], []]
    d = run_case(score, noise, intervals, {"none": None, "four": [4] * B})
    # keep the file small: inputs fp32 (640 KB), full grads dropped in favour of a few rows
    g = d.pop("dScore_logProb"); fb = d.pop("fb_grad")
    rows = np.asarray([0, 1, 2, 7, 19, 100, 198, 199])
    d["rows"] = rows
    d["dScore_logProb_rows"] = g[rows]
    d["fb_grad_rows"] = fb[rows]
    d["score"] = score.numpy(); d["noise"] = noise.numpy()
    ip, io = pack(intervals)
    d["intervals_pairs"] = ip; d["intervals_offsets"] = io
    save("minimal_T200_B4", d)


EDGE = [
    # name, T, B, kind, seed, transform
    ("T2_B3", 2, 3, "randn", 11, None),
    ("T3_B5", 3, 5, "randn", 12, None),
    ("T24_B1", 24, 1, "randn", 13, None),
    ("T24_B63", 24, 63, "randn", 14, None),
    ("T24_B65", 24, 65, "randn", 15, None),
    ("T40_B90", 40, 90, "randn", 16, None),
    ("T33_B7_negdiag", 33, 7, "randn", 17, "negdiag"),
    ("T33_B7_posdiag", 33, 7, "randn", 18, "posdiag"),
    ("T33_B7_noise0", 33, 7, "randn", 19, "noise0"),
    ("T48_B9_ties", 48, 9, "ties", 20, None),
    ("T48_B6_huge", 48, 6, "model", 21, "huge"),
    ("T70_B20_model", 70, 20, "model", 22, None),
]


def edge_inputs(T, B, kind, seed, tr):
    score, noise = synth.crf_inputs(T, B, seed, "cpu", kind)
    score = score.clone(); noise = noise.clone()
    d = torch.diagonal(score, dim1=0, dim2=1)
    if tr == "negdiag":
        d.copy_(-d.abs() - 0.125)
    elif tr == "posdiag":
        d.copy_(d.abs() + 0.125)
    elif tr == "noise0":
        noise.zero_()
    elif tr == "huge":
        dd = d.clone()
        score.mul_(4000.0)          # off-diagonal up to ~ +-1e3*|e-b| (SURVEY hard part 6)
        d.copy_(dd)
    return score.contiguous(), noise.contiguous()


def edge_starts(T, B):
    mixed = [(c * 7) % T for c in range(B)]
    return {"none": None, "zero": [0] * B, "Tm2": [max(T - 2, 0)] * B, "Tm1": [T - 1] * B, "mixed": mixed}


def case_edges():
    for name, T, B, kind, seed, tr in EDGE:
        score, noise = edge_inputs(T, B, kind, seed, tr)
        intervals = synth.synthetic_intervals(T, B, seed=seed, every=5, active_every=2)
        d = run_case(score, noise, intervals, edge_starts(T, B))
        ip, io = pack(intervals)
        d["intervals_pairs"] = ip; d["intervals_offsets"] = io
        d["meta"] = np.asarray([T, B, seed])
        save("edge_" + name, d)


def case_medium():
    for kind in ("randn", "model"):
        T, B, seed = 256, 90, 31
        score, noise = synth.crf_inputs(T, B, seed, "cpu", kind)
        intervals = synth.synthetic_intervals(T, B, seed=seed)
        d = run_case(score, noise, intervals, {"none": None, "four": [4] * B, "mixed": [(c * 13) % T for c in range(B)]},
                     want_grad="sums")
        ip, io = pack(intervals)
        d["intervals_pairs"] = ip; d["intervals_offsets"] = io
        d["meta"] = np.asarray([T, B, seed])
        save(f"medium_T{T}_B{B}_{kind}", d)


def case_scorer():
    from transkun.LayersTransformer import ScaledInnerProductIntervalScorer
    for name, N, P, T, D, ls in (("small", 1, 3, 40, 16, "linear"), ("sqrt", 2, 2, 24, 32, "sqrt"),
                                 ("none", 1, 4, 24, 8, "none"), ("medium", 2, 5, 96, 256, "linear")):
        torch.manual_seed(5)
        m = ScaledInnerProductIntervalScorer(D, 1, lengthScaling=ls)
        W = synth.hash_normal((2 * D + 1) * D, 41, "cpu").view(2 * D + 1, D) * (1.0 / D ** 0.5)
        bvec = synth.hash_normal(2 * D + 1, 42, "cpu") * 0.1
        with torch.no_grad():
            m.map[0].weight.copy_(W); m.map[0].bias.copy_(bvec)
        ctx = synth.hash_normal(N * P * T * D, 43, "cpu").view(N, P, T, D).requires_grad_()
        S, b = m(ctx)
        d = {"meta": np.asarray([N, P, T, D]), "ls": np.asarray(ls)}
        # a fixed cotangent to pin the backward (row f1 of SURVEY 8f comes later; data is cheap now)
        cot = synth.hash_normal(S.numel(), 44, "cpu").view_as(S)
        cot = cot * torch.ones(T, T).tril()[:, :, None, None]
        (S * cot).sum().backward()
        if S.numel() <= 200_000:
            d["S"] = S.detach().numpy()
        w = grad_weights(T)
        Sd = S.detach().numpy().astype(np.float64).reshape(T, T, -1)
        d["S_sum"] = Sd.sum(axis=(0, 1)); d["S_wsum"] = (Sd * w[:, :, None]).sum(axis=(0, 1))
        d["S_tril_sum"] = (Sd * np.tril(np.ones((T, T)))[:, :, None]).sum(axis=(0, 1))
        d["noise_absmax"] = np.float64(b.abs().max())
        d["dctx_sum"] = ctx.grad.numpy().astype(np.float64).sum(axis=(2, 3))
        d["dW_rows"] = m.map[0].weight.grad.numpy()[[0, 1, D - 1, D, 2 * D - 1, 2 * D]]
        d["dbias"] = m.map[0].bias.grad.numpy()
        save("scorer_" + name, d)


def case_large():
    """BASELINE.json configs[1] and [2] plus the headline size.  Outputs only."""
    torch.set_num_threads(8)
    # configs[1]: T=1024, NBatch=88, logProb fwd+bwd
    for T, B in ((1024, 88), (1024, 352)):
        seed = 1234
        t0 = time.time()
        score, noise = synth.crf_inputs(T, B, seed, "cpu", "randn")
        intervals = synth.synthetic_intervals(T, B, seed=seed)
        s = score.requires_grad_(); n = noise.requires_grad_()
        crf = REF.NeuralSemiCRFInterval(s, n)
        path = crf.evalPath(intervals); logz = crf.computeLogZ()
        lp = path - logz
        (-lp.sum()).backward()
        g = s.grad.numpy(); w = grad_weights(T)
        d = {"meta": np.asarray([T, B, seed]), "evalPath": path.detach().numpy(), "logZ": logz.detach().numpy(),
             "logProb": lp.detach().numpy(), "dNoise_logProb": n.grad.numpy(),
             "dScore_logProb_sum": g.astype(np.float64).sum(axis=(0, 1)),
             "dScore_logProb_wsum": np.einsum("ebc,eb->c", g.astype(np.float64), w.astype(np.float64)),
             "dScore_rows": np.asarray([1, 500, T - 1]), "dScore_logProb_rows": g[[1, 500, T - 1]][:, :, :8].copy()}
        print(f"  T={T} B={B} reference logProb fwd+bwd took {time.time() - t0:.1f}s")
        # f64 truth from the C oracle (REAL=double instantiation): lets the GPU test separate kernel
        # error from the reference's own fp32 round-off (~2e-6*|logZ| per marginal at this size)
        from oracle import oracle as cpu_oracle
        t0 = time.time()
        lz64, _, gn64, _, _ = cpu_oracle.forward_backward_f64(score.detach().numpy(), noise.detach().numpy())
        unc = np.ones((T - 1, B))
        for c, lst in enumerate(intervals):
            for b0, e0 in lst:
                unc[b0:e0, c] -= 1.0
        d["truth_logZ"] = lz64
        d["truth_dNoise_logProb"] = (gn64 - unc).astype(np.float32)      # d(-sum logProb)/d noise
        print(f"  f64 truth took {time.time() - t0:.1f}s; reference dNoise err vs truth "
              f"{np.abs(d['dNoise_logProb'] - d['truth_dNoise_logProb']).max():.2e}")
        save(f"large_T{T}_B{B}_randn", d)
        del g, s, n, crf, score, noise
    # configs[2]: Viterbi decode T=2048, NBatch=352, forcedStartPos set
    T, B, seed = 2048, 352, 1234
    score, noise = synth.crf_inputs(T, B, seed, "cpu", "randn")
    crf = REF.NeuralSemiCRFInterval(score, noise)
    d = {"meta": np.asarray([T, B, seed])}
    with torch.no_grad():
        for name, st in (("four", [4] * B), ("mixed", [(c * 37) % T for c in range(B)])):
            t0 = time.time()
            res = crf.decode(forcedStartPos=st)
            print(f"  T={T} B={B} reference decode({name}) took {time.time() - t0:.1f}s")
            p, o = pack(res)
            d[f"decode_{name}_offsets"] = o
            d[f"decode_{name}_sha256"] = np.asarray(digest(res))
            d[f"decode_{name}_head"] = p[:64]
            d[f"decode_{name}_start"] = np.asarray(st, np.int32)
    save(f"large_T{T}_B{B}_decode", d)


def reference_fetch_interval_features():
    """The reference's TransKun.fetchIntervalFeaturesBatch itself (ModelTransformer.py:501-532).  The module cannot be
    imported here (pretty_midi / torchaudio / moduleconf are absent), so the method is lifted out of the reference's
    source file at run time (ast: the one FunctionDef) and executed with the reference's own Util.listToIdx -- nothing of
    it is copied into this repository."""
    import ast
    import types
    src_path = "/root/reference/transkun/ModelTransformer.py"
    tree = ast.parse(open(src_path).read())
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "fetchIntervalFeaturesBatch":
            fn = node
    assert fn is not None
    mod = ast.Module(body=[fn], type_ignores=[])
    util_src = open("/root/reference/transkun/Util.py").read()
    util_tree = ast.parse(util_src)
    lti = [n for n in util_tree.body if isinstance(n, ast.FunctionDef) and n.name == "listToIdx"]
    assert len(lti) == 1
    ns = {"torch": torch}
    exec(compile(ast.Module(body=lti, type_ignores=[]), "/root/reference/transkun/Util.py", "exec"), ns)
    exec(compile(mod, src_path, "exec"), ns)
    return ns["fetchIntervalFeaturesBatch"]


def case_attr():
    """interval features for the attribute heads, from the reference's own method."""
    import types
    ref_fn = reference_fetch_interval_features()
    for name, (N, SYM, T, D, seed) in {"small": (2, 5, 40, 8, 3), "model": (2, 90, 96, 256, 4)}.items():
        ctx = synth.hash_normal(N * SYM * T * D, 100 + seed, "cpu").view(N, SYM, T, D)
        flat = synth.synthetic_intervals(T, N * SYM, seed=seed)
        if name == "small":
            flat[1] = []                                   # an empty chain, singletons and touching intervals
            flat[2] = [(3, 3), (4, 6), (6, 6), (7, 12)]
        batch = [flat[n * SYM:(n + 1) * SYM] for n in range(N)]
        self_like = types.SimpleNamespace(targetMIDIPitch=list(range(SYM)))      # the method only reads len(self.targetMIDIPitch)
        a, b, sym, sc = ref_fn(self_like, ctx, batch)
        pairs, off = pack(flat)
        d = {"meta": np.asarray([N, SYM, T, D, seed]), "pairs": pairs, "offsets": off,
             "symIdx": sym.numpy(), "scatterIdx": sc.numpy()}
        if name == "small":
            d["ctx_a"] = a.numpy(); d["ctx_b"] = b.numpy()
        else:                                              # larger case: digests (inputs are regenerable from the hash)
            d["ctx_a_sum"] = a.double().sum(0).numpy(); d["ctx_b_sum"] = b.double().sum(0).numpy()
            d["ab_sum"] = (a.double() * b.double()).sum(0).numpy()
            w = (torch.arange(a.shape[0], dtype=torch.float64) % 7 + 1)[:, None]
            d["ctx_a_wsum"] = (a.double() * w).sum(0).numpy(); d["ctx_b_wsum"] = (b.double() * w).sum(0).numpy()
        save(f"attr_{name}", d)


def ctx_weights(T, D):
    t = np.arange(T)[:, None]
    d = np.arange(D)[None, :]
    return (((t * 31 + d * 17) % 64).astype(np.float64) / 64.0)


SEGMENTS = {
    # name: (N, P, T, D, ctx scale, weight scale, seed)      -- must match tests/conftest.py SEGMENT_CASES
    "small": (2, 5, 48, 32, 0.5, 0.3, 3),
    "T691_P90": (1, 90, 691, 256, 1.0, 1.0, 5),        # BASELINE.json configs[3]: one 16 s segment, model-scale scores
    "T691_N4": (4, 90, 691, 256, 0.5, 0.3, 6),         # train.py's default batch of 4 segments (NBatch = 360), tame scores
}


def segment_inputs(N, P, T, D, cscale, wscale, seed, device="cpu"):
    """ctx, Linear weight/bias, interval lists, upstream gradient and forced starts of a segment case -- all from the
    integer hash (tests rebuild the same bits)."""
    ctx = synth.hash_normal(N * P * T * D, 200 + seed, device).view(N, P, T, D) * cscale
    W = synth.hash_normal((2 * D + 1) * D, 300 + seed, device).view(2 * D + 1, D) * (wscale / D ** 0.5)
    bias = synth.hash_normal(2 * D + 1, 400 + seed, device) * 0.1
    iv = synth.synthetic_intervals(T, N * P, seed=seed)
    gout = synth.hash_normal(N * P, 500 + seed, device)
    starts = [(c * 29 + 3) % (T // 2) for c in range(N * P)]
    return ctx, W, bias, iv, gout, starts


def case_segment(names=None):
    """Scorer -> CRF -> logProb -> backward, decode(forcedStartPos) -> attribute features, all by the REFERENCE's modules
    (LayersTransformer.ScaledInnerProductIntervalScorer, CRF.NeuralSemiCRFInterval, the lifted
    fetchIntervalFeaturesBatch) on the glue of ModelTransformer.py:199-225, :256-266, :537-582.  Pins SURVEY 8f rank 1
    (fused route) and BASELINE.json configs[3]'s scorer + CRF part at the model's real shape."""
    import types
    from transkun.LayersTransformer import ScaledInnerProductIntervalScorer
    ref_fn = reference_fetch_interval_features()
    torch.set_num_threads(8)
    for name, (N, P, T, D, cscale, wscale, seed) in SEGMENTS.items():
        if names and name not in names:
            continue
        t0 = time.time()
        ctx, W, bias, iv, gout, starts = segment_inputs(N, P, T, D, cscale, wscale, seed)
        m = ScaledInnerProductIntervalScorer(D, 1)
        with torch.no_grad():
            m.map[0].weight.copy_(W); m.map[0].bias.copy_(bias)
        ctx = ctx.clone().requires_grad_()
        S, b = m(ctx)                                                   # ModelTransformer.py:199-200
        crf = REF.NeuralSemiCRFInterval(S.flatten(-2, -1), b.flatten(-2, -1))      # :215-222
        path = crf.evalPath(iv)                                         # :263
        logz = crf.computeLogZ()                                        # :264
        lp = path - logz                                                # :265
        (lp * gout).sum().backward()
        C = N * P
        g = ctx.grad.numpy().astype(np.float64)
        w = ctx_weights(T, D)
        psel = sorted(set([0, 1, P // 2, P - 1]))
        tsel = sorted(set([0, 1, T // 3, T - 2, T - 1]))
        d = {"meta": np.asarray([N, P, T, D, seed]), "scales": np.asarray([cscale, wscale]),
             "logProb": lp.detach().numpy(), "logZ": logz.detach().numpy(), "evalPath": path.detach().numpy(),
             "dctx_sum": g.sum(axis=(2, 3)), "dctx_wsum": (g * w[None, None]).sum(axis=(2, 3)),
             "dctx_abs_sum": np.abs(g).sum(axis=(2, 3)),
             "psel": np.asarray(psel), "tsel": np.asarray(tsel),
             "dctx_rows": ctx.grad.numpy()[0][np.ix_(psel, tsel)].copy(),
             "dW_rows": m.map[0].weight.grad.numpy()[[0, 1, D - 1, D, 2 * D - 1, 2 * D]].copy(),
             "dW_rowsum": m.map[0].weight.grad.numpy().astype(np.float64).sum(axis=1),
             "dbias": m.map[0].bias.grad.numpy().copy()}
        Sd = S.detach()
        d["S_tril_sum"] = torch.tril(Sd.permute(2, 3, 0, 1).double()).sum(dim=(2, 3)).reshape(-1).numpy()
        print(f"  segment {name}: reference scorer + CRF logProb fwd+bwd {time.time() - t0:.1f}s; |logProb| max "
              f"{np.abs(d['logProb']).max():.3g}")
        # decode with forced starts (transcribeFrames :549) and the attribute-head inputs of the decoded path (:575-582)
        t0 = time.time()
        with torch.no_grad():
            crf_d = REF.NeuralSemiCRFInterval(Sd.flatten(-2, -1), b.detach().flatten(-2, -1))
            dec = crf_d.decode(forcedStartPos=starts, forward=False)
            dec0 = crf_d.decode()
        pr, off = pack(dec)
        d["decode_start"] = np.asarray(starts, np.int32)
        d["decode_pairs"] = pr; d["decode_offsets"] = off
        d["decode_sha256"] = np.asarray(digest(dec))
        d["decode_nostart_sha256"] = np.asarray(digest(dec0))
        d["decode_nostart_offsets"] = pack(dec0)[1]
        batch = [dec[n * P:(n + 1) * P] for n in range(N)]
        if sum(len(x) for x in dec) > 0:
            self_like = types.SimpleNamespace(targetMIDIPitch=list(range(P)))
            a, bb, sym, sc = ref_fn(self_like, ctx.detach(), batch)
            wv = (torch.arange(a.shape[0], dtype=torch.float64) % 7 + 1)[:, None]
            d["attr_a_sum"] = a.double().sum(0).numpy(); d["attr_b_sum"] = bb.double().sum(0).numpy()
            d["attr_ab_sum"] = (a.double() * bb.double()).sum(0).numpy()
            d["attr_a_wsum"] = (a.double() * wv).sum(0).numpy(); d["attr_b_wsum"] = (bb.double() * wv).sum(0).numpy()
            d["attr_symIdx"] = sym.numpy().astype(np.int32); d["attr_scatterIdx"] = sc.numpy().astype(np.int32)
        print(f"  segment {name}: reference decode x2 + features {time.time() - t0:.1f}s; {len(pr)} intervals")
        save(f"segment_{name}", d)
        del S, crf, ctx, g


MODEL_LARGE = [(1024, 88, 1234), (691, 90, 77), (691, 360, 78)]


def case_model_large():
    """CRF kernels on model-like scores (+-1e2..1e3 * |e-b|, noise == 0: SURVEY hard part 6) at the large sizes, and at the
    model's real shape T=691 x 90/360 chains; randn at T=691 too.  Outputs only."""
    torch.set_num_threads(8)
    from oracle import oracle as cpu_oracle
    for T, B, seed in MODEL_LARGE:
        for kind in (("model",) if T == 1024 else ("model", "randn")):
            t0 = time.time()
            score, noise = synth.crf_inputs(T, B, seed, "cpu", kind)
            intervals = synth.synthetic_intervals(T, B, seed=seed)
            s = score.requires_grad_(); n = noise.requires_grad_()
            crf = REF.NeuralSemiCRFInterval(s, n)
            path = crf.evalPath(intervals); logz = crf.computeLogZ()
            lp = path - logz
            (-lp.sum()).backward()
            g = s.grad.numpy(); w = grad_weights(T)
            rows = [1, T // 2, T - 1]
            d = {"meta": np.asarray([T, B, seed]), "evalPath": path.detach().numpy(), "logZ": logz.detach().numpy(),
                 "logProb": lp.detach().numpy(), "dNoise_logProb": n.grad.numpy(),
                 "dScore_logProb_sum": g.astype(np.float64).sum(axis=(0, 1)),
                 "dScore_logProb_wsum": np.einsum("ebc,eb->c", g.astype(np.float64), w.astype(np.float64)),
                 "dScore_rows": np.asarray(rows), "dScore_logProb_rows": g[rows][:, :, :8].copy()}
            lz64, _, gn64, _, _ = cpu_oracle.forward_backward_f64(score.detach().numpy(), noise.detach().numpy())
            unc = np.ones((T - 1, B))
            for c, lst in enumerate(intervals):
                for b0, e0 in lst:
                    unc[b0:e0, c] -= 1.0
            d["truth_logZ"] = lz64
            d["truth_dNoise_logProb"] = (gn64 - unc).astype(np.float32)
            with torch.no_grad():
                sd, nd = score.detach(), noise.detach()
                crf_d = REF.NeuralSemiCRFInterval(sd, nd)
                for nm, st in (("none", None), ("mixed", [(c * 37) % T for c in range(B)])):
                    res = crf_d.decode(forcedStartPos=st)
                    p, o = pack(res)
                    d[f"decode_{nm}_offsets"] = o
                    d[f"decode_{nm}_sha256"] = np.asarray(digest(res))
                    d[f"decode_{nm}_head"] = p[:64]
                    if st is not None:
                        d[f"decode_{nm}_start"] = np.asarray(st, np.int32)
            print(f"  T={T} B={B} {kind}: reference logProb fwd+bwd + 2 decodes + f64 truth {time.time() - t0:.1f}s; "
                  f"ref dNoise err vs truth {np.abs(d['dNoise_logProb'] - d['truth_dNoise_logProb']).max():.2e}")
            save(f"large_T{T}_B{B}_{kind}", d)
            del g, s, n, crf, score, noise


TRANSCRIBE = {
    # name: (D, hidden, fs, hop, window, segment s, step s, audio s, ctx scale, weight scale, seed)   -- tests/segment_common.py mirrors this
    "small": (32, 48, 44100, 1024, 4096, 1.2, 0.6, 1.9, 0.5, 0.3, 11),
    "real": (256, 512, 44100, 1024, 4096, 16.0, 8.0, 37.0, 0.5, 0.3, 12),        # the shipped 2.0.conf geometry: T = 691 frames per segment
}


def transcribe_inputs(name, device="cpu"):
    """Everything the segment loop consumes besides the (out of scope) backbone: one ctx tensor per segment, the scorer's and
    the two attribute heads' weights -- all from the integer hash."""
    import math
    D, H, fs, hop, win, seg_s, step_s, audio_s, cscale, wscale, seed = TRANSCRIBE[name]
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
                ctxs=ctxs, W=W, bias=bias, heads=heads, n_sample_unpadded=int(audio_s * fs))


def lift_reference_transcribe():
    """The reference's segment loop itself: TransKun.transcribe (ModelTransformer.py:726-848), .transcribeFrames (:537-725)
    and .fetchIntervalFeaturesBatch (:501-532) lifted out of the source file with ast at run time, together with Note and
    resolveOverlapping (Data.py:20-30, :170-201), makeFrame and listToIdx (Util.py:21-43, :173-176).  The module itself
    cannot be imported here (pretty_midi / torchaudio / moduleconf are absent); nothing of it is copied into this
    repository."""
    import ast
    import math
    from collections import defaultdict
    ns = {"torch": torch, "F": torch.nn.functional, "nn": torch.nn, "math": math, "defaultdict": defaultdict}

    def lift(path, names, kinds=(ast.FunctionDef, ast.ClassDef)):
        tree = ast.parse(open(path).read())
        found = {}
        for node in ast.walk(tree):
            if isinstance(node, kinds) and node.name in names and node.name not in found:
                found[node.name] = node
        assert set(found) == set(names), (names, list(found))
        exec(compile(ast.Module(body=[found[n] for n in names], type_ignores=[]), path, "exec"), ns)

    lift("/root/reference/transkun/Data.py", ["Note", "resolveOverlapping", "validateNotes"])
    lift("/root/reference/transkun/Util.py", ["makeFrame", "listToIdx"])
    lift("/root/reference/transkun/ModelTransformer.py", ["fetchIntervalFeaturesBatch", "transcribeFrames", "transcribe"])
    return ns


def case_transcribe(names=None):
    """the transcription segment loop (forcedStartPos hand-off, lastP, event assembly, incomplete-event
    merge) run by the reference's own code on synthetic per-segment ctx (the backbone is replaced by a stub that hands the
    reference scorer + CRF the prepared ctx of the segment)."""
    import types
    from transkun.LayersTransformer import ScaledInnerProductIntervalScorer
    ns = lift_reference_transcribe()
    torch.set_num_threads(8)
    for name in TRANSCRIBE:
        if names and name not in names:
            continue
        t0 = time.time()
        I = transcribe_inputs(name)
        D, H, P, T = I["D"], I["H"], I["P"], I["T"]
        scorer = ScaledInnerProductIntervalScorer(D, 1)
        with torch.no_grad():
            scorer.map[0].weight.copy_(I["W"]); scorer.map[0].bias.copy_(I["bias"])

        def head(nout, w):
            m = torch.nn.Sequential(torch.nn.Linear(3 * D, H), torch.nn.GELU(), torch.nn.Dropout(0.1), torch.nn.Linear(H, nout))
            with torch.no_grad():
                m[0].weight.copy_(w[0]); m[0].bias.copy_(w[1]); m[3].weight.copy_(w[2]); m[3].bias.copy_(w[3])
            return m.eval()

        vel, of = head(128, I["heads"]["velocity"]), head(4, I["heads"]["of"])
        rec = {"start": [], "lastP": [], "pairs": [], "offsets": [], "velocity": [], "of": []}
        state = {"i": 0}

        class CrfRec:
            def __init__(self, crf): self.crf = crf
            def decode(self, forcedStartPos=None, forward=False):
                out = self.crf.decode(forcedStartPos=forcedStartPos, forward=forward)
                rec["start"].append(np.asarray(forcedStartPos, np.int32))
                p, o = pack(out)
                rec["pairs"].append(p); rec["offsets"].append(o)
                return out

        def processFramesBatch(framesBatch):
            ctx = I["ctxs"][state["i"]]
            state["i"] += 1
            with torch.no_grad():
                S, b = scorer(ctx)
                return CrfRec(REF.NeuralSemiCRFInterval(S.flatten(-2, -1), b.flatten(-2, -1))), ctx

        vel.register_forward_hook(lambda m, i, o: rec["velocity"].append(o.detach().numpy().copy()))
        of.register_forward_hook(lambda m, i, o: rec["of"].append(o.detach().numpy().copy()))
        me = types.SimpleNamespace(hopSize=I["hop"], windowSize=I["win"], fs=I["fs"], segmentHopSizeInSecond=I["step_s"],
                                   segmentSizeInSecond=I["seg_s"], targetMIDIPitch=[-64, -67] + list(range(21, 108 + 1)),
                                   velocityPredictor=vel, refinedOFPredictor=of, processFramesBatch=processFramesBatch)
        me.fetchIntervalFeaturesBatch = types.MethodType(ns["fetchIntervalFeaturesBatch"], me)
        tf = ns["transcribeFrames"]

        def transcribeFrames(framesBatch, **kw):
            notes, lastP = tf(me, framesBatch, **kw)
            rec["lastP"].append(np.asarray(lastP, np.int32))
            return notes, lastP

        me.transcribeFrames = transcribeFrames
        x = synth.hash_normal(I["n_sample_unpadded"], 900, "cpu").view(-1, 1)           # [nSample, nChannel]; only its length matters
        ns_resolve = ns["resolveOverlapping"]
        kept = {}
        ns["resolveOverlapping"] = lambda ev: kept.setdefault("before", [(e.start, e.end, e.pitch, e.velocity, e.hasOnset, e.hasOffset) for e in ev]) and ns_resolve(ev) or ns_resolve(ev)
        with torch.no_grad():
            events = ns["transcribe"](me, x)
        ns["resolveOverlapping"] = ns_resolve

        def table(evs):
            return (np.asarray([[e[0], e[1]] for e in evs], np.float64).reshape(-1, 2), np.asarray([[e[2], e[3]] for e in evs], np.int32).reshape(-1, 2),
                    np.asarray([[e[4], e[5]] for e in evs], np.uint8).reshape(-1, 2))

        final = [(e.start, e.end, e.pitch, e.velocity, e.hasOnset, e.hasOffset) for e in events]
        d = {"meta": np.asarray([I["n_seg"], P, T, D, H]), "n_lastP": np.asarray([len(x) for x in rec["lastP"]])}
        for k, evs in (("final", final), ("merged", kept["before"])):
            t, pv, fl = table(evs)
            d[k + "_times"] = t; d[k + "_pitch_velocity"] = pv; d[k + "_flags"] = fl
        for i in range(len(rec["start"])):
            d[f"seg{i}_start"] = rec["start"][i]; d[f"seg{i}_pairs"] = rec["pairs"][i]; d[f"seg{i}_offsets"] = rec["offsets"][i]
        for i in range(len(rec["lastP"])):
            d[f"seg{i}_lastP"] = rec["lastP"][i]
        for i in range(len(rec["velocity"])):
            d[f"head{i}_velocity_argmax"] = rec["velocity"][i].argmax(-1).astype(np.int32)
            d[f"head{i}_of"] = rec["of"][i].astype(np.float32)
        print(f"  transcribe {name}: {I['n_seg']} segments of T={T}, {len(final)} events after the merge ({len(kept['before'])} before "
              f"resolveOverlapping), {sum(len(p) for p in rec['pairs'])} decoded intervals, {time.time() - t0:.1f}s")
        save(f"transcribe_{name}", d)


def case_frontend():
    """framing + six-window spectrum by the reference's own classes (Util.py:21-124; MelSpectrum itself needs
    torchaudio, absent here: the mel filterbank stays parity-unpinned)."""
    from transkun.Util import GaussianWindows, Spectrum, makeFrame
    x = synth.hash_normal(2 * 20000, 950, "cpu").view(2, 20000)
    d = {}
    for hop, win in ((1024, 4096), (160, 400)):
        fr = makeFrame(x, hop, win)
        d[f"frames_{hop}_shape"] = np.asarray(fr.shape)
        d[f"frames_{hop}_rowsum"] = fr.double().sum(-1).numpy()
        d[f"frames_{hop}_first"] = fr[0, :2, :8].numpy().copy(); d[f"frames_{hop}_last"] = fr[1, -2:, -8:].numpy().copy()
    torch.manual_seed(3)
    gw = GaussianWindows(5, 4096)
    with torch.no_grad():
        gw.sigma.add_(synth.hash_normal(5, 951, "cpu") * 0.3); gw.center.add_(synth.hash_normal(5, 952, "cpu") * 0.3)
    d["gw_sigma"] = gw.sigma.detach().numpy().copy(); d["gw_center"] = gw.center.detach().numpy().copy()
    Y = gw.get().detach()
    d["gw_colsum"] = Y.double().sum(0).numpy(); d["gw_rows"] = Y[[0, 1000, 2048, 4095]].numpy().copy()
    sp = Spectrum(4096, nExtraWins=5)
    with torch.no_grad():
        sp.winGen.sigma.copy_(gw.sigma); sp.winGen.center.copy_(gw.center)
        frames = makeFrame(x, 1024, 4096)
        S = sp(frames)                                        # [2, nFrame, 2049, 6] complex
    P = S.abs().pow(2)
    d["spec_shape"] = np.asarray(S.shape)
    d["spec_power_sum"] = P.double().sum(dim=(0, 1, 2)).numpy()
    d["spec_power_bins"] = P[0, 3, [0, 1, 17, 500, 2048], :].numpy().copy()
    d["spec_re_im"] = torch.view_as_real(S[1, 5, [2, 300], :]).numpy().copy()
    save("frontend", d)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    todo = a.only.split(",") if a.only else ["minimal", "edges", "medium", "scorer", "attr", "segment", "transcribe", "frontend"] + (["large", "model_large"] if a.large else [])
    for t in todo:
        print("case", t)
        {"minimal": case_minimal, "edges": case_edges, "medium": case_medium, "scorer": case_scorer,
         "large": case_large, "attr": case_attr, "segment": case_segment, "model_large": case_model_large, "transcribe": case_transcribe, "frontend": case_frontend}[t]()
