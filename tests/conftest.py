import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real AMD GPU (run on the MI355X box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def unpack_lists(pairs, offsets):
    flat = [tuple(int(x) for x in p) for p in np.asarray(pairs).reshape(-1, 2)]
    off = [int(x) for x in offsets]
    return [flat[off[c]:off[c + 1]] for c in range(len(off) - 1)]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


EDGE_CASES = [
    # name, T, B, kind, seed, transform   (must match tools/make_golden.py EDGE)
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


def edge_inputs(T, B, kind, seed, tr, device="cpu"):
    """Same construction as tools/make_golden.py:edge_inputs (inputs are not stored in fixtures)."""
    import torch
    from transkun_amd import synth
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
        score.mul_(4000.0)
        d.copy_(dd)
    return score.contiguous().to(device), noise.contiguous().to(device)


def grad_weights(T):
    e = np.arange(T)[:, None]
    b = np.arange(T)[None, :]
    return (((e * 31 + b * 17) % 64).astype(np.float32) / 64.0)


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))) if a.size else 0.0


def score_backward_torch(dS, q, k, N, P, T, D, mode, full=False):
    """Plain-torch differentiation of the interval scorer (LayersTransformer.py:410-433) for a cotangent dS [T,T,N,P]: the
    reference the HIP backward kernels are tested against (test infrastructure; the product has no torch formulation)."""
    import math
    import torch
    C = N * P
    g = dS.reshape(T, T, C).permute(2, 0, 1)                     # [C, e, b]
    t = torch.arange(T, device=g.device)
    ln = (t[:, None] - t[None, :]).abs().to(torch.float32)
    if mode == 1:
        ln = ln.sqrt()
    elif mode == 2:
        ln = torch.ones_like(ln)
    gl = g * ln
    if not full:
        gl = torch.tril(gl)                                       # only e >= b was produced by the forward
    qs = 1.0 / math.sqrt(D)
    dq = torch.bmm(gl, k) * qs                                    # [C,T,D]
    dk = torch.bmm(gl.transpose(1, 2), q) * qs
    ddiag = torch.diagonal(g, dim1=1, dim2=2)                     # [C,T]
    return dq.view(N, P, T, D), dk.view(N, P, T, D), ddiag.reshape(N, P, T).contiguous()


def interval_score_variant(variant, q, k, dg, T, C, D, qs, mode, full, group=0, pitch=0, rowc=None):
    """interval_score_fwd with ONE of its kernels forced (semicrf_debug_score_variant).  The release library carries the register-load
    (0), streaming (32), tiled (2) and three-limb kernels; the 64- / 128-row tile kernels (64, 128) -- the bit-level reference of the
    tiled kernel -- live in the DEBUG library (transkun_amd/libsemicrf_hip_debug.so), called here through ctypes on the same C ABI."""
    import ctypes
    import torch
    from transkun_amd import _lib
    from transkun_amd.scorer import _interval_score_raw
    if variant not in (64, 128):
        lib = _lib.load()
        lib.semicrf_debug_score_variant(int(variant))
        try:
            return _interval_score_raw(q, k, dg, T, C, D, qs, mode, full, group, pitch, rowc=rowc)
        finally:
            lib.semicrf_debug_score_variant(-1)
    lib = _lib.load_debug()
    if not group:
        group = pitch = C
    Cs = C // group * pitch
    S = torch.empty(T, T, Cs, dtype=torch.float32, device=q.device)
    noise = torch.empty(max(T - 1, 0), Cs, dtype=torch.float32, device=q.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.semicrf_debug_score_variant(int(variant))
    try:
        rc = lib.interval_score_fwd_pc(vp(q), vp(k), vp(dg), vp(rowc) if rowc is not None else None, C, T, D, q.stride(-2), k.stride(-2),
                                       dg.stride(-1), rowc.stride(-1) if rowc is not None else 1, float(qs), int(mode), int(full), int(group),
                                       int(pitch), vp(S), vp(noise), ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
    finally:
        lib.semicrf_debug_score_variant(-1)
    assert rc == 0, lib.semicrf_last_error().decode()
    return S, noise
