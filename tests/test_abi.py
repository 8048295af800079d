"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/semicrf_hip.h declares; the Python mirror keeps the reference's surface and
refuses to run without a GPU (no CPU fallback)."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "semicrf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n.startswith(("semicrf_", "interval_score", "interval_features", "segment_", "scorer_proj", "scorer_merge", "scorer_stage"))))


def test_library_builds_and_exports_header_symbols():
    from transkun_amd import _build, _lib
    path = _build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    declared = _declared_functions()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in semicrf_hip.h but not exported"
    assert set(_lib.EXPORTED) == set(declared)
    loaded = _lib.load()
    assert loaded.semicrf_abi_version() == 2
    assert loaded.semicrf_workspace_bytes(_lib.OP_VITERBI, 1024, 352) > 0


def test_invalid_arguments_are_rejected_without_gpu():
    from transkun_amd import _lib
    lib = _lib.load()
    rc = lib.semicrf_logz_fwd(None, None, 0, 4, None, None, None, 0, None)
    assert rc == 1
    assert b"must be >= 1" in lib.semicrf_last_error()
    rc = lib.semicrf_logz_fwd(None, None, 8, 4, None, None, None, 0, None)
    assert rc == 1 and b"NULL" in lib.semicrf_last_error()
    # full_square: triangle mode 0..2 in bits 0-1, SEMICRF_SCORE_BF16X3 (4) in bit 2, nothing else
    import ctypes
    fake = ctypes.c_void_p(4096)            # never dereferenced: the argument check comes first
    for bad in (3, 7, 8, -1):
        rc = lib.interval_score_fwd(fake, fake, fake, 2, 8, 64, 64, 64, 1, 0.125, 0, bad, fake, None, None)
        assert rc == 1 and b"full_square" in lib.semicrf_last_error(), bad


def test_python_surface_matches_reference():
    from transkun_amd import CRF
    cls = CRF.NeuralSemiCRFInterval
    assert list(inspect.signature(cls.__init__).parameters) == ["self", "score", "noiseScore"]
    sig = inspect.signature(cls.decode)
    assert list(sig.parameters) == ["self", "forcedStartPos", "forward"]
    assert sig.parameters["forcedStartPos"].default is None and sig.parameters["forward"].default is False
    assert list(inspect.signature(cls.evalPath).parameters) == ["self", "intervals"]
    assert inspect.signature(cls.computeLogZ).parameters["noBackward"].default is False
    assert list(inspect.signature(cls.logProb).parameters) == ["self", "intervals", "noBackward"]
    for fn in ("viterbi", "viterbiBackward", "computeLogZ", "forward_backward", "evalPath", "computeLogZFasterGrad"):
        assert hasattr(CRF, fn)
    crf = cls(torch.zeros(3, 3, 2), torch.zeros(2, 2))
    assert crf.score.shape == (3, 3, 2) and crf.noiseScore.shape == (2, 2)


def test_gpu_only_entry_points_refuse_cpu_tensors():
    """CPU tensors are dispatched to the shim's host kernels for the CRF ops (tests/test_cpu_dispatch.py); the GPU-only entry
    points say so instead of computing anything elsewhere, and no op has a fallback from one device to another."""
    from transkun_amd import _lib
    with pytest.raises(RuntimeError, match="only runs on an AMD GPU"):
        _lib.require_gpu(torch.zeros(1), "ctx")
    ops = _lib.ops()
    with pytest.raises((NotImplementedError, RuntimeError)):        # no CPU kernel registered for the segment loop
        ops.segment_onset_filter(torch.zeros(1, 2, dtype=torch.int32), torch.zeros(2, dtype=torch.int32), 1, 4,
                                 torch.zeros(1, 2, dtype=torch.int32), torch.zeros(2, dtype=torch.int32), torch.zeros(1, dtype=torch.int32))


def test_shape_asserts_like_reference():
    from transkun_amd import CRF
    with pytest.raises(AssertionError):
        CRF.NeuralSemiCRFInterval(torch.zeros(4, 5, 2), torch.zeros(3, 2)).computeLogZ()
    with pytest.raises(AssertionError):
        CRF.NeuralSemiCRFInterval(torch.zeros(4, 4, 2), torch.zeros(4, 2)).computeLogZ()
    with pytest.raises(AssertionError):
        CRF.NeuralSemiCRFInterval(torch.zeros(4, 4), torch.zeros(3, 2)).decode()


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "transkun_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, f


def test_pack_unpack_roundtrip():
    from transkun_amd.CRF.NeuralSemiCRFInterval import pack_intervals, unpack_intervals
    iv = [[(0, 2), (4, 6), (6, 6), (7, 8)], [(1, 2), (3, 5), (19, 19)], [(0, 0), (4, 7)], []]
    pairs, offsets = pack_intervals(iv, 200, 4, "cpu")
    assert offsets.tolist() == [0, 4, 7, 9, 9]
    assert unpack_intervals(pairs, offsets) == iv
    with pytest.raises(IndexError):
        pack_intervals([[(0, 200)], [], [], []], 200, 4, "cpu")
    p2, o2 = pack_intervals([[], []], 5, 2, "cpu")
    assert o2.tolist() == [0, 0, 0]
    assert unpack_intervals(p2[:0], o2) == [[], []]


def test_synth_numpy_torch_identical():
    import numpy as np
    from transkun_amd import synth
    a = synth.hash_normal(10007, 99, "cpu").numpy()
    b = synth.hash_normal_numpy(10007, 99)
    assert np.array_equal(a, b)
    assert abs(float(a.mean())) < 0.05 and 1.0 < float(a.std()) < 1.3
    iv = synth.synthetic_intervals(256, 12, seed=3)
    for lst in iv:
        last = -1
        for (b0, e0) in lst:
            assert 0 <= b0 <= e0 < 256 and b0 >= last
            last = e0


def test_scorer_regrouped_linear_rows():
    """The mirror runs the reference's Linear ([q | k | diag] rows, LayersTransformer.py:392-397) as [q | diag | pad] and k:
    same numbers as slicing the packed output, gradients reach the original parameter rows, pad rows stay zero."""
    import torch
    import torch.nn.functional as F
    from transkun_amd.scorer import QPAD, ScaledInnerProductIntervalScorer, qd_weights
    D = 8
    m = ScaledInnerProductIntervalScorer(D, 1)
    assert list(m.state_dict().keys()) == ["map.0.weight", "map.0.bias"]          # checkpoints of the reference load
    W, b = m.map[0].weight, m.map[0].bias
    assert W.shape == (2 * D + 1, D)
    x = torch.randn(3, 5, D)
    packed = F.linear(x, W, b)
    Wqd, bqd = qd_weights(W, b, D)
    qd = F.linear(x, Wqd, bqd)
    assert qd.shape[-1] == D + QPAD and (D + QPAD) % 4 == 0
    assert torch.allclose(qd[..., :D], packed[..., :D], atol=1e-6)
    assert torch.allclose(qd[..., D], packed[..., 2 * D], atol=1e-6)
    assert float(qd[..., D + 1:].abs().max()) == 0.0
    qd.sum().backward()
    g = W.grad
    assert float(g[D:2 * D].abs().max()) == 0.0 and float(g[:D].abs().max()) > 0 and float(g[2 * D].abs().max()) > 0


def test_torch_ops_shim_registers_every_op():
    """libsemicrf_torch.so (LibTorch stable ABI, csrc/torch_ops.cpp) registers the compute entry points of the C ABI as
    torch.ops.semicrf.*: the CRF ops for the dispatch keys CUDA (HIP kernels) and CPU (the shim's own host kernels), the
    scorer / gather / segment ops for CUDA only (a CPU tensor fails in the dispatcher; nothing falls back across devices)."""
    import torch
    from transkun_amd import _lib
    ops = _lib.ops()
    for name in ("logz_fwd", "logz_bwd", "beta", "viterbi", "eval_path", "eval_path_bwd", "interval_score_fwd",
                 "interval_score_bwd_ws", "interval_score_bwd_fused_ws", "interval_score_path_bwd",
                 "interval_features_gather", "interval_features_gather_bwd"):
        assert hasattr(ops, name), name
    s = torch.zeros(4, 4, 2); n = torch.zeros(3, 2)
    lz = torch.full((2,), float("nan"))
    ops.logz_fwd(s, n, lz, torch.zeros(4, 2), True, torch.zeros(0, dtype=torch.uint8))
    # all-zero scores: logZ = log(number of segmentations weighted by 2 per frame) -- finite, equal for both chains
    assert bool(torch.isfinite(lz).all()) and float(lz[0]) == float(lz[1])
    with pytest.raises(NotImplementedError):
        ops.interval_score_fwd(torch.zeros(2, 4, 8), torch.zeros(2, 4, 8), torch.zeros(2, 4), torch.zeros(2, 4), 2, 4, 8, 8, 8, 1, 0, 1.0, 0, 0, 2, 2,
                               torch.zeros(4, 4, 2), torch.zeros(3, 2))


def test_workgroup_role_map_is_a_permutation():
    """The sweeps deal roles by workgroup index (rings of one 32-chain panel group on one XCD): for every launch shape the map
    must hit every ticket exactly once, and the rings of a panel group must share their index modulo 8 when they fit."""
    from transkun_amd import _lib
    lib = _lib.load()
    for n_spine, grid in [(88, 256), (90, 256), (22, 256), (12, 40), (5, 256), (128, 256), (1, 9), (3, 7), (64, 100), (88, 88), (2, 2),
                          (23, 256), (75, 256), (128, 304)]:
        t = [lib.semicrf_debug_wg_ticket(n_spine, grid, b) for b in range(grid)]
        assert sorted(t) == list(range(grid)), (n_spine, grid)
        where = {tk: b for b, tk in enumerate(t)}
        if t != list(range(grid)):                           # the XCD-aware map is in force
            for g in range((n_spine + 7) // 8):
                xs = {where[sg] % 8 for sg in range(8 * g, min(8 * g + 8, n_spine))}
                assert len(xs) == 1, (n_spine, grid, g, xs)


@pytest.mark.parametrize("src", ["proj_gemm.hip", "scorer_bwd_gemm.hip"])
def test_no_register_copies_ahead_of_the_lds_waits(src):
    """The matrix-core kernels issue their LDS reads through inline asm and wait for them in a later asm statement; a compiler copy
    of a destination register between the two reads stale data (tools/check_asm_waits.py: the failure, seen on proj_gemm.hip in round
    4, depends on timing and passes small tests).  The generated gfx950 assembly must show none, nor a select on a stale SCC."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_asm_waits", os.path.join(root, "tools", "check_asm_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = mod.compile_to_asm(os.path.join(root, "transkun_amd", "csrc", src), [])
    assert mod.check(text, src) == 0
