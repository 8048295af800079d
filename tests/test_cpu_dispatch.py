"""BASELINE.json configs[0]: crfMinimalExample.py on the CPU (T=200, NBatch=4: logProb, decode, forced-start decode), and the
rest of the CRF surface on CPU tensors: torch.ops.semicrf.* dispatches them to the product's own host kernels
(transkun_amd/csrc/cpu_ops.cpp).  Same checks as the GPU parity tests (tests/test_gpu_parity.py::_check_case), against the
golden vectors the reference itself produced (tools/make_golden.py): decode bit-exact, logZ / logProb 1e-5, marginals at the
reference's fp32 noise floor.  Runs without a GPU."""
import numpy as np
import pytest
import torch

from conftest import EDGE_CASES, edge_inputs, load_golden, rel_err, unpack_lists
from test_gpu_parity import _check_case, grad_tol


def test_crf_minimal_example_on_cpu(oracle):
    """The reference's example script, step by step (crfMinimalExample.py:9-38), on CPU tensors."""
    from transkun_amd import CRF
    g = load_golden("minimal_T200_B4")
    score = torch.from_numpy(g["score"]).requires_grad_()
    noise = torch.from_numpy(g["noise"]).requires_grad_()
    intervals = unpack_lists(g["intervals_pairs"], g["intervals_offsets"])
    crf = CRF.NeuralSemiCRFInterval(score, noise)
    logProb = crf.logProb(intervals)
    assert rel_err(logProb.detach().numpy(), g["logProb"]) < 1e-5
    (-logProb.sum()).backward()
    assert rel_err(score.grad.numpy()[g["rows"]], g["dScore_logProb_rows"]) < grad_tol(g["fb_logZ"])     # the reference's fp32 floor
    assert crf.decode() == unpack_lists(g["decode_none_bwd_pairs"], g["decode_none_bwd_offsets"])
    start = [int(x) for x in g["decode_four_bwd_start"]]                       # crfMinimalExample.py:38: forcedStartPos=[4]*NBatch
    assert start == [4] * 4
    assert crf.decode(forcedStartPos=start) == unpack_lists(g["decode_four_bwd_pairs"], g["decode_four_bwd_offsets"])


def test_minimal_golden_cpu(oracle):
    g = load_golden("minimal_T200_B4")
    _check_case(g, torch.from_numpy(g["score"]), torch.from_numpy(g["noise"]), oracle)


@pytest.mark.parametrize("case", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_cases_cpu(oracle, case):
    name, T, B, kind, seed, tr = case
    g = load_golden("edge_" + name)
    score, noise = edge_inputs(T, B, kind, seed, tr, "cpu")
    _check_case(g, score, noise, oracle)


@pytest.mark.parametrize("kind", ["randn", "model"])
def test_medium_cpu(oracle, kind):
    from transkun_amd import synth
    g = load_golden(f"medium_T256_B90_{kind}")
    T, B, seed = (int(x) for x in g["meta"])
    score, noise = synth.crf_inputs(T, B, seed, "cpu", kind)
    _check_case(g, score, noise, oracle, check_oracle=False)


def test_T1_cpu():
    from transkun_amd import CRF
    score = torch.tensor([[[0.5, -0.25, 2.0]]])
    crf = CRF.NeuralSemiCRFInterval(score, torch.zeros(0, 3))
    assert torch.allclose(crf.computeLogZ(), torch.nn.functional.softplus(score[0, 0]), atol=1e-6)
    assert crf.decode() == [[(0, 0)], [], [(0, 0)]]
    assert crf.decode(forward=True) == [[(0, 0)], [], [(0, 0)]]
    assert torch.allclose(crf.evalPath([[(0, 0)], [], []]), torch.tensor([0.5, 0.0, 0.0]))


def test_cpu_ops_check_their_arguments():
    """The ops are dispatcher-visible: a wrong dtype, a short buffer or an interval out of range is an error, not an
    out-of-bounds access (advisor, round 2)."""
    from transkun_amd import _lib
    ops = _lib.ops()
    T, B = 6, 3
    score = torch.zeros(T, T, B); noise = torch.zeros(T - 1, B)
    logz = torch.empty(B); v = torch.empty(T, B); ws = torch.empty(0, dtype=torch.uint8)
    ops.logz_fwd(score, noise, logz, v, True, ws)
    with pytest.raises(RuntimeError):
        ops.logz_fwd(score.double(), noise, logz, v, True, ws)
    with pytest.raises(RuntimeError):
        ops.logz_fwd(score, noise, torch.empty(B - 1), v, True, ws)
    with pytest.raises(RuntimeError):
        ops.logz_fwd(score, noise[:-1], logz, v, True, ws)
    pairs = torch.tensor([[0, 9]], dtype=torch.int32); offsets = torch.tensor([0, 1, 1, 1], dtype=torch.int32)
    with pytest.raises(RuntimeError):
        ops.eval_path(score, noise, pairs, 1, offsets, torch.empty(B), ws)
    with pytest.raises(RuntimeError):
        ops.viterbi(score, noise, torch.tensor([0, 0, 7], dtype=torch.int32), True, False, torch.empty(2 * T * B, 2, dtype=torch.int32),
                    torch.empty(B + 1, dtype=torch.int32), ws)
    with pytest.raises(RuntimeError):
        ops.viterbi(score, noise, torch.zeros(B, dtype=torch.int64), True, False, torch.empty(2 * T * B, 2, dtype=torch.int32),
                    torch.empty(B + 1, dtype=torch.int32), ws)
