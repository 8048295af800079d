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


def _restated_logz(score, noise):
    """NeuralSemiCRFInterval.py:206-246 restated with torch ops (float64, autograd-traceable) -- test infrastructure."""
    import torch.nn.functional as F
    T = score.shape[0]
    v = [F.softplus(score[0, 0])]
    for i in range(1, T):
        cand = torch.stack([v[i - 1] + noise[i - 1]] + [v[j] + score[i, j] for j in range(i)])
        v.append(torch.logsumexp(cand, 0) + F.softplus(score[i, i]))
    return v[-1]


@pytest.mark.parametrize("mask", ["one_cell", "whole_row"])
def test_masked_cells_cpu(mask):
    """-inf scores (masked intervals): torch.logsumexp ignores them and the reference's gradients stay finite; the host
    kernels must too (Lse::push(-inf) on an empty accumulator gave NaN)."""
    from transkun_amd import CRF, synth
    T, B = 6, 5
    score, noise = synth.crf_inputs(T, B, 11, "cpu")
    if mask == "one_cell":
        score[T - 1, 0, :] = float("-inf")
    else:                                   # every way into frame 1: the skip and the interval (0, 1)
        noise[0, :] = float("-inf")
        score[1, 0, :] = float("-inf")
    s64 = score.double().requires_grad_(); n64 = noise.double().requires_grad_()
    want = _restated_logz(s64, n64)
    want.sum().backward()
    s = score.clone().requires_grad_(); n = noise.clone().requires_grad_()
    got = CRF.NeuralSemiCRFInterval(s, n).computeLogZ()
    assert torch.isfinite(got).all() and torch.allclose(got.double(), want.detach(), rtol=1e-5)
    got.sum().backward()
    assert torch.isfinite(s.grad).all() and torch.isfinite(n.grad).all()
    if mask == "one_cell":                  # (an unreachable frame: autograd of logsumexp over nothing but -inf is NaN in the restatement)
        tril = torch.tril(torch.ones(T, T, dtype=torch.bool)).unsqueeze(-1)
        assert torch.allclose(s.grad.double() * tril, s64.grad * tril, atol=2e-6)
        assert torch.allclose(n.grad.double(), n64.grad, atol=2e-6)


def test_eval_path_gradient_overlapping_intervals_cpu():
    """evalPath is linear in the noise: a gap counts once, minus once per interval covering it -- also when intervals overlap
    (outside evalPath's contract, but the backward must stay the derivative of the forward: ADVICE r3)."""
    from transkun_amd import CRF, synth
    T, B = 7, 2
    score, noise = synth.crf_inputs(T, B, 5, "cpu")
    n = noise.clone().requires_grad_(); s = score.clone().requires_grad_()
    crf = CRF.NeuralSemiCRFInterval(s, n)
    intervals = [[(0, 3), (2, 4)], [(1, 1), (5, 6)]]
    out = crf.evalPath(intervals)
    out.backward(torch.tensor([2.0, -1.0]))
    cover = torch.zeros(T - 1, B)
    for c, lst in enumerate(intervals):
        for b, e in lst:
            cover[b:e, c] += 1
    assert torch.equal(n.grad, (1 - cover) * torch.tensor([2.0, -1.0]))
    # and it is the derivative of the forward
    eps = torch.zeros_like(noise); eps[2, 0] = 1.0
    d = CRF.NeuralSemiCRFInterval(score, noise + eps).evalPath(intervals) - CRF.NeuralSemiCRFInterval(score, noise).evalPath(intervals)
    assert abs(float(d[0]) - float(n.grad[2, 0]) / 2.0) < 1e-5
