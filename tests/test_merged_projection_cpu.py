"""The merged projection's algebra on the CPU in float64 (no kernel involved): transkun_amd.fused.merged_weights turns the reference
scorer's two projections (LayersTransformer.py:392-397, 406-410) into one; scores and parameter gradients must equal the
two-projection formula's."""
import pytest
import torch


@pytest.mark.parametrize("size,exp", [(64, 1), (32, 2), (48, 1)])
def test_merged_weights_equal_two_projections(size, exp):
    from transkun_amd.fused import merged_weights
    from transkun_amd.scorer import QPAD
    torch.manual_seed(size + exp)
    D = size * exp
    W = torch.randn(2 * D + 1, size, dtype=torch.float64, requires_grad=True)
    b = torch.randn(2 * D + 1, dtype=torch.float64, requires_grad=True)
    x = torch.randn(7, size, dtype=torch.float64)
    Wm, bm = merged_weights(W, b, D)
    assert Wm.shape == (size + QPAD, size) and bm.shape == (size + QPAD,)
    zc = x @ Wm.t() + bm
    q = x @ W[:D].t() + b[:D]
    k = x @ W[D:2 * D].t() + b[D:2 * D]
    S_merged = zc[:, :size] @ x.t() + zc[:, size:size + 1]           # <z_e, x_b> + c_e
    S_two = q @ k.t()
    assert float((S_merged - S_two).abs().max()) < 1e-10
    assert float((zc[:, size + 1] - (x @ W[2 * D] + b[2 * D])).abs().max()) < 1e-12     # the diagonal term's column
    assert float(zc[:, size + 2:].abs().max()) == 0.0                                   # padding columns
    g = torch.randn_like(S_two)
    gm = torch.autograd.grad((S_merged * g).sum(), (W, b), retain_graph=True)
    gt = torch.autograd.grad((S_two * g).sum(), (W, b), allow_unused=True)
    assert float((gm[0] - gt[0]).abs().max()) < 1e-9 and float((gm[1] - gt[1]).abs().max()) < 1e-9


def test_mm_blocks_is_a_matrix_product():
    from transkun_amd.fused import _mm_blocks
    torch.manual_seed(0)
    a = torch.randn(33, 48, dtype=torch.float64, requires_grad=True)
    b = torch.randn(48, 64, dtype=torch.float64, requires_grad=True)
    out = _mm_blocks(a, b)
    assert float((out - a @ b).abs().max()) < 1e-12
    g = torch.randn_like(out)
    ga = torch.autograd.grad((out * g).sum(), (a, b))
    gb = torch.autograd.grad(((a @ b) * g).sum(), (a, b))
    assert float((ga[0] - gb[0]).abs().max()) < 1e-12 and float((ga[1] - gb[1]).abs().max()) < 1e-12
    c = torch.randn(48, 50, dtype=torch.float64)                      # a width the blocks do not divide: plain product
    assert float((_mm_blocks(a, c) - a @ c).abs().max()) < 1e-12
