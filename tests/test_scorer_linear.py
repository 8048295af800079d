"""The scorer's Linear map with its own backward (transkun_amd.scorer._ScorerLinear): pure torch, so it is checked on the CPU
against torch's own nn.Linear autograd (LayersTransformer.py:392-397, :408)."""
import torch
import torch.nn.functional as F


def test_scorer_linear_backward_matches_autograd(monkeypatch):
    import transkun_amd.scorer as sc
    torch.manual_seed(3)
    x = torch.randn(3, 5, 11, 8, dtype=torch.float64, requires_grad=True)
    Wq = torch.randn(6, 8, dtype=torch.float64, requires_grad=True); bq = torch.randn(6, dtype=torch.float64, requires_grad=True)
    Wk = torch.randn(4, 8, dtype=torch.float64, requires_grad=True); bk = torch.randn(4, dtype=torch.float64, requires_grad=True)
    gq = torch.randn(3, 5, 11, 6, dtype=torch.float64); gk = torch.randn(3, 5, 11, 4, dtype=torch.float64)
    for rows in (7, 40, 165, 10 ** 6):          # chunks with a ragged tail, exact chunks, one chunk, no split at all
        monkeypatch.setattr(sc, "SPLITK_ROWS", rows)
        assert torch.autograd.gradcheck(lambda *a: sc._ScorerLinear.apply(*a), (x, Wq, bq, Wk, bk))
        qd, k = sc._ScorerLinear.apply(x, Wq, bq, Wk, bk)
        got = torch.autograd.grad([qd, k], [x, Wq, bq, Wk, bk], [gq, gk])
        want = torch.autograd.grad([F.linear(x, Wq, bq), F.linear(x, Wk, bk)], [x, Wq, bq, Wk, bk], [gq, gk])
        for a, b in zip(got, want):
            assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    # gradients only where they are asked for
    x0 = x.detach()
    qd, k = sc._ScorerLinear.apply(x0, Wq, bq, Wk.detach(), bk)
    (qd.sum() + k.sum()).backward()
    assert Wq.grad is not None and bk.grad is not None
    # one of the two outputs unused
    x1 = x.detach().requires_grad_()
    qd, k = sc._ScorerLinear.apply(x1, Wq, bq, Wk, bk)
    (gx,) = torch.autograd.grad(k.sum(), [x1])
    assert torch.allclose(gx, torch.autograd.grad(F.linear(x1, Wk, bk).sum(), [x1])[0])
