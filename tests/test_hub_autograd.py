"""Host logic of the two-node pattern (crf.evalPath(...) - crf.computeLogZ(), ModelTransformer.py:263-265) on the CPU: the
raw kernel calls of the mirror are replaced by the CPU oracle (test infrastructure), everything else -- the private hub
node, the owned gradient accumulator, parked evalPath scatters -- is the product's Python and runs as it does on the GPU.
Checked against plain autograd of a dense formula, for every order and number of consumers of `score`."""
import importlib

import numpy as np
import pytest
import torch

from oracle import oracle as cpu_oracle


@pytest.fixture()
def nsci(monkeypatch):
    m = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")

    def logz_fwd(score, noise, want_v):
        logz, _, _, v, _ = cpu_oracle.forward_backward(score.numpy(), noise.numpy())
        return torch.from_numpy(np.asarray(logz, np.float32)), (torch.from_numpy(np.asarray(v, np.float32)) if want_v else None)

    def logz_bwd(score, noise, v, logz, gout, want_q=False):
        _, grad, gn, _, _ = cpu_oracle.forward_backward(score.numpy(), noise.numpy())
        g = gout.numpy()
        return torch.from_numpy((grad * g).astype(np.float32)), torch.from_numpy((gn * g).astype(np.float32)), None

    def eval_path(score, noise, pairs, offsets):
        iv = [[tuple(map(int, pairs[i])) for i in range(int(offsets[c]), int(offsets[c + 1]))] for c in range(score.shape[2])]
        return torch.from_numpy(np.asarray(cpu_oracle.eval_path(iv, score.numpy(), noise.numpy()), np.float32))

    def eval_path_bwd(g, T, B, pairs, offsets, dscore, dnoise, K, pooled=False):
        for c in range(B):
            covered = np.zeros(max(T - 1, 0), bool)
            for i in range(int(offsets[c]), int(offsets[c + 1])):
                b, e = int(pairs[i, 0]), int(pairs[i, 1])
                if dscore is not None:
                    dscore[e, b, c] += g[c]
                covered[b:e] = True
            if dnoise is not None and T > 1:
                dnoise[:, c] += g[c] * torch.from_numpy(~covered).float()

    def pack(intervals, T, B, device, overlap=False):
        flat = [p for lst in intervals for p in lst]
        pairs = torch.tensor(flat if flat else [(0, 0)], dtype=torch.int32).view(-1, 2)
        offsets = torch.tensor(np.concatenate([[0], np.cumsum([len(x) for x in intervals])]), dtype=torch.int32)
        pairs._semicrf_K = len(flat)
        return pairs, offsets

    monkeypatch.setattr(m, "_logz_fwd_raw", logz_fwd)
    monkeypatch.setattr(m, "_logz_bwd_raw", logz_bwd)
    monkeypatch.setattr(m, "_eval_path_raw", eval_path)
    monkeypatch.setattr(m, "_eval_path_bwd_raw", eval_path_bwd)
    monkeypatch.setattr(m, "pack_intervals", pack)
    monkeypatch.setattr(m._lib, "require_gpu", lambda t, name: None)
    return m


def _case():
    g = torch.Generator().manual_seed(5)
    T, B = 12, 3
    score = torch.randn(T, T, B, generator=g)
    noise = torch.randn(T - 1, B, generator=g)
    iv = [[(0, 2), (2, 2), (4, 7)], [], [(1, 1), (3, 10)]]
    w = torch.randn(B, generator=g)
    r = torch.randn(T, T, B, generator=g)
    return T, B, score, noise, iv, w, r


def _want(score, noise, iv, w, r):
    logz, grad, gn, _, _ = cpu_oracle.forward_backward(score.numpy(), noise.numpy())
    T, B = score.shape[0], score.shape[2]
    ds = -(grad * w.numpy()).astype(np.float64)
    dn = -(gn * w.numpy()).astype(np.float64)
    for c, lst in enumerate(iv):
        cov = np.zeros(T - 1, bool)
        for b, e in lst:
            ds[e, b, c] += float(w[c]); cov[b:e] = True
        dn[:, c] += float(w[c]) * (~cov)
    return ds + r.numpy(), dn


@pytest.mark.parametrize("order", ["path_first", "logz_first", "reg_before", "reg_after", "two_logz", "path_only_twice"])
def test_hub_collects_every_order(nsci, order):
    T, B, score, noise, iv, w, r = _case()
    s = score.clone().requires_grad_(); n = noise.clone().requires_grad_()
    want_s, want_n = _want(score, noise, iv, w, r)
    if order == "reg_before":
        reg = (s * r).sum()
    c = nsci.NeuralSemiCRFInterval(s, n)
    if order in ("path_first", "reg_after", "reg_before"):
        lp = c.evalPath(iv) - c.computeLogZ()
    elif order == "logz_first":
        lz = c.computeLogZ(); lp = c.evalPath(iv) - lz
    elif order == "two_logz":
        lp = c.evalPath(iv) - 0.5 * c.computeLogZ() - 0.5 * c.computeLogZ(noBackward=True)
    else:
        lp = 0.5 * c.evalPath(iv) + 0.5 * c.evalPath(iv) - c.computeLogZ()
    if order != "reg_before":
        reg = (s * r).sum()
    ((lp * w).sum() + reg).backward()
    assert np.abs(s.grad.numpy() - want_s).max() < 1e-4
    assert np.abs(n.grad.numpy() - want_n).max() < 1e-4
    assert not c._hub[4].acc, "the hub's accumulator of a finished pass must be gone"


def test_hub_path_only_and_retained_graph(nsci):
    T, B, score, noise, iv, w, r = _case()
    s = score.clone().requires_grad_(); n = noise.clone().requires_grad_()
    c = nsci.NeuralSemiCRFInterval(s, n)
    p = (c.evalPath(iv) * w).sum()
    g1 = torch.autograd.grad(p, [s, n], retain_graph=True)
    g2 = torch.autograd.grad(p, [s, n])
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    assert float(g1[0].sum()) == pytest.approx(float(sum(w[k] * len(iv[k]) for k in range(B))), rel=1e-6)
    # the module-level functions (no object, no hub) differentiate the plain way and agree
    s2 = score.clone().requires_grad_(); n2 = noise.clone().requires_grad_()
    (nsci.evalPath(iv, s2, n2) * w).sum().backward()
    assert torch.equal(s2.grad, g1[0]) and torch.equal(n2.grad, g1[1])


def test_no_hub_without_grad(nsci):
    T, B, score, noise, iv, w, r = _case()
    c = nsci.NeuralSemiCRFInterval(score, noise)
    with torch.no_grad():
        lp = c.evalPath(iv) - c.computeLogZ()
    assert c._hub is None and not lp.requires_grad
