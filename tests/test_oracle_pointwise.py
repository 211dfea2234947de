"""Pins oracle/pointwise_mf.py (MF / FunkSVD / PMF / LogisticMF losses and gradients) against torch autograd, and its
optimisers against torch.optim on a dense-materialised gradient."""
import numpy as np
import pytest
import torch

from oracle import pointwise_mf as pw


def _weights(rs, U, I, F, bias):
    w = {"Gu": rs.normal(scale=0.4, size=(U, F)), "Gi": rs.normal(scale=0.4, size=(I, F))}
    if bias:
        w["Bu"], w["Bi"] = rs.normal(scale=0.2, size=U), rs.normal(scale=0.2, size=I)
    return w


def _torch_loss(w, kind, u, i, y, alpha, l_w):
    gu, gi = w["Gu"][u], w["Gi"][i]
    x = (gu * gi).sum(-1)
    if "Bu" in w:
        x = x + w["Bu"][u] + w["Bi"][i]
    if kind == "logistic":
        return (-(alpha * y * x - (1 + alpha * y) * torch.log(1 + torch.exp(x)))).sum() \
            + l_w * ((gu ** 2).sum() / 2 + (gi ** 2).sum() / 2)
    o = torch.sigmoid(x) if kind == "mse_sigmoid" else x
    return ((y - o) ** 2).mean()


@pytest.mark.parametrize("kind,bias,alpha,l_w", [("mse", False, 0, 0), ("mse", True, 0, 0), ("mse_sigmoid", False, 0, 0),
                                                  ("logistic", True, 0.5, 0.1), ("logistic", True, 2.0, 0.0)])
def test_loss_and_gradients_match_autograd(kind, bias, alpha, l_w):
    rs = np.random.RandomState(3)
    U, I, F, n = 9, 7, 5, 64                       # n >> U, I: every row is a duplicate-heavy segment
    w = _weights(rs, U, I, F, bias)
    u, i = rs.randint(0, U, n), rs.randint(0, I, n)
    y = rs.randint(0, 2, n).astype(np.float64)
    loss, g = pw.loss_and_grads(w, kind, u, i, y, alpha, l_w, dtype=np.float64)
    tw = {k: torch.tensor(v, requires_grad=True) for k, v in w.items()}
    tl = _torch_loss(tw, kind, torch.tensor(u), torch.tensor(i), torch.tensor(y), alpha, l_w)
    tl.backward()
    assert abs(loss - float(tl.detach())) < 1e-10 * max(1.0, abs(loss))
    for k in w:
        assert np.abs(g[k] - tw[k].grad.numpy()).max() < 1e-12, k


def test_adagrad_matches_torch_on_touched_rows():
    rs = np.random.RandomState(0)
    th = rs.normal(size=(6, 3)).astype(np.float32)
    g = np.zeros_like(th)
    g[[1, 4]] = rs.normal(size=(2, 3))
    ref = torch.tensor(th.copy(), requires_grad=True)
    opt = torch.optim.Adagrad([ref], lr=0.05, initial_accumulator_value=0.1, eps=1e-7)
    acc = np.full_like(th, 0.1)
    mine = th.copy()
    for _ in range(3):
        ref.grad = torch.tensor(g)
        opt.step()
        pw.adagrad_apply(mine, acc, g, 0.05)
    assert np.abs(mine - ref.detach().numpy()).max() < 1e-6
    assert np.array_equal(mine[[0, 2, 3, 5]], th[[0, 2, 3, 5]])


def test_oracle_steps_reduce_the_loss():
    rs = np.random.RandomState(5)
    U, I, F = 30, 40, 8
    for kind, bias, opt in (("mse", False, "adam"), ("mse", True, "adam"), ("mse_sigmoid", False, "adam"), ("logistic", True, "adagrad")):
        w = _weights(rs, U, I, F, bias)
        o = pw.PointwiseOracle(w, kind, 0.05, optimizer=opt, alpha=0.5, l_w=0.01)
        u, i = rs.randint(0, U, 256), rs.randint(0, I, 256)
        y = ((u + i) % 2).astype(np.float32)
        first = o.train_step((u, i, y))
        for _ in range(30):
            last = o.train_step((u, i, y))
        assert last < first, (kind, first, last)
