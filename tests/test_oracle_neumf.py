"""Pins oracle/neumf.py (NeuMF + GMF forward / BCE / gradients) against torch autograd."""
import numpy as np
import torch

from oracle import neumf as on


def _torch_loss(w, u, i, y):
    parts = []
    if "Umf" in w:
        parts.append(w["Umf"][u] * w["Imf"][i])
    if "Umlp" in w:
        x = torch.cat([w["Umlp"][u], w["Imlp"][i]], 1)
        for W, b in zip(w["W"], w["b"]):
            x = torch.relu(x @ W + b)
        parts.append(x)
    logit = torch.cat(parts, 1) @ w["hw"] + (w["hb"][0] if "hb" in w else 0)
    p = torch.clamp(torch.sigmoid(logit), 1e-7, 1 - 1e-7)
    return -(y * torch.log(p) + (1 - y) * torch.log(1 - p)).mean()


def _check(w, U, I):
    rs = np.random.RandomState(1)
    n = 50
    u, i = rs.randint(0, U, n), rs.randint(0, I, n)
    y = rs.randint(0, 2, n).astype(np.float64)
    w64 = {k: ([x.astype(np.float64) * 3 for x in v] if isinstance(v, list) else v.astype(np.float64) * 3) for k, v in w.items()}
    if "b" in w64:
        w64["b"] = [rs.normal(scale=0.1, size=b.shape) for b in w64["b"]]
    c = on.forward(w64, u, i, dtype=np.float64)
    g = on.gradients(w64, c, u, i, y)
    tw = {k: ([torch.tensor(x, requires_grad=True) for x in v] if isinstance(v, list) else torch.tensor(v, requires_grad=True))
          for k, v in w64.items()}
    loss = _torch_loss(tw, torch.tensor(u), torch.tensor(i), torch.tensor(y))
    loss.backward()
    assert abs(float(loss.detach()) - on.bce(c["p"], y)) < 1e-12
    for k, v in g.items():
        if isinstance(v, list):
            for a, b in zip(v, tw[k]):
                assert np.abs(a - b.grad.numpy()).max() < 1e-12, k
        else:
            assert np.abs(v - tw[k].grad.numpy()).max() < 1e-12, k


def test_neumf_gradients_match_autograd():
    _check(on.init_neumf(11, 13, 4, 0), 11, 13)


def test_gmf_gradients_match_autograd():
    _check(on.init_gmf(11, 13, 6, 0), 11, 13)


def test_mlp_only_and_mf_only_branches():
    w = on.init_neumf(9, 8, 4, 2)
    mlp_only = {k: v for k, v in w.items() if k not in ("Umf", "Imf")}
    mlp_only["hw"] = w["hw"][4:].copy()
    _check(mlp_only, 9, 8)
    mf_only = {"Umf": w["Umf"], "Imf": w["Imf"], "hw": w["hw"][:4].copy(), "hb": w["hb"]}
    _check(mf_only, 9, 8)
