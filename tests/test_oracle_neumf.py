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


def test_dropout_masks_enter_forward_and_backward_like_autograd():
    """Dropout in front of every Dense (neural_matrix_factorization_model.py:58-61): x * mask with mask in {0, 1/(1-rate)}."""
    rs = np.random.RandomState(4)
    U, I, F, n = 9, 11, 4, 30
    w = on.init_neumf(U, I, F, 1)
    u, i = rs.randint(0, U, n), rs.randint(0, I, n)
    y = rs.randint(0, 2, n).astype(np.float64)
    masks = on.dropout_masks(n, [2 * F, 4 * F, 2 * F], 0.4, 42, 3)
    assert [m.shape for m in masks] == [(n, 2 * F), (n, 4 * F), (n, 2 * F)]
    for m in masks:
        assert set(np.unique(m)) <= {np.float32(0), np.float32(1) / (np.float32(1) - np.float32(0.4))}
        assert 0.25 < (m == 0).mean() < 0.55
    w64 = {k: ([x.astype(np.float64) * 3 for x in v] if isinstance(v, list) else v.astype(np.float64) * 3) for k, v in w.items()}
    c = on.forward(w64, u, i, dtype=np.float64, masks=masks)
    g = on.gradients(w64, c, u, i, y)
    tw = {k: ([torch.tensor(x, requires_grad=True) for x in v] if isinstance(v, list) else torch.tensor(v, requires_grad=True))
          for k, v in w64.items()}
    tu, ti, ty = torch.tensor(u), torch.tensor(i), torch.tensor(y)
    x = torch.cat([tw["Umlp"][tu], tw["Imlp"][ti]], 1)
    for l, (W, b) in enumerate(zip(tw["W"], tw["b"])):
        x = torch.relu((x * torch.tensor(masks[l].astype(np.float64))) @ W + b)
    logit = torch.cat([tw["Umf"][tu] * tw["Imf"][ti], x], 1) @ tw["hw"] + tw["hb"][0]
    p = torch.clamp(torch.sigmoid(logit), 1e-7, 1 - 1e-7)
    loss = -(ty * torch.log(p) + (1 - ty) * torch.log(1 - p)).mean()
    loss.backward()
    assert abs(float(loss.detach()) - on.bce(c["p"], y)) < 1e-12
    for k, v in g.items():
        pairs = zip(v, tw[k]) if isinstance(v, list) else [(v, tw[k])]
        for a, b in pairs:
            assert np.abs(a - b.grad.numpy()).max() < 1e-12, k
