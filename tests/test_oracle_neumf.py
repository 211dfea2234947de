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
    # K.binary_crossentropy as oracle/tf_clauses.py reads it: clip to [1e-7, 1 - 1e-7], epsilon added again inside the logarithms
    p = torch.clamp(torch.sigmoid(logit), 1e-7, 1 - 1e-7)
    return -(y * torch.log(p + 1e-7) + (1 - y) * torch.log(1 - p + 1e-7)).mean()


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
    loss = -(ty * torch.log(p + 1e-7) + (1 - ty) * torch.log(1 - p + 1e-7)).mean()
    loss.backward()
    assert abs(float(loss.detach()) - on.bce(c["p"], y)) < 1e-12
    for k, v in g.items():
        pairs = zip(v, tw[k]) if isinstance(v, list) else [(v, tw[k])]
        for a, b in pairs:
            assert np.abs(a - b.grad.numpy()).max() < 1e-12, k


def test_c_oracle_logits_equal_the_network_in_fp64():
    """oracle/c/el_oracle.c::orc_nmf_logits (the pinned-order checker of el_nmf_score_topk: separable layer 1, k-ordered fma
    chains, two interleaved head chains) is the same function as oracle/neumf.py::forward -- in fp64 maths, to fp32 round-off."""
    from oracle import cref
    for F, units in ((16, None), (9, [36, 18, 9]), (32, [100, 40, 20])):
        U, I = 12, 400
        w = on.init_neumf(U, I, F, 3, units=units)
        rs = np.random.RandomState(F)
        for k in ("Umf", "Imf", "Umlp", "Imlp"):
            w[k] = (w[k] * 5).astype(np.float32)
        w["b"] = [rs.normal(scale=0.1, size=b.shape).astype(np.float32) for b in w["b"]]
        w["hb"] = np.array([-0.2], np.float32)
        L = cref.nmf_logits(w, np.arange(U))
        u, i = np.repeat(np.arange(U), I), np.tile(np.arange(I), U)
        p = on.forward(w, u, i, dtype=np.float64)["p"].reshape(U, I)
        ref = np.log(p) - np.log1p(-p)
        assert np.abs(L - ref).max() < 5e-6 * max(1.0, np.abs(ref).max())
        # a shard of the items and a subset of the users: the same numbers
        L2 = cref.nmf_logits(w, np.array([7, 2]), 100, 250)
        assert np.array_equal(L2, L[[7, 2], 100:250])
        # without the MF branch / without the head bias
        w2 = {k: v for k, v in w.items() if k not in ("Umf", "Imf", "hb")}
        w2["hw"] = w["hw"][F:].copy()
        p2 = on.forward(w2, u, i, dtype=np.float64)["p"].reshape(U, I)
        assert np.abs(cref.nmf_logits(w2, np.arange(U)) - (np.log(p2) - np.log1p(-p2))).max() < 5e-6


def test_gradients_under_a_given_relu_branch_pattern():
    """gradients(relu_masks=...) (the hook tests/test_gpu_neumf.py uses to compare under the DEVICE's audited branch pattern): the
    network's own pattern reproduces the default gradients bit for bit; flipping ONE unit of the top layer to 'on' changes the kernel
    gradient of that layer in that unit's column only -- by exactly that sample's input row times its upstream derivative -- and moves
    every layer below it by one sample's contribution."""
    U, I, F, n = 30, 40, 8, 64
    w = on.init_neumf(U, I, F, 5)
    rs = np.random.RandomState(2)
    w = {k: ([x.astype(np.float64) * 3 for x in v] if isinstance(v, list) else v.astype(np.float64) * 3) for k, v in w.items()}
    w["b"] = [rs.normal(scale=0.1, size=b.shape) for b in w["b"]]
    u, i = rs.randint(0, U, n), rs.randint(0, I, n)
    y = rs.randint(0, 2, n).astype(np.float64)
    c = on.forward(w, u, i, dtype=np.float64)
    g0 = on.gradients(w, c, u, i, y)
    own = [o > 0 for o in c["outs"]]
    g1 = on.gradients(w, c, u, i, y, relu_masks=own)
    for k in ("Umf", "Imf", "Umlp", "Imlp", "hw", "hb"):
        assert np.array_equal(g0[k], g1[k]), k
    for l in range(3):
        assert np.array_equal(g0["W"][l], g1["W"][l]) and np.array_equal(g0["b"][l], g1["b"][l]), l
    # one 'off' unit of the top layer taken as 'on'
    top = len(own) - 1
    rows, cols = np.nonzero(~own[top])
    r, col = int(rows[0]), int(cols[0])
    flipped = [m.copy() for m in own]
    flipped[top][r, col] = True
    g2 = on.gradients(w, c, u, i, y, relu_masks=flipped)
    dW = g2["W"][top] - g0["W"][top]
    other = np.delete(dW, col, axis=1)
    assert np.abs(other).max() == 0.0                                       # only that unit's column moves in its own layer
    p = c["p"][r]
    dlogit = -(y[r] / (p + 1e-7) - (1 - y[r]) / (1 - p + 1e-7)) * p * (1 - p) / n if (1e-7 < p < 1 - 1e-7) else 0.0
    d_unit = dlogit * w["hw"][F + col]                                      # upstream derivative of that unit for that sample
    assert np.allclose(dW[:, col], c["ins"][top][r] * d_unit, rtol=1e-12, atol=1e-18)
    assert np.isclose(g2["b"][top][col] - g0["b"][top][col], d_unit, rtol=1e-12, atol=1e-18)
    if d_unit != 0.0:
        assert np.abs(g2["W"][0] - g0["W"][0]).max() > 0.0                   # ... and every layer below sees one sample's contribution
