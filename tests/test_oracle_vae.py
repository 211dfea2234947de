"""Pins oracle/multi_vae.py: loss/gradients against an independent torch-autograd derivation, Adam step by hand."""
import numpy as np
import torch

from oracle import multi_vae as ov


def test_vae_loss_and_gradients_match_autograd():
    rs = np.random.RandomState(0)
    B, I, H, L = 7, 23, 12, 5
    w = {k: rs.normal(scale=0.3, size=v.shape) for k, v in ov.init_weights(I, H, L, 1).items()}
    x = (rs.rand(B, I) < 0.25).astype(np.float64)
    x[0, :] = 0
    x[0, 3] = 1
    eps = rs.normal(size=(B, L))
    drop = (rs.rand(B, I) > 0.3) / 0.7
    anneal = 0.13
    c = ov.forward(w, x, eps, drop, dtype=np.float64)
    loss = ov.loss_from(c, anneal)
    g = ov.gradients(w, c, anneal)

    tw = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    tx, te, td = torch.tensor(x), torch.tensor(eps), torch.tensor(drop)
    xn = tx / torch.sqrt(torch.clamp((tx * tx).sum(1, keepdim=True), min=1e-12)) * td
    h = torch.tanh(xn @ tw["W1"] + tw["b1"])
    mu, lv = h @ tw["Wm"] + tw["bm"], h @ tw["Wv"] + tw["bv"]
    z = mu + torch.exp(0.5 * lv) * te
    logits = torch.tanh(z @ tw["W3"] + tw["b3"]) @ tw["W4"] + tw["b4"]
    kl = -0.5 * torch.mean(lv - mu ** 2 - torch.exp(lv) + 1)
    tl = -torch.mean(torch.sum(torch.log_softmax(logits, 1) * tx, 1)) + anneal * kl
    tl.backward()
    assert abs(float(tl.detach()) - loss) < 1e-12
    for k in ov.NAMES:
        assert np.abs(g[k] - tw[k].grad.numpy()).max() < 1e-12, k


def test_vae_adam_dense_step_known_answer():
    w = ov.init_weights(6, 4, 2, 3)
    o = ov.MultiVAEOracle(w, lr=0.001)
    x = np.array([[1, 0, 1, 0, 0, 0], [0, 1, 0, 0, 1, 1]], np.float32)
    before = {k: v.copy() for k, v in o.w.items()}
    o.train_step(x, np.zeros((2, 2), np.float32), 0.0)
    # first Adam step moves every element with a non-zero gradient by ~lr * sign(g)
    d = before["W4"] - o.w["W4"]
    nz = np.abs(d) > 0
    assert nz.any() and np.allclose(np.abs(d[nz]), 0.001, rtol=2e-2)
    assert np.array_equal(before["W1"][3], o.w["W1"][3])     # item 3 never appears in the batch -> zero gradient row


def test_dae_loss_and_gradients_match_autograd():
    """oracle/multi_dae.py (Mult-DAE: tanh on the latent layer, no sampling, no KL) against torch autograd."""
    from oracle import multi_dae as od
    rs = np.random.RandomState(1)
    B, I, H, L = 6, 19, 8, 4
    w = {k: rs.normal(scale=0.3, size=v.shape) for k, v in od.init_weights(I, H, L, 1).items()}
    x = (rs.rand(B, I) < 0.3).astype(np.float64)
    x[0, :] = 0
    x[0, 2] = 1
    drop = (rs.rand(B, I) > 0.25) / 0.75
    c = od.forward(w, x, drop, dtype=np.float64)
    loss, g = od.loss_from(c), od.gradients(w, c)
    tw = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    tx, td = torch.tensor(x), torch.tensor(drop)
    xn = tx / torch.sqrt(torch.clamp((tx * tx).sum(1, keepdim=True), min=1e-12)) * td
    z = torch.tanh(torch.tanh(xn @ tw["W1"] + tw["b1"]) @ tw["Wm"] + tw["bm"])
    logits = torch.tanh(z @ tw["W3"] + tw["b3"]) @ tw["W4"] + tw["b4"]
    tl = -torch.mean(torch.sum(torch.log_softmax(logits, 1) * tx, 1))
    tl.backward()
    assert abs(float(tl.detach()) - loss) < 1e-12
    for k in od.NAMES:
        assert np.abs(g[k] - tw[k].grad.numpy()).max() < 1e-12, k
