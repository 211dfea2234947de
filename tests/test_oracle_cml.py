"""Pins oracle/cml.py against torch autograd on the reference's own (broadcasting) expression."""
import numpy as np
import torch

from oracle import cml


def test_loss_and_gradients_match_autograd_on_the_broadcast_expression():
    rs = np.random.RandomState(2)
    U, I, F, B = 12, 15, 6, 40
    Gu, Gi = rs.normal(scale=0.5, size=(U, F)), rs.normal(scale=0.5, size=(I, F))
    Bi = rs.normal(scale=0.3, size=I)
    u, i, j = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
    l_w, l_b, margin = 0.01, 0.02, 0.5
    loss, dGu, dGi, dBi = cml.loss_and_grads(Gu, Gi, Bi, u, i, j, l_w, l_b, margin, dtype=np.float64)
    tGu, tGi, tBi = (torch.tensor(x, requires_grad=True) for x in (Gu, Gi, Bi))
    tu, ti, tj = (torch.tensor(x).reshape(-1, 1) for x in (u, i, j))                 # [B,1] like the sampler's batches

    def call(item):                                                                   # CML_model.call :58-67
        beta = tBi[item].squeeze()                                                    # [B]
        gu, gi = tGu[tu].squeeze(), tGi[item].squeeze()                               # [B,F]
        l2 = ((gu - gi) ** 2).sum(-1, keepdim=True)                                   # [B,1]
        return -l2 + beta, beta, gu, gi                                               # [B,B]
    xp, bp, gu, gp = call(ti)
    xn, bn, _, gn = call(tj)
    assert xp.shape == (B, B)
    diff = torch.clamp(xp - xn, -80.0, 1e8)
    tl = torch.clamp(margin - diff, min=0).sum() + l_w * ((gu ** 2).sum() / 2 + (gp ** 2).sum() / 2 + (gn ** 2).sum() / 2) \
        + l_b * (bp ** 2).sum() / 2 + l_b * ((bn ** 2).sum() / 2) / 10
    tl.backward()
    assert abs(loss - float(tl.detach())) < 1e-9 * abs(loss)
    for mine, ref in ((dGu, tGu), (dGi, tGi), (dBi, tBi)):
        assert np.abs(mine - ref.grad.numpy()).max() < 1e-10


def test_clip_region_contributes_constant_terms():
    Gu, Gi = np.array([[10.0, 0.0]]), np.array([[10.0, 0.0], [0.0, 0.0]])             # d+ = 0, d- = 100 -> D = 100
    Bi = np.zeros(2)
    loss, dGu, dGi, dBi = cml.loss_and_grads(Gu, Gi, Bi, np.array([0]), np.array([0]), np.array([1]), 0.0, 0.0, 0.5, dtype=np.float64)
    assert loss == 0.0 and not dGu.any()                                              # diff = 100 > margin: inactive
    loss, dGu, *_ = cml.loss_and_grads(Gu, Gi, Bi, np.array([0]), np.array([1]), np.array([0]), 0.0, 0.0, 0.5, dtype=np.float64)
    assert loss == 80.5 and not dGu.any()                                             # diff = -100 < -80: clipped, constant


def test_separable_form_equals_the_matrix_form():
    """What the device evaluates (sorted counts) against the [B,B] matrix restatement, incl. pairs beyond the -80 clip."""
    rs = np.random.RandomState(5)
    U, I, F, B = 15, 12, 4, 60
    for scale in (0.5, 5.0):
        Gu, Gi, Bi = rs.normal(scale=scale, size=(U, F)), rs.normal(scale=scale, size=(I, F)), rs.normal(scale=0.4, size=I)
        u, i, j = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
        loss, dGu, dGi, dBi = cml.loss_and_grads(Gu, Gi, Bi, u, i, j, 0.01, 0.02, 0.5, dtype=np.float64)
        D, E = cml.distances(Gu, Gi, Bi, u, i, j, dtype=np.float64)
        cD, cE, hinge = cml.coefficients(D, E, D, E, 0.5)
        g = cml.row_gradients(Gu, Gi, Bi, u, i, j, cD, cE, 0.01, 0.02, dtype=np.float64)
        assert abs(hinge + cml.regulariser(Gu, Gi, Bi, u, i, j, 0.01, 0.02) - loss) < 1e-8 * abs(loss)
        for a, b in zip(g, (dGu, dGi, dBi)):
            assert np.abs(a - b).max() < 1e-9
