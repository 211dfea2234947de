"""Pins the NumPy restatement of the TF BPR-MF model (oracle/bprmf_batch.py) with hand-computed
known answers and an independent autograd derivation (TF itself is not installable: SURVEY 8c)."""
import math

import numpy as np
import torch

from oracle import bprmf_batch as ob


def micro():
    Gu = np.array([[1.0, 0.0], [0.0, 1.0]], np.float32)
    Gi = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]], np.float32)
    Bi = np.array([0.1, 0.2, 0.3], np.float32)
    return Gu, Gi, Bi


def test_loss_known_answer():
    Gu, Gi, Bi = micro()
    u, i, j = np.array([0]), np.array([0]), np.array([1])
    l_w, l_b = 0.1, 0.001
    # x_ui = 0.1 + 1, x_uj = 0.2 + 0 -> d = 0.9
    expect = math.log1p(math.exp(-0.9)) + l_w * (0.5 + 0.5 + 0.5) + l_b * 0.5 * 0.01 + l_b * 0.5 * 0.04 / 10
    got = ob.forward_loss(Gu, Gi, Bi, u, i, j, l_w, l_b, dtype=np.float64)
    assert abs(float(got) - expect) < 1e-7  # Bi is stored in fp32
    got32 = ob.forward_loss(Gu, Gi, Bi, u, i, j, l_w, l_b, dtype=np.float32)
    assert abs(float(got32) - expect) < 1e-6


def test_loss_is_batch_sum_and_clips_at_minus_80():
    Gu = np.array([[100.0]], np.float32)
    Gi = np.array([[-1.0], [1.0]], np.float32)
    Bi = np.zeros(2, np.float32)
    u, i, j = np.array([0, 0]), np.array([0, 0]), np.array([1, 1])
    # d = -200 -> clipped to -80 -> softplus(80) = 80 (+2e-35); two identical triplets -> sum
    loss = ob.forward_loss(Gu, Gi, Bi, u, i, j, 0.0, 0.0, dtype=np.float64)
    assert abs(float(loss) - 160.0) < 1e-9
    dBi, dGu, dGi = ob.gradients(Gu, Gi, Bi, u, i, j, 0.0, 0.0, dtype=np.float64)
    assert not dGu.any() and not dGi.any() and not dBi.any()  # clip blocks the gradient below -80


def torch_loss(Gu, Gi, Bi, u, i, j, l_w, l_b):
    gu, gi, gj = Gu[u], Gi[i], Gi[j]
    bi, bj = Bi[i], Bi[j]
    xui = bi + (gu * gi).sum(1)
    xuj = bj + (gu * gj).sum(1)
    d = torch.clamp(xui - xuj, -80.0, 1e8)
    loss = torch.nn.functional.softplus(-d).sum()
    l2 = lambda x: (x * x).sum() / 2
    return loss + l_w * (l2(gu) + l2(gi) + l2(gj)) + l_b * l2(bi) + l_b * l2(bj) / 10


def test_gradients_match_autograd():
    rs = np.random.RandomState(0)
    U, I, F, B = 13, 17, 5, 64
    Gu, Gi, Bi = rs.normal(size=(U, F)), rs.normal(size=(I, F)), rs.normal(size=I)
    u, i, j = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)  # duplicates on purpose
    l_w, l_b = 0.1, 0.01
    dBi, dGu, dGi = ob.gradients(Gu, Gi, Bi, u, i, j, l_w, l_b, dtype=np.float64)
    tGu, tGi, tBi = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (Gu, Gi, Bi))
    loss = torch_loss(tGu, tGi, tBi, torch.tensor(u), torch.tensor(i), torch.tensor(j), l_w, l_b)
    loss.backward()
    assert abs(float(loss.detach()) - float(ob.forward_loss(Gu, Gi, Bi, u, i, j, l_w, l_b, dtype=np.float64))) < 1e-10
    assert np.abs(dGu - tGu.grad.numpy()).max() < 1e-12
    assert np.abs(dGi - tGi.grad.numpy()).max() < 1e-12
    assert np.abs(dBi - tBi.grad.numpy()).max() < 1e-12


def test_adam_tf_sparse_apply_known_answer():
    f = np.float32
    theta = np.array([[1.0, 2.0], [3.0, 4.0]], f)
    m, v = np.zeros_like(theta), np.zeros_like(theta)
    g = np.array([[0.5, -0.25], [0.0, 0.0]], f)   # row 1 untouched
    lr = 0.001
    ob.adam_tf_sparse_apply(theta, m, v, g, lr, 1)
    lr1 = lr * math.sqrt(1 - 0.999) / (1 - 0.9)
    exp00 = 1.0 - lr1 * (0.1 * 0.5) / (math.sqrt(0.001 * 0.25) + 1e-7)
    assert abs(theta[0, 0] - exp00) < 1e-6 and abs(theta[0, 0] - (1.0 - lr)) < 1e-5
    assert theta[1, 0] == f(3.0) and theta[1, 1] == f(4.0)          # m = v = 0 -> no move at t = 1
    # step 2 with a zero gradient everywhere: the previously touched row KEEPS moving (dense semantics)
    before = theta.copy()
    ob.adam_tf_sparse_apply(theta, m, v, np.zeros_like(g), lr, 2)
    assert theta[0, 0] < before[0, 0] and theta[0, 1] > before[0, 1]
    assert np.array_equal(theta[1], before[1])
    lr2 = lr * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    exp = before[0, 0] - lr2 * (0.9 * 0.05) / (math.sqrt(0.999 * 0.00025) + 1e-7)
    assert abs(theta[0, 0] - exp) < 1e-6


def test_train_step_decreases_loss_and_matches_manual_composition():
    rs = np.random.RandomState(1)
    U, I, F, B = 30, 40, 8, 128
    Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
    Bi = np.zeros(I, np.float32)
    u, i, j = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
    o = ob.BPRMFBatchOracle(Gu, Gi, Bi, lr=0.01, l_w=0.1, l_b=0.001)
    l0 = o.train_step((u[:, None], i[:, None], j[:, None]))
    for _ in range(20):
        l1 = o.train_step((u, i, j))
    assert l1 < l0
    # manual first step
    dBi, dGu, dGi = ob.gradients(Gu, Gi, Bi, u, i, j, 0.1, 0.001)
    th, m, v = Gu.copy(), np.zeros_like(Gu), np.zeros_like(Gu)
    ob.adam_tf_sparse_apply(th, m, v, dGu, 0.01, 1)
    o2 = ob.BPRMFBatchOracle(Gu, Gi, Bi, lr=0.01, l_w=0.1, l_b=0.001)
    o2.train_step((u, i, j))
    assert np.array_equal(o2.Gu, th)


def test_postponed_gradient_free_adam_steps_replay_to_the_same_bits():
    """The invariant the device's deferred decay rests on, on the oracle itself: Keras' sparse apply moves
    EVERY row at every step, but for a row without a gradient the update reads nothing except that row -- so postponing it and
    replaying the missed steps later (the same fp32 operations, same order, each step's own lr_t) gives the same bits as moving
    the row at every step.  Eager table against a lazily caught-up one over 30 steps with random touched sets, rows never
    touched, rows touched in bursts."""
    from oracle.bprmf_batch import adam_tf_sparse_apply
    rs = np.random.RandomState(4)
    R, F, T, lr = 300, 12, 30, 0.01
    th0 = rs.standard_normal((R, F)).astype(np.float32)
    eager = [th0.copy(), np.zeros_like(th0), np.zeros_like(th0)]
    lazy = [th0.copy(), np.zeros_like(th0), np.zeros_like(th0)]
    last = np.zeros(R, np.int64)

    def catch_up(rows, upto):
        """rows of the lazy table to step `upto`: every missed gradient-free step, in order, with that step's lr_t"""
        for r in rows:
            for s in range(last[r] + 1, upto + 1):
                z = np.zeros((1, F), np.float32)
                sl = [a[r:r + 1] for a in lazy]
                adam_tf_sparse_apply(sl[0], sl[1], sl[2], z, lr, s)
            last[r] = max(last[r], upto)

    for t in range(1, T + 1):
        n = 0 if t % 7 == 0 else rs.randint(1, 40)
        rows = np.unique(rs.randint(0, R // 2 if t % 2 else R - 20, n))        # the last 20 rows never get a gradient
        g = np.zeros((R, F), np.float32)
        g[rows] = rs.standard_normal((len(rows), F)).astype(np.float32)
        adam_tf_sparse_apply(eager[0], eager[1], eager[2], g, lr, t)          # every row, every step
        catch_up(rows, t - 1)                                                 # the batch's rows to t - 1 ...
        for r in rows:                                                        # ... then step t with their gradient
            sl = [a[r:r + 1] for a in lazy]
            adam_tf_sparse_apply(sl[0], sl[1], sl[2], g[r:r + 1], lr, t)
            last[r] = t
        if t in (11, T):                                                      # a read of the whole table: flush
            catch_up(range(R), t)
            for a, b in zip(eager, lazy):
                assert np.array_equal(a, b), t
