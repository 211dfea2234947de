"""NumPy restatement of the TensorFlow BPR-MF model.  TEST INFRASTRUCTURE -- "parity unpinned":
TensorFlow 2.3.2 cannot be installed here, so the TF library semantics below (marked [TF]) are
recalled from its documentation/source, not executed.  Pinned by hand-computed known-answer cases
and an independent autograd derivation in tests/test_oracle_bprmf_batch.py.

Follows elliot/recommender/latent_factor_models/BPRMF_batch/BPRMF_batch_model.py:
  call        :47-55   xui = Bi[item] + sum(Gu[user] * Gi[item], 1)
  train_step  :58-80   difference = clip(xu_pos - xu_neg, -80, 1e8); loss = sum softplus(-difference)
                       + l_w * (l2(gu) + l2(gi) + l2(gj)) + l_b * l2(bi) + l_b * l2(bj) / 10
                       grads -> Adam.apply_gradients
[TF] tf.nn.l2_loss(x) = sum(x**2) / 2; clip_by_value passes gradient where -80 <= x <= 1e8;
[TF] IndexedSlices gradients with duplicate indices are summed before the optimiser;
[TF] Keras Adam sparse apply (beta1 .9, beta2 .999, eps 1e-7): m <- m*b1 (all rows); m[idx] += (1-b1) g;
     v <- v*b2 (all rows); v[idx] += (1-b2) g*g; theta <- theta - lr_t * m / (sqrt(v) + eps) (ALL rows),
     lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t).   
"""
import numpy as np

from . import tf_clauses

BETA1, BETA2, EPS = 0.9, 0.999, 1e-7


def softplus(x):
    return np.logaddexp(0.0, x)


def forward_loss(Gu, Gi, Bi, u, i, j, l_w, l_b, dtype=np.float32):
    """Loss of one batch (:62-75) computed in `dtype` (fp32 = what TF does; fp64 = exact maths)."""
    gu, gi, gj = Gu[u].astype(dtype), Gi[i].astype(dtype), Gi[j].astype(dtype)
    bi, bj = Bi[i].astype(dtype), Bi[j].astype(dtype)
    xui = bi + np.sum(gu * gi, axis=1, dtype=dtype)
    xuj = bj + np.sum(gu * gj, axis=1, dtype=dtype)
    d = np.clip(xui - xuj, dtype(-80.0), dtype(1e8))
    loss = np.sum(softplus(-d).astype(dtype), dtype=dtype)
    l2 = lambda x: np.sum(x * x, dtype=dtype) / dtype(2)
    reg = dtype(l_w) * (l2(gu) + l2(gi) + l2(gj)) + dtype(l_b) * l2(bi) + dtype(l_b) * l2(bj) / dtype(10)
    return dtype(loss + reg)


def gradients(Gu, Gi, Bi, u, i, j, l_w, l_b, dtype=np.float32):
    """Dense gradients of the batch loss w.r.t. (Bi, Gu, Gi) (:77), duplicates summed."""
    gu, gi, gj = Gu[u].astype(dtype), Gi[i].astype(dtype), Gi[j].astype(dtype)
    bi, bj = Bi[i].astype(dtype), Bi[j].astype(dtype)
    xui = bi + np.sum(gu * gi, axis=1, dtype=dtype)
    xuj = bj + np.sum(gu * gj, axis=1, dtype=dtype)
    d = xui - xuj
    inside = (d >= -80.0) if tf_clauses.get("clip_gradient_inclusive_at_bound") else (d > -80.0)      # [TF] clause, oracle/tf_clauses.py
    s = np.where(inside, -1.0 / (1.0 + np.exp(d.astype(np.float64))), 0.0).astype(dtype)
    dGu = np.zeros(Gu.shape, dtype)
    dGi = np.zeros(Gi.shape, dtype)
    dBi = np.zeros(Bi.shape, dtype)
    np.add.at(dGu, u, s[:, None] * (gi - gj) + dtype(l_w) * gu)
    np.add.at(dGi, i, s[:, None] * gu + dtype(l_w) * gi)
    np.add.at(dGi, j, -s[:, None] * gu + dtype(l_w) * gj)
    np.add.at(dBi, i, s + dtype(l_b) * bi)
    np.add.at(dBi, j, -s + dtype(l_b / 10.0) * bj)
    return dBi, dGu, dGi


def adam_lr_t(lr, t):
    b1p = np.power(np.float32(BETA1), np.float32(t))
    b2p = np.power(np.float32(BETA2), np.float32(t))
    return np.float32(lr) * np.sqrt(np.float32(1.0) - b2p) / (np.float32(1.0) - b1p)


def adam_tf_sparse_apply(theta, m, v, g, lr, t):
    """[TF] Keras Adam._resource_apply_sparse on a dense-materialised gradient (zero rows untouched by
    the batch still decay and move).  fp32, in place."""
    f = np.float32
    lr_t = adam_lr_t(lr, t)
    if not tf_clauses.get("adam_sparse_apply_moves_all_rows"):          # [TF] clause switched off: only the touched rows move
        rows = np.flatnonzero(np.any(np.reshape(g, (g.shape[0], -1)) != 0, axis=1))
        return adam_lazy_apply(theta, m, v, g, rows, lr, t)
    m *= f(BETA1)
    m += g * tf_clauses.one_minus(BETA1)
    v *= f(BETA2)
    v += (g * g) * tf_clauses.one_minus(BETA2)
    if tf_clauses.get("adam_epsilon_outside_sqrt_with_folded_bias_correction"):
        theta -= (lr_t * m) / (np.sqrt(v) + f(EPS))
    else:                                                                # "epsilon hat" form
        b1p, b2p = np.power(f(BETA1), f(t)), np.power(f(BETA2), f(t))
        theta -= f(lr) * (m / (f(1) - b1p)) / (np.sqrt(v / (f(1) - b2p)) + f(EPS))


def adam_lazy_apply(theta, m, v, g, rows, lr, t):
    """Touched-rows-only variant (EL_OPT_ADAM_LAZY; NOT the reference's semantics)."""
    f = np.float32
    lr_t = adam_lr_t(lr, t)
    rows = np.unique(rows)
    m[rows] = m[rows] * f(BETA1) + g[rows] * tf_clauses.one_minus(BETA1)
    v[rows] = v[rows] * f(BETA2) + (g[rows] * g[rows]) * tf_clauses.one_minus(BETA2)
    theta[rows] = theta[rows] - (lr_t * m[rows]) / (np.sqrt(v[rows]) + f(EPS))


class BPRMFBatchOracle:
    """BPRMF_batch_model (:18-88) with injected initial weights (TF's GlorotUniform stream cannot be
    reproduced without TF -- SURVEY A.5)."""

    def __init__(self, Gu, Gi, Bi, lr, l_w, l_b, optimizer="adam_tf_dense"):
        self.Gu, self.Gi, self.Bi = (np.array(x, dtype=np.float32, copy=True) for x in (Gu, Gi, Bi))
        self.lr, self.l_w, self.l_b = lr, l_w, l_b
        self.optimizer = optimizer
        self.t = 0
        self.slots = {n: (np.zeros_like(p), np.zeros_like(p)) for n, p in
                      (("Bi", self.Bi), ("Gu", self.Gu), ("Gi", self.Gi))}

    def train_step(self, batch):
        u, i, j = (np.asarray(x).reshape(-1).astype(np.int64) for x in batch)
        loss = forward_loss(self.Gu, self.Gi, self.Bi, u, i, j, self.l_w, self.l_b)
        dBi, dGu, dGi = gradients(self.Gu, self.Gi, self.Bi, u, i, j, self.l_w, self.l_b)
        self.t += 1
        rows = {"Bi": np.concatenate([i, j]), "Gu": u, "Gi": np.concatenate([i, j])}
        for name, theta, g in (("Bi", self.Bi, dBi), ("Gu", self.Gu, dGu), ("Gi", self.Gi, dGi)):
            m, v = self.slots[name]
            if self.optimizer == "adam_tf_dense":
                adam_tf_sparse_apply(theta, m, v, g, self.lr, self.t)
            elif self.optimizer == "adam_lazy":
                adam_lazy_apply(theta, m, v, g, rows[name], self.lr, self.t)
            elif self.optimizer == "sgd":
                theta -= np.float32(self.lr) * g
            else:
                raise ValueError(self.optimizer)
        return float(loss)

    def predict(self, start, stop):
        """:83-84 (NumPy matmul; summation order differs from the pinned fma chain -> use cref for top-k)."""
        return self.Bi + self.Gu[start:stop] @ self.Gi.T
