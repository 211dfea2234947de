"""NumPy restatement of Collaborative Metric Learning as the reference computes it.  TEST INFRASTRUCTURE -- "parity unpinned"
vs TensorFlow (not installable here); loss and gradients pinned by torch autograd on the same broadcast expression
(tests/test_oracle_cml.py).

Follows elliot/recommender/latent_factor_models/CML/CML_model.py
  :58-67  call: beta_i = squeeze(Bi(item)) [B]; gamma_* = squeeze(G*(.)) [B,F];
          l2 = reduce_sum(square(gamma_u - gamma_i), -1, keepdims=True) [B,1];  score = -l2 + beta_i  -> BROADCAST [B,B]
  :69-95  train_step: difference = clip(xu_pos - xu_neg, -80, 1e8) [B,B]; loss = sum(max(margin - difference, 0))
          + l_w * sum(l2_loss(gamma_u), l2_loss(gamma_pos), l2_loss(gamma_neg)) + l_b*l2_loss(beta_pos) + l_b*l2_loss(beta_neg)/10
          Adam on [Gu, Gi, Bi] (embedding variables -> sparse apply, every row decays)
  :97-102 predict: -sum_f (Gu[u,f] - Gi[i,f])^2 + Bi[i]
[TF] LatentFactor = Embedding(initializer 'uniform' = RandomUniform(-0.05, 0.05)), also for the [I,1] bias table (:134-145).
[TF] gradient conventions on the measure-zero boundaries (maximum at equality, clip at its ends) follow
     tf.maximum / tf.clip_by_value: pass-through on >= / inside-or-equal.
"""
import numpy as np

from .bprmf_batch import adam_tf_sparse_apply


def broadcast_difference(Gu, Gi, Bi, u, i, j):
    """The [B,B] matrix the reference feeds its hinge: diff[a,b] = (-d+_a + b(i_b)) - (-d-_a + b(j_b))."""
    dpos = np.sum((Gu[u] - Gi[i]) ** 2, axis=-1, keepdims=True)
    dneg = np.sum((Gu[u] - Gi[j]) ** 2, axis=-1, keepdims=True)
    return (-dpos + Bi[i][None, :]) - (-dneg + Bi[j][None, :])


def loss_and_grads(Gu, Gi, Bi, u, i, j, l_w, l_b, margin, dtype=np.float32):
    f = lambda a: np.asarray(a, dtype=dtype)
    Gu, Gi, Bi = f(Gu), f(Gi), f(Bi)
    diff = broadcast_difference(Gu, Gi, Bi, u, i, j)
    clipped = np.clip(diff, -80.0, 1e8)
    hinge = margin - clipped
    loss = np.sum(np.maximum(hinge, 0)) + l_w * 0.5 * (np.sum(Gu[u] ** 2) + np.sum(Gi[i] ** 2) + np.sum(Gi[j] ** 2)) \
        + l_b * 0.5 * np.sum(Bi[i] ** 2) + l_b * 0.5 * np.sum(Bi[j] ** 2) / 10
    g = -((hinge >= 0) & (diff >= -80.0) & (diff <= 1e8)).astype(dtype)     # dloss / ddiff[a,b]
    cD, cE = g.sum(1), g.sum(0)                                             # diff = D_a + E_b
    dGu, dGi, dBi = np.zeros_like(Gu), np.zeros_like(Gi), np.zeros_like(Bi)
    np.add.at(dGu, u, (2 * cD)[:, None] * (Gi[i] - Gi[j]) + dtype(l_w) * Gu[u])
    np.add.at(dGi, i, (2 * cD)[:, None] * (Gu[u] - Gi[i]) + dtype(l_w) * Gi[i])
    np.add.at(dGi, j, -(2 * cD)[:, None] * (Gu[u] - Gi[j]) + dtype(l_w) * Gi[j])
    np.add.at(dBi, i, cE + dtype(l_b) * Bi[i])
    np.add.at(dBi, j, -cE + dtype(l_b) / 10 * Bi[j])
    return float(loss), dGu, dGi, dBi


class CMLOracle:
    def __init__(self, Gu, Gi, Bi, lr, l_w, l_b, margin):
        self.Gu, self.Gi, self.Bi = (np.array(x, dtype=np.float32) for x in (Gu, Gi, Bi))
        self.lr, self.l_w, self.l_b, self.margin = lr, l_w, l_b, margin
        self.m = [np.zeros_like(x) for x in (self.Gu, self.Gi, self.Bi)]
        self.v = [np.zeros_like(x) for x in (self.Gu, self.Gi, self.Bi)]
        self.t = 0

    def train_step(self, batch):
        u, i, j = (np.asarray(x).reshape(-1) for x in batch)
        loss, *grads = loss_and_grads(self.Gu, self.Gi, self.Bi, u, i, j, self.l_w, self.l_b, self.margin)
        self.t += 1
        for th, m, v, g in zip((self.Gu, self.Gi, self.Bi), self.m, self.v, grads):
            adam_tf_sparse_apply(th, m, v, g.astype(np.float32), self.lr, self.t)
        return loss

    def predict(self, start, stop):
        d = self.Gu[start:stop, None, :] - self.Gi[None, :, :]
        return -np.sum(d * d, axis=-1) + self.Bi[None, :]


# ---- the separable form the device evaluates (el_cml.hip): used by the multi-rank tests, pinned against loss_and_grads -------
def distances(Gu, Gi, Bi, u, i, j, dtype=np.float32):
    f = lambda a: np.asarray(a, dtype=dtype)
    Gu, Gi, Bi = f(Gu), f(Gi), f(Bi)
    D = np.sum((Gu[u] - Gi[j]) ** 2, -1) - np.sum((Gu[u] - Gi[i]) ** 2, -1)
    return D, Bi[i] - Bi[j]


def coefficients(D, E, D_all, E_all, margin):
    """cD_a = -#{b : -80 - D_a <= E_b <= margin - D_a}, cE_a = -#{b : -80 - E_a <= D_b <= margin - E_a} over the GLOBAL batch,
    and the local triplets' share of the hinge sum."""
    Es, Ds = np.sort(np.asarray(E_all, np.float64)), np.sort(np.asarray(D_all, np.float64))
    D64, E64 = np.asarray(D, np.float64), np.asarray(E, np.float64)
    low = np.searchsorted(Es, -80.0 - D64, side="left")
    n_a = np.searchsorted(Es, margin - D64, side="right") - low
    m_a = np.searchsorted(Ds, margin - E64, side="right") - np.searchsorted(Ds, -80.0 - E64, side="left")
    hinge = float(np.sum(n_a * (margin - D64)) - np.sum(m_a * E64) + np.sum(low) * (margin + 80.0))
    return -n_a.astype(np.float64), -m_a.astype(np.float64), hinge


def row_gradients(Gu, Gi, Bi, u, i, j, cD, cE, l_w, l_b, dtype=np.float32):
    f = lambda a: np.asarray(a, dtype=dtype)
    Gu, Gi, Bi, cD, cE = f(Gu), f(Gi), f(Bi), f(cD), f(cE)
    dGu, dGi, dBi = np.zeros_like(Gu), np.zeros_like(Gi), np.zeros_like(Bi)
    np.add.at(dGu, u, (2 * cD)[:, None] * (Gi[i] - Gi[j]) + dtype(l_w) * Gu[u])
    np.add.at(dGi, i, (2 * cD)[:, None] * (Gu[u] - Gi[i]) + dtype(l_w) * Gi[i])
    np.add.at(dGi, j, -(2 * cD)[:, None] * (Gu[u] - Gi[j]) + dtype(l_w) * Gi[j])
    np.add.at(dBi, i, cE + dtype(l_b) * Bi[i])
    np.add.at(dBi, j, -cE + dtype(l_b) / 10 * Bi[j])
    return dGu, dGi, dBi


def regulariser(Gu, Gi, Bi, u, i, j, l_w, l_b):
    return float(l_w * 0.5 * (np.sum(Gu[u] ** 2) + np.sum(Gi[i] ** 2) + np.sum(Gi[j] ** 2)) + l_b * 0.5 * np.sum(Bi[i] ** 2)
                 + l_b * 0.5 * np.sum(Bi[j] ** 2) / 10)
