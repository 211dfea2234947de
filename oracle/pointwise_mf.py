"""NumPy restatement of the point-wise factor models MF, PMF, FunkSVD and LogisticMF.  TEST INFRASTRUCTURE -- "parity
unpinned" vs TensorFlow (not installable here); losses and gradients pinned by torch autograd
(tests/test_oracle_pointwise.py).

Follows (all under elliot/recommender/latent_factor_models/)
  MF/matrix_factorization_model.py:52-71            x = <U[u], I[i]>;                loss = mean (y - x)^2
  FunkSVD/funk_svd_model.py:62-85                   x = <U[u], I[i]> + (bu[u] + bi[i]); same loss
  PMF/probabilistic_matrix_factorization_model.py:60-89   o = sigmoid(<U[u], I[i]>); loss = mean (y - o)^2
        [TF] `self.noise(...)` (GaussianNoise) is called without training=True from a custom train_step, so Keras
        resolves `training` to the learning phase (0): the layer is the identity.  Not restated.
  LogisticMF/logistic_matrix_factorization_model.py:52-85
        x = <Gu[u], Gi[i]> + Bu[u] + Bi[i]
        loss = sum -(alpha y x - (1 + alpha y) log(1 + exp(x))) + l_w (l2_loss(Gu[u]) + l2_loss(Gi[i])),  l2_loss = sum(.^2)/2
        over the GATHERED rows (a row sampled c times is penalised c times); Adagrad on (Gi, Bi) or (Gu, Bu).
        [TF] the `_user_update` flag is a Python attribute read at trace time inside @tf.function (:45-47,74-79): which
        side a traced graph updates depends on when it was traced.  Restated here is the documented intent
        (logistic_matrix_factorization.py:96-110): one pass updating the items, one pass updating the users.
[TF] keras.losses.MeanSquaredError(): labels cast to float, mean over the batch.  The `embeddings_regularizer`s only
     add to `model.losses`, which these train_steps never read: no L2 term for MF / PMF / FunkSVD.
[TF] Embedding gradients are IndexedSlices, duplicates summed, Adam sparse apply (every row decays);
     Adagrad sparse apply touches the sampled rows only: acc += g^2; theta -= lr g / (sqrt(acc) + 1e-7), acc0 = 0.1.
"""
import numpy as np

from .bprmf_batch import adam_tf_sparse_apply

KINDS = ("mse", "mse_sigmoid", "logistic")


def sigmoid(x):
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-x))


def softplus(x):
    return np.where(x > 15, x + np.exp(-np.abs(x)), np.log1p(np.exp(np.minimum(x, 15))))


def raw_score(w, u, i, dtype=np.float32):
    f = lambda a: np.asarray(a, dtype=dtype)
    x = np.sum(f(w["Gu"])[u] * f(w["Gi"])[i], axis=-1)
    if "Bu" in w:
        x = x + (f(w["Bu"])[u] + f(w["Bi"])[i])
    return x


def predict(w, kind, u, i, dtype=np.float32):
    x = raw_score(w, u, i, dtype)
    return sigmoid(x) if kind == "mse_sigmoid" else x


def loss_and_grads(w, kind, u, i, y, alpha=0.0, l_w=0.0, dtype=np.float32):
    """-> (loss, {name: dense gradient}) for every variable of w."""
    f = lambda a: np.asarray(a, dtype=dtype)
    Gu, Gi = f(w["Gu"]), f(w["Gi"])
    y = f(y)
    n = len(y)
    x = raw_score(w, u, i, dtype)
    if kind == "logistic":
        wgt = 1 + alpha * y
        loss = np.sum(wgt * softplus(x) - alpha * y * x) + 0.5 * l_w * (np.sum(Gu[u] ** 2) + np.sum(Gi[i] ** 2))
        c = wgt * sigmoid(x) - alpha * y
    else:
        o = sigmoid(x) if kind == "mse_sigmoid" else x
        loss = np.mean((y - o) ** 2)
        c = 2 * (o - y) / n
        if kind == "mse_sigmoid":
            c = c * o * (1 - o)
    c = c.astype(dtype)
    g = {"Gu": np.zeros_like(Gu), "Gi": np.zeros_like(Gi)}
    np.add.at(g["Gu"], u, c[:, None] * Gi[i])
    np.add.at(g["Gi"], i, c[:, None] * Gu[u])
    if kind == "logistic" and l_w:
        np.add.at(g["Gu"], u, dtype(l_w) * Gu[u])
        np.add.at(g["Gi"], i, dtype(l_w) * Gi[i])
    if "Bu" in w:
        g["Bu"], g["Bi"] = np.zeros_like(f(w["Bu"])), np.zeros_like(f(w["Bi"]))
        np.add.at(g["Bu"], u, c)
        np.add.at(g["Bi"], i, c)
    return float(loss), g


def adagrad_apply(theta, acc, g, lr, eps=1e-7):
    f = np.float32
    hit = g != 0
    acc[hit] += g[hit] * g[hit]
    theta[hit] -= f(lr) * g[hit] / (np.sqrt(acc[hit]) + f(eps))


class PointwiseOracle:
    """train_step / predict of one model with injected initial weights (fp32 state, like the TF variables)."""

    def __init__(self, weights, kind, lr, optimizer="adam", alpha=0.0, l_w=0.0):
        assert kind in KINDS
        self.w = {k: np.array(v, dtype=np.float32).reshape(-1) if k in ("Bu", "Bi") else np.array(v, dtype=np.float32)
                  for k, v in weights.items()}
        self.kind, self.lr, self.optimizer, self.alpha, self.l_w = kind, lr, optimizer, alpha, l_w
        init = 0.0 if optimizer == "adam" else 0.1
        self.m = {k: np.full_like(v, init) for k, v in self.w.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.t = 0

    def train_step(self, batch, side="both"):
        u, i, y = (np.asarray(a).reshape(-1) for a in batch)
        loss, g = loss_and_grads(self.w, self.kind, u, i, y, self.alpha, self.l_w)
        self.t += 1
        names = {"both": ("Gu", "Bu", "Gi", "Bi"), "items": ("Gi", "Bi"), "users": ("Gu", "Bu")}[side]
        for k in names:
            if k not in self.w:
                continue
            if self.optimizer == "adam":
                adam_tf_sparse_apply(self.w[k], self.m[k], self.v[k], g[k].astype(np.float32), self.lr, self.t)
            else:
                adagrad_apply(self.w[k], self.m[k], g[k].astype(np.float32), self.lr)
        return loss

    def predict_all(self, start, stop):
        """[stop-start, I] model scores (get_recs on the full grid / LogisticMF.predict_batch)."""
        x = self.w["Gu"][start:stop] @ self.w["Gi"].T
        if "Bu" in self.w:
            x = x + (self.w["Bu"][start:stop, None] + self.w["Bi"][None, :])
        return sigmoid(x) if self.kind == "mse_sigmoid" else x
