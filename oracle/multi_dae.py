"""NumPy restatement of the Mult-DAE model.  TEST INFRASTRUCTURE -- "parity unpinned" vs TensorFlow (not installable
here); gradients are pinned by an independent torch-autograd derivation (tests/test_oracle_vae.py).

Follows elliot/recommender/autoencoders/dae/multi_dae_model.py:
  Encoder.call   :44-51   l2_normalize(axis=1) -> Dropout -> Dense(intermediate, tanh) -> Dense(latent, tanh)
  Decoder.call   :69-72   Dense(intermediate, tanh) -> Dense(original_dim)
  train_step     :114-127 loss = -mean_b sum_i log_softmax(logits) x   (no KL term, no sampling) ; Adam
  predict        :129-139 log_softmax(logits)
The declared kernel_regularizers are never added to the loss, as in the VAE.  Arithmetic conventions ([TF] notes) are those
of oracle/multi_vae.py, whose helpers are reused.
"""
import numpy as np

from . import tf_clauses

from .multi_vae import BETA1, BETA2, EPS, adam_lr_t, glorot_normal, log_softmax

NAMES = ("W1", "b1", "Wm", "bm", "W3", "b3", "W4", "b4")


def init_weights(n_items, hidden, latent, seed):
    rs = np.random.RandomState(seed)
    z = lambda n: np.zeros(n, np.float32)
    return {"W1": glorot_normal(rs, n_items, hidden), "b1": z(hidden), "Wm": glorot_normal(rs, hidden, latent), "bm": z(latent),
            "W3": glorot_normal(rs, latent, hidden), "b3": z(hidden), "W4": glorot_normal(rs, hidden, n_items), "b4": z(n_items)}


def forward(w, x, drop_scale=None, dtype=np.float32):
    f = lambda a: np.asarray(a, dtype=dtype)
    x = f(x)
    xn = x / np.sqrt(np.maximum((x * x).sum(axis=1, keepdims=True), dtype(1e-12)))
    if drop_scale is not None:
        xn = xn * f(drop_scale)
    h = np.tanh(xn @ f(w["W1"]) + f(w["b1"]))
    z = np.tanh(h @ f(w["Wm"]) + f(w["bm"]))
    h2 = np.tanh(z @ f(w["W3"]) + f(w["b3"]))
    logits = h2 @ f(w["W4"]) + f(w["b4"])
    return dict(x=x, xn=xn, h=h, z=z, h2=h2, logits=logits)


def loss_from(c):
    return -np.mean(np.sum(log_softmax(c["logits"]) * c["x"], axis=-1))


def gradients(w, c):
    B = c["x"].shape[0]
    x = c["x"]
    dl = (np.exp(log_softmax(c["logits"])) * x.sum(axis=1, keepdims=True) - x) / B
    W = lambda k: np.asarray(w[k], dl.dtype)
    g = {"W4": c["h2"].T @ dl, "b4": dl.sum(0)}
    dh2 = (dl @ W("W4").T) * (1 - c["h2"] ** 2)
    g["W3"], g["b3"] = c["z"].T @ dh2, dh2.sum(0)
    dz = (dh2 @ W("W3").T) * (1 - c["z"] ** 2)
    g["Wm"], g["bm"] = c["h"].T @ dz, dz.sum(0)
    dh = (dz @ W("Wm").T) * (1 - c["h"] ** 2)
    g["W1"], g["b1"] = c["xn"].T @ dh, dh.sum(0)
    return g


class MultiDAEOracle:
    def __init__(self, weights, lr):
        self.w = {k: np.array(v, dtype=np.float32, copy=True) for k, v in weights.items()}
        self.m = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.lr, self.t = lr, 0

    def train_step(self, x, drop_scale=None):
        c = forward(self.w, x, drop_scale)
        loss = loss_from(c)
        g = gradients(self.w, c)
        self.t += 1
        a = adam_lr_t(self.lr, self.t)
        f = np.float32
        for k in NAMES:
            gg = g[k].astype(np.float32)
            self.m[k] += (gg - self.m[k]) * tf_clauses.one_minus(BETA1)
            self.v[k] += (gg * gg - self.v[k]) * tf_clauses.one_minus(BETA2)
            self.w[k] -= (self.m[k] * a) / (np.sqrt(self.v[k]) + f(EPS))
        return float(loss)

    def predict(self, x):
        return log_softmax(forward(self.w, x)["logits"])
