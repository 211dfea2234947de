"""NumPy restatement of the Mult-VAE model.  TEST INFRASTRUCTURE -- "parity unpinned" vs TensorFlow (not
installable here); gradients are pinned by an independent torch-autograd derivation (tests/test_oracle_vae.py).

Follows elliot/recommender/autoencoders/vae/multi_vae_model.py:
  Sampling.call        :24-29    z = mu + exp(0.5 logvar) eps
  Encoder.call         :57-64    l2_normalize(axis=1) -> Dropout -> Dense(tanh) -> Dense mean / Dense log_var
  Decoder.call         :81-83    Dense(tanh) -> Dense
  call                 :115-123  KL = -0.5 * reduce_mean(logvar - mu^2 - exp(logvar) + 1)   (mean over batch AND latent)
  train_step           :126-142  neg_ll = -mean_b sum_i log_softmax(logits) x ; loss = neg_ll + anneal * KL ; Adam
  predict              :145-155  log_softmax(logits), z sampled also at inference
[TF] K.l2_normalize(x, 1) = x / sqrt(max(sum x^2, 1e-12)); Dropout scales kept units by 1/(1-rate);
[TF] dense variables use ApplyAdam: m += (g-m)(1-b1); v += (g*g-v)(1-b2); var -= lr_t m / (sqrt(v)+eps), eps=1e-7;
the declared kernel_regularizers are never added to the loss (train_step uses neg_ll + anneal*KL only).
"""
import numpy as np

from . import tf_clauses

BETA1, BETA2, EPS = 0.9, 0.999, 1e-7
NAMES = ("W1", "b1", "Wm", "bm", "Wv", "bv", "W3", "b3", "W4", "b4")


def glorot_normal(rs, fan_in, fan_out):
    """keras GlorotNormal: truncated normal, stddev sqrt(2/(fan_in+fan_out))/0.87962566 (distribution only)."""
    std = np.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    x = rs.normal(size=(fan_in, fan_out))
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rs.normal(size=int(bad.sum()))
        bad = np.abs(x) > 2
    return (x * std).astype(np.float32)


def init_weights(n_items, hidden, latent, seed):
    rs = np.random.RandomState(seed)
    z = lambda n: np.zeros(n, np.float32)
    return {"W1": glorot_normal(rs, n_items, hidden), "b1": z(hidden), "Wm": glorot_normal(rs, hidden, latent),
            "bm": z(latent), "Wv": glorot_normal(rs, hidden, latent), "bv": z(latent),
            "W3": glorot_normal(rs, latent, hidden), "b3": z(hidden), "W4": glorot_normal(rs, hidden, n_items),
            "b4": z(n_items)}


def log_softmax(a):
    m = a.max(axis=1, keepdims=True)
    return a - m - np.log(np.exp(a - m).sum(axis=1, keepdims=True))


def forward(w, x, eps, drop_scale=None, dtype=np.float32):
    """x: dense [B, I] batch; eps: [B, L]; drop_scale: [B, I] of {0, 1/(1-rate)} or None."""
    f = lambda a: np.asarray(a, dtype=dtype)
    x = f(x)
    ss = (x * x).sum(axis=1, keepdims=True)
    # [TF] clause (oracle/tf_clauses.py): K.l2_normalize = x * rsqrt(max(sum x^2, 1e-12)); switched off: x / (sqrt(sum x^2) + 1e-12)
    xn = x / np.sqrt(np.maximum(ss, dtype(1e-12))) if tf_clauses.get("l2_normalize_epsilon_1e12_inside_max") else x / (np.sqrt(ss) + dtype(1e-12))
    if drop_scale is not None:
        xn = xn * f(drop_scale)
    h = np.tanh(xn @ f(w["W1"]) + f(w["b1"]))
    mu = h @ f(w["Wm"]) + f(w["bm"])
    lv = h @ f(w["Wv"]) + f(w["bv"])
    z = mu + np.exp(dtype(0.5) * lv) * f(eps)
    h2 = np.tanh(z @ f(w["W3"]) + f(w["b3"]))
    logits = h2 @ f(w["W4"]) + f(w["b4"])
    return dict(x=x, xn=xn, h=h, mu=mu, lv=lv, z=z, h2=h2, logits=logits, eps=f(eps))


def loss_from(c, anneal):
    kl = -0.5 * np.mean(c["lv"] - c["mu"] ** 2 - np.exp(c["lv"]) + 1)
    neg_ll = -np.mean(np.sum(log_softmax(c["logits"]) * c["x"], axis=-1))
    return neg_ll + anneal * kl


def gradients(w, c, anneal):
    B, L = c["mu"].shape
    x = c["x"]
    sm = np.exp(log_softmax(c["logits"]))
    dl = (sm * x.sum(axis=1, keepdims=True) - x) / B
    g = {}
    g["W4"] = c["h2"].T @ dl
    g["b4"] = dl.sum(0)
    dh2 = (dl @ np.asarray(w["W4"], dl.dtype).T) * (1 - c["h2"] ** 2)
    g["W3"] = c["z"].T @ dh2
    g["b3"] = dh2.sum(0)
    dz = dh2 @ np.asarray(w["W3"], dl.dtype).T
    ks = anneal / (B * L)
    dmu = dz + ks * c["mu"]
    dlv = dz * c["eps"] * 0.5 * np.exp(0.5 * c["lv"]) + ks * 0.5 * (np.exp(c["lv"]) - 1)
    g["Wm"], g["bm"] = c["h"].T @ dmu, dmu.sum(0)
    g["Wv"], g["bv"] = c["h"].T @ dlv, dlv.sum(0)
    dh = (dmu @ np.asarray(w["Wm"], dl.dtype).T + dlv @ np.asarray(w["Wv"], dl.dtype).T) * (1 - c["h"] ** 2)
    g["W1"] = c["xn"].T @ dh
    g["b1"] = dh.sum(0)
    return g


def adam_lr_t(lr, t):
    b1p = np.power(np.float32(BETA1), np.float32(t))
    b2p = np.power(np.float32(BETA2), np.float32(t))
    return np.float32(lr) * np.sqrt(np.float32(1.0) - b2p) / (np.float32(1.0) - b1p)


class MultiVAEOracle:
    def __init__(self, weights, lr):
        self.w = {k: np.array(v, dtype=np.float32, copy=True) for k, v in weights.items()}
        self.m = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.lr, self.t = lr, 0

    def train_step(self, x, eps, anneal, drop_scale=None):
        c = forward(self.w, x, eps, drop_scale)
        loss = loss_from(c, anneal)
        g = gradients(self.w, c, np.float32(anneal))
        self.t += 1
        a = adam_lr_t(self.lr, self.t)
        f = np.float32
        for k in NAMES:
            gg = g[k].astype(np.float32)
            self.m[k] += (gg - self.m[k]) * tf_clauses.one_minus(BETA1)
            self.v[k] += (gg * gg - self.v[k]) * tf_clauses.one_minus(BETA2)
            self.w[k] -= (self.m[k] * a) / (np.sqrt(self.v[k]) + f(EPS))
        return float(loss)

    def predict(self, x, eps):
        return log_softmax(forward(self.w, x, eps)["logits"])
