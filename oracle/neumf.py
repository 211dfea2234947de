"""NumPy restatement of NeuMF and GMF.  TEST INFRASTRUCTURE -- "parity unpinned" vs TensorFlow (not installable);
gradients pinned by torch autograd (tests/test_oracle_neumf.py).

Follows
  neural/NeuMF/neural_matrix_factorization_model.py:75-93 (call), :96-106 (train_step), :120-144 (get_recs)
  neural/GeneralizedMF/generalized_matrix_factorization_model.py:59-66 (call), :68-79 (train_step)
  mf = Umf[u]*Imf[i]; mlp = relu Dense chain on concat(Umlp[u], Imlp[i]) (Dropout(0) before each Dense);
  y = sigmoid(Dense(1)([mf ; mlp]))      (GMF: sigmoid((Umf[u]*Imf[i]) @ h), no bias)
[TF] keras.losses.BinaryCrossentropy(): probabilities clipped to [1e-7, 1-1e-7], mean over the batch;
[TF] Embedding gradients are IndexedSlices -> Adam sparse apply (all rows decay); Dense variables use
     ApplyAdam (m += (g-m)(1-b1) ...).
"""
import numpy as np

from .bprmf_batch import adam_lr_t, adam_tf_sparse_apply

BETA1, BETA2, EPS = 0.9, 0.999, 1e-7


def glorot_uniform(rs, rows, cols):
    lim = np.sqrt(6.0 / (rows + cols))
    return rs.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)


def init_neumf(U, I, F, seed, units=None):
    """neural_matrix_factorization.py:71-72: mlp_hidden_size = (4F, 2F, F), mlp_factors = F."""
    rs = np.random.RandomState(seed)
    units = list(units) if units is not None else [4 * F, 2 * F, F]
    w = {"Umf": glorot_uniform(rs, U, F), "Imf": glorot_uniform(rs, I, F), "Umlp": glorot_uniform(rs, U, F),
         "Imlp": glorot_uniform(rs, I, F), "W": [], "b": []}
    kin = 2 * F
    for n in units:
        w["W"].append(glorot_uniform(rs, kin, n))
        w["b"].append(np.zeros(n, np.float32))
        kin = n
    w["hw"] = glorot_uniform(rs, F + units[-1], 1)[:, 0].copy()
    w["hb"] = np.zeros(1, np.float32)
    return w


def init_gmf(U, I, F, seed):
    rs = np.random.RandomState(seed)
    return {"Umf": glorot_uniform(rs, U, F), "Imf": glorot_uniform(rs, I, F), "hw": glorot_uniform(rs, F, 1)[:, 0].copy()}


def forward(w, u, i, dtype=np.float32):
    f = lambda a: np.asarray(a, dtype=dtype)
    c = {}
    parts = []
    if "Umf" in w:
        c["eu"], c["ei"] = f(w["Umf"])[u], f(w["Imf"])[i]
        c["mf"] = c["eu"] * c["ei"]
        parts.append(c["mf"])
    if "Umlp" in w:
        x = np.concatenate([f(w["Umlp"])[u], f(w["Imlp"])[i]], axis=1)
        c["acts"] = [x]
        for W, b in zip(w["W"], w["b"]):
            x = np.maximum(x @ f(W) + f(b), 0)
            c["acts"].append(x)
        parts.append(x)
    c["cat"] = np.concatenate(parts, axis=1)
    logit = c["cat"] @ f(w["hw"]) + (f(w["hb"])[0] if "hb" in w else 0)
    c["p"] = 1 / (1 + np.exp(-logit))
    return c


def bce(p, y):
    pc = np.clip(p, 1e-7, 1 - 1e-7)
    return float(-np.mean(y * np.log(pc) + (1 - y) * np.log(1 - pc)))


def gradients(w, c, u, i, y):
    n = len(y)
    p = c["p"]
    dlogit = np.where((p > 1e-7) & (p < 1 - 1e-7), (p - y) / n, 0.0).astype(p.dtype)
    g = {"hw": c["cat"].T @ dlogit}
    if "hb" in w:
        g["hb"] = np.array([dlogit.sum()])
    hw = np.asarray(w["hw"], p.dtype)
    F = w["Umf"].shape[1] if "Umf" in w else 0
    if "Umf" in w:
        dmf = dlogit[:, None] * hw[None, :F]
        g["Umf"], g["Imf"] = np.zeros(w["Umf"].shape, p.dtype), np.zeros(w["Imf"].shape, p.dtype)
        np.add.at(g["Umf"], u, dmf * c["ei"])
        np.add.at(g["Imf"], i, dmf * c["eu"])
    if "Umlp" in w:
        d = dlogit[:, None] * hw[None, F:]
        g["W"], g["b"] = [None] * len(w["W"]), [None] * len(w["W"])
        for l in range(len(w["W"]) - 1, -1, -1):
            d = d * (c["acts"][l + 1] > 0)
            g["W"][l] = c["acts"][l].T @ d
            g["b"][l] = d.sum(0)
            d = d @ np.asarray(w["W"][l], p.dtype).T
        E = w["Umlp"].shape[1]
        g["Umlp"], g["Imlp"] = np.zeros(w["Umlp"].shape, p.dtype), np.zeros(w["Imlp"].shape, p.dtype)
        np.add.at(g["Umlp"], u, d[:, :E])
        np.add.at(g["Imlp"], i, d[:, E:])
    return g


class NeuMFOracle:
    def __init__(self, weights, lr):
        cp = lambda v: [np.array(x, np.float32, copy=True) for x in v] if isinstance(v, list) else np.array(v, np.float32, copy=True)
        self.w = {k: cp(v) for k, v in weights.items()}
        zl = lambda v: [np.zeros_like(x) for x in v] if isinstance(v, list) else np.zeros_like(v)
        self.m = {k: zl(v) for k, v in self.w.items()}
        self.v = {k: zl(v) for k, v in self.w.items()}
        self.lr, self.t = lr, 0

    def _dense(self, th, m, v, g):
        f = np.float32
        a = adam_lr_t(self.lr, self.t)
        g = g.astype(np.float32)
        m += (g - m) * f(1 - BETA1)
        v += (g * g - v) * f(1 - BETA2)
        th -= (m * a) / (np.sqrt(v) + f(EPS))

    def train_step(self, u, i, y):
        u, i = np.asarray(u, np.int64), np.asarray(i, np.int64)
        y = np.asarray(y, np.float32)
        c = forward(self.w, u, i)
        loss = bce(c["p"], y)
        g = gradients(self.w, c, u, i, y)
        self.t += 1
        for k in ("Umf", "Imf", "Umlp", "Imlp"):
            if k in self.w:
                adam_tf_sparse_apply(self.w[k], self.m[k], self.v[k], g[k].astype(np.float32), self.lr, self.t)
        if "W" in self.w:
            for l in range(len(self.w["W"])):
                self._dense(self.w["W"][l], self.m["W"][l], self.v["W"][l], g["W"][l])
                self._dense(self.w["b"][l], self.m["b"][l], self.v["b"][l], g["b"][l])
        self._dense(self.w["hw"], self.m["hw"], self.v["hw"], g["hw"])
        if "hb" in self.w:
            self._dense(self.w["hb"], self.m["hb"], self.v["hb"], g["hb"])
        return loss

    def predict(self, u, i):
        return forward(self.w, np.asarray(u, np.int64), np.asarray(i, np.int64))["p"]
