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

from . import tf_clauses
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


def forward(w, u, i, dtype=np.float32, masks=None):
    """masks: per Dense layer l the Dropout scale matrix ({0, 1/(1-rate)}, shape of the layer INPUT) applied in front of it
    (neural_matrix_factorization_model.py:58-61, training only) or None."""
    f = lambda a: np.asarray(a, dtype=dtype)
    c = {}
    parts = []
    if "Umf" in w:
        c["eu"], c["ei"] = f(w["Umf"])[u], f(w["Imf"])[i]
        c["mf"] = c["eu"] * c["ei"]
        parts.append(c["mf"])
    if "Umlp" in w:
        x = np.concatenate([f(w["Umlp"])[u], f(w["Imlp"])[i]], axis=1)
        c["ins"], c["outs"], c["masks"] = [], [], masks
        for l, (W, b) in enumerate(zip(w["W"], w["b"])):
            if masks is not None:
                x = x * f(masks[l])
            c["ins"].append(x)
            x = np.maximum(x @ f(W) + f(b), 0)
            c["outs"].append(x)
        parts.append(x)
    c["cat"] = np.concatenate(parts, axis=1)
    logit = c["cat"] @ f(w["hw"]) + (f(w["hb"])[0] if "hb" in w else 0)
    c["p"] = 1 / (1 + np.exp(-logit))
    return c


def bce(p, y):
    """keras.losses.BinaryCrossentropy() on probabilities [TF clauses bce_clips_probabilities_at_1e7, bce_adds_epsilon_inside_log]."""
    dt = np.asarray(p).dtype.type
    pc = np.clip(p, dt(1e-7), dt(1) - dt(1e-7)) if tf_clauses.get("bce_clips_probabilities_at_1e7") else p
    eps = dt(1e-7) if tf_clauses.get("bce_adds_epsilon_inside_log") else dt(0)
    return float(-np.mean(y * np.log(pc + eps) + (1 - y) * np.log(1 - pc + eps)))


def gradients(w, c, u, i, y, relu_masks=None):
    """relu_masks: per Dense layer a boolean [n, units] array used as the ReLU derivative instead of (output > 0) -- the branch
    pattern another implementation took (the derivative is a step function: a pre-activation within round-off of 0 takes either
    branch depending on the summation order; tests/test_gpu_neumf.py compares gradients under the device's pattern after checking
    that the two patterns differ only there)."""
    n = len(y)
    p = c["p"]
    dt = p.dtype.type
    live = ((p > 1e-7) & (p < 1 - 1e-7)) if tf_clauses.get("bce_clips_probabilities_at_1e7") else np.ones(p.shape, bool)
    if tf_clauses.get("bce_adds_epsilon_inside_log"):
        # d loss / d p through the two logarithms, then the sigmoid's p (1 - p)
        dp = -(y / (p + dt(1e-7)) - (1 - y) / (1 - p + dt(1e-7)))
        dlogit = np.where(live, dp * (p * (1 - p)) / n, 0.0).astype(p.dtype)
    else:
        dlogit = np.where(live, (p - y) / n, 0.0).astype(p.dtype)
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
            d = d * ((c["outs"][l] > 0) if relu_masks is None else np.asarray(relu_masks[l], bool))
            g["W"][l] = c["ins"][l].T @ d
            g["b"][l] = d.sum(0)
            d = d @ np.asarray(w["W"][l], p.dtype).T
            if c["masks"] is not None:
                d = d * np.asarray(c["masks"][l], p.dtype)
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
        m += (g - m) * tf_clauses.one_minus(BETA1)
        v += (g * g - v) * tf_clauses.one_minus(BETA2)
        th -= (m * a) / (np.sqrt(v) + f(EPS))

    def train_step(self, u, i, y, masks=None):
        u, i = np.asarray(u, np.int64), np.asarray(i, np.int64)
        y = np.asarray(y, np.float32)
        c = forward(self.w, u, i, masks=masks)
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


def dropout_masks(n, widths, rate, seed, step):
    """The device's Dropout masks (el_neural.hip, k_nmf_dropout): Philox4x32-10 with counter (row, column // 4, step, layer),
    key = seed; the four outputs serve four consecutive columns; keep where uniform >= rate, scale 1 / (1 - rate)."""
    out = []
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    mask32 = np.uint64(0xFFFFFFFF)
    for layer, width in enumerate(widths):
        w4 = (width + 3) // 4
        b, c4 = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(w4, dtype=np.uint64), indexing="ij")
        c0, c1 = b & mask32, c4
        c2, c3 = np.full_like(b, step), np.full_like(b, layer)
        k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
        for _ in range(10):
            p0, p1 = M0 * c0, M1 * c2
            hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask32, p1 >> np.uint64(32), p1 & mask32
            c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & mask32, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & mask32, lo0
            k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
        r = np.stack([c0, c1, c2, c3], axis=-1).reshape(n, w4 * 4)[:, :width]
        uni = (r >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        out.append(np.where(uni < np.float32(rate), np.float32(0), np.float32(1) / (np.float32(1) - np.float32(rate))).astype(np.float32))
    return out
