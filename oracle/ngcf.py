"""NumPy restatement of the reference's NGCF model.  TEST INFRASTRUCTURE -- TensorFlow library semantics ([TF]) recalled, not executed
("parity unpinned" for them); the FILE's algebra is pinned by executing NGCF_model.py on the tensorflow stand-in
(oracle/gen_golden_tfshim.py -> tests/golden/tfshim_ngcf.npz).

Follows elliot/recommender/graph_based/ngcf/NGCF_model.py:
  _create_weights        :85-104   Gu, Gi = tf.zeros([rows, sum(weight_size_list)]) (the first embed_k columns are the layer-0 embeddings);
                                   per layer W_1, b_1 ([1, kout]), W_2, b_2 ~ GlorotUniform
  _propagate_embeddings  :106-142  ego_0 = first embed_k columns; per layer: lap = L ego; first = (lap + ego) W_1 + b_1; second =
                                   (ego * lap) W_2 + b_2; ego = leaky_relu(first + second) [TF: slope 0.2]; ego = dropout(ego, rate) [TF];
                                   all_embeddings += l2_normalize(ego, axis=1) [TF: x / sqrt(max(sum x^2, 1e-12))]; concat along columns;
                                   ASSIGNED to Gu / Gi (:141-142): no gradient flows through the propagation
  train_step             :187-217  the bias-free BPR head on the full-width rows; reg_loss = l_w * (l2 of the three gathers + l2 of every
                                   GraphLayers value) * 2; Adam on Gu, Gi (IndexedSlices) and on the GraphLayers (dense; their gradient is
                                   the L2 term's alone: 2 l_w theta)
With all-zero tables (the reference's initialisation) ego_0 = 0, every node gets the same propagated row and the layer-0 columns never
receive a gradient: the model as written cannot learn; parity tests inject tables.  message_dropout > 0 draws TensorFlow's stateful
stream: only rate 0 is restated (the device draws its own counter-based mask)."""
import numpy as np

from . import bprmf_batch as ob

f32 = np.float32


def propagate(Gu, Gi, lap, layers, embed_k):
    U = Gu.shape[0]
    ego = np.concatenate([Gu[:, :embed_k], Gi[:, :embed_k]], 0).astype(f32)
    all_emb = [ego]
    for l in layers:
        lp = (lap @ ego).astype(f32)                                                    # :118-121
        first = ((lp + ego) @ l["W1"] + l["b1"]).astype(f32)                            # :123-126
        second = ((ego * lp) @ l["W2"] + l["b2"]).astype(f32)                           # :128-132
        s = first + second
        ego = np.where(s > 0, s, f32(0.2) * s).astype(f32)                              # :134 leaky_relu
        norm = ego / np.sqrt(np.maximum((ego * ego).sum(1, keepdims=True), f32(1e-12)))  # :138
        all_emb.append(norm.astype(f32))
    allc = np.concatenate(all_emb, 1)                                                   # :140
    return allc[:U].copy(), allc[U:].copy()


class NGCFOracle:
    def __init__(self, Gu, Gi, lap, layers, embed_k, lr, l_w):
        self.Gu, self.Gi = np.array(Gu, f32, copy=True), np.array(Gi, f32, copy=True)
        self.layers = [{k: np.array(v, f32, copy=True) for k, v in l.items()} for l in layers]
        self.lap, self.embed_k, self.lr, self.l_w = lap, embed_k, lr, l_w
        self.t = 0
        self.slots = {"Gu": (np.zeros_like(self.Gu), np.zeros_like(self.Gu)), "Gi": (np.zeros_like(self.Gi), np.zeros_like(self.Gi))}
        self.lslots = [{k: (np.zeros_like(v), np.zeros_like(v)) for k, v in l.items()} for l in self.layers]

    def train_step(self, batch):
        u, i, j = (np.asarray(x).reshape(-1).astype(np.int64) for x in batch)
        self.Gu, self.Gi = propagate(self.Gu, self.Gi, self.lap, self.layers, self.embed_k)
        Bi = np.zeros(self.Gi.shape[0], f32)
        loss = float(ob.forward_loss(self.Gu, self.Gi, Bi, u, i, j, 2.0 * self.l_w, 0.0))
        loss += float(self.l_w) * float(sum((p.astype(np.float64) ** 2).sum() for l in self.layers for p in l.values()))
        _, dGu, dGi = ob.gradients(self.Gu, self.Gi, Bi, u, i, j, 2.0 * self.l_w, 0.0)
        self.t += 1
        for name, theta, g in (("Gu", self.Gu, dGu), ("Gi", self.Gi, dGi)):
            m, v = self.slots[name]
            ob.adam_tf_sparse_apply(theta, m, v, g, self.lr, self.t)
        lr_t = ob.adam_lr_t(self.lr, self.t)
        for l, sl in zip(self.layers, self.lslots):
            for k, p in l.items():
                m, v = sl[k]
                g = f32(2.0 * self.l_w) * p
                m += (g - m) * f32(1 - 0.9)                                             # [TF] Keras dense apply
                v += (g * g - v) * f32(1 - 0.999)
                p -= (m * lr_t) / (np.sqrt(v) + f32(1e-7))
        return loss

    def predict(self, start, stop):
        return self.Gu[start:stop] @ self.Gi.T
