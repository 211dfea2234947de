"""NGCFModel on the MI355X -- counterpart of elliot/recommender/graph_based/ngcf/NGCF_model.py:18-226.

Same constructor arguments.  `train_step(batch)` = `_propagate_embeddings` (:106-142: sparse product, the two dense transforms, leaky_relu,
message dropout, row normalisation, concat, assigned to the variables) + the bias-free BPR head on the full-width rows with the doubled L2
term (:187-217) + Adam (tables: every-row sparse apply; GraphLayers: their L2-only gradient).  The tables start at ZERO as the reference
creates them (:88-91) -- a state in which the layer-0 columns never receive a gradient -- unless `init_weights=(Gu, Gi[, layers])`
injects them; the GraphLayers follow GlorotUniform ([kin, kout] and [1, kout] limits, :95-104; TensorFlow's seeded stream itself is not
reproducible).  node_dropout (:56-60, :153-158): one fixed sparsification of the Laplacian at construction, keep probability
node_dropout[0], kept entries scaled by 1 / keep.  `n_fold` only cuts TensorFlow's sparse product into row blocks: ignored.
"""
import pickle

import numpy as np
import torch

from .... import ops
from ...latent_factor_models.BPRMF_batch.BPRMF_batch_model import DeferredLoss


class NGCFModel:
    def __init__(self, num_users, num_items, learning_rate, embed_k, l_w, weight_size, n_layers, node_dropout, message_dropout, n_fold,
                 adjacency, laplacian, random_seed, name="NGFC", ctx=None, init_weights=None, **kwargs):
        self.ctx = ctx or ops.get_context(0)
        self.num_users, self.num_items, self.embed_k = int(num_users), int(num_items), int(embed_k)
        self.learning_rate, self.l_w = learning_rate, l_w
        self.weight_size_list = [self.embed_k] + [int(w) for w in weight_size]
        W = sum(self.weight_size_list)
        rs = np.random.RandomState(random_seed)
        layers = None
        if init_weights is not None:
            Gu, Gi = init_weights[0], init_weights[1]
            layers = init_weights[2] if len(init_weights) > 2 else None
        else:
            Gu, Gi = np.zeros((self.num_users, W), np.float32), np.zeros((self.num_items, W), np.float32)     # :88-91 tf.zeros
        if layers is None:
            def glorot(a, b):
                lim = np.sqrt(6.0 / (a + b))
                return rs.uniform(-lim, lim, size=(a, b)).astype(np.float32)
            layers = []
            for k in range(int(n_layers)):
                kin, kout = self.weight_size_list[k], self.weight_size_list[k + 1]
                layers.append({"W1": glorot(kin, kout), "b1": glorot(1, kout), "W2": glorot(kin, kout), "b2": glorot(1, kout)})
        lap = laplacian.tocsr().astype(np.float32)
        lap.sort_indices()
        vals = lap.data.copy()
        if len(node_dropout):
            keep = float(node_dropout[0])
            mask = np.floor(keep + rs.uniform(size=vals.shape[0])).astype(bool)                                 # :74-83
            vals = np.where(mask, vals * np.float32(1.0 / keep), np.float32(0.0)).astype(np.float32)
        self.graph = ops.GraphCSR(self.ctx, lap.indptr, lap.indices, vals, self.num_users, max(self.weight_size_list))
        drop = [float(x) for x in message_dropout] if len(message_dropout) else [0.0] * int(n_layers)
        self.state = ops.NgcfDeviceState(self.ctx, Gu, Gi, self.graph, layers, self.embed_k, message_dropout=drop, dropout_seed=random_seed)
        self._weights_version, self._scored_version = 0, -1

    def _as_index(self, x):
        if isinstance(x, torch.Tensor):
            return x.reshape(-1).to(device=self.ctx.device, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.int32)).to(self.ctx.device)

    def train_step(self, batch):
        u, i, j = (self._as_index(x) for x in batch)
        self._weights_version += 1
        self.state.train_step(u, i, j, self.learning_rate, self.l_w)
        return DeferredLoss(self.state)

    def recommend(self, mask, k, start, stop, item_offset=0):
        kind, csr = mask if mask is not None else (None, None)
        st = self.state
        same = self._scored_version == self._weights_version
        self._scored_version = self._weights_version
        return ops.score_topk(self.ctx, st.Gu, st.Gi, None, start, stop, k, excl=csr if kind == "excl" else None,
                              cand=csr if kind == "cand" else None, item_offset=item_offset, items_unchanged=same)

    def get_top_k(self, predictions, train_mask, k=100):
        kind, csr = train_mask
        idx, val = ops.dense_topk(self.ctx, predictions, 0, predictions.shape[0], k, excl=csr if kind == "excl" else None,
                                  cand=csr if kind == "cand" else None)
        return val, idx

    def get_model_state(self):
        st, b = self.state, self.state.bpr
        b.sync()
        return {"Gu": b.Gu.cpu().numpy(), "Gi": b.Gi.cpu().numpy(), "_step": b.step,
                "layers": [{k: v.cpu().numpy() for k, v in l.items()} for l in st.layers]}

    def set_model_state(self, d):
        st, b = self.state, self.state.bpr
        self._weights_version += 1
        b.Gu.copy_(torch.from_numpy(d["Gu"]))
        b.Gi.copy_(torch.from_numpy(d["Gi"]))
        b.step = int(d.get("_step", 0))
        for l, src in zip(st.layers, d.get("layers", [])):
            for k, v in src.items():
                l[k].copy_(torch.from_numpy(v))

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.get_model_state(), f)

    def load_weights(self, path):
        with open(path, "rb") as f:
            self.set_model_state(pickle.load(f))
