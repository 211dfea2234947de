"""LightGCNModel on the MI355X -- counterpart of elliot/recommender/graph_based/lightgcn/LightGCN_model.py:19-172.

Same constructor arguments.  `train_step(batch)` = `_propagate_embeddings` (:68-94, assigned to the variables) + the bias-free BPR
head with the doubled L2 term (:136-167) + Keras Adam; scoring through `recommend(...)` (predict :131-133 + get_top_k :171-172 fused).
The tables start at ZERO, as the reference's `_create_weights` (:63-65) creates them -- where every gradient of the head vanishes and
the model never moves; `init_weights=(Gu, Gi)` injects tables (tests, warm starts).  `n_fold` (:47, :96-109) only cuts TensorFlow's
sparse product into row blocks: accepted and ignored.
"""
import pickle

import numpy as np
import torch

from .... import ops
from ...latent_factor_models.BPRMF_batch.BPRMF_batch_model import DeferredLoss


class LightGCNModel:
    def __init__(self, num_users, num_items, learning_rate, embed_k, l_w, n_layers, n_fold, adjacency, laplacian, random_seed,
                 name="LightGCN", ctx=None, init_weights=None, **kwargs):
        self.ctx = ctx or ops.get_context(0)
        self.num_users, self.num_items, self.embed_k = int(num_users), int(num_items), int(embed_k)
        self.learning_rate, self.l_w, self.n_layers, self.n_fold = learning_rate, l_w, int(n_layers), n_fold
        if init_weights is not None:
            Gu, Gi = init_weights
        else:
            Gu = np.zeros((self.num_users, self.embed_k), np.float32)          # :64-65 tf.zeros
            Gi = np.zeros((self.num_items, self.embed_k), np.float32)
        lap = laplacian.tocsr()
        lap.sort_indices()
        self.graph = ops.GraphCSR(self.ctx, lap.indptr, lap.indices, lap.data.astype(np.float32), self.num_users, self.embed_k)
        self.state = ops.LightGcnDeviceState(self.ctx, Gu, Gi, self.graph, n_layers=self.n_layers)
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
        b = self.state.bpr
        b.sync()
        d = {"Gu": b.Gu.cpu().numpy(), "Gi": b.Gi.cpu().numpy(), "_step": b.step}
        for n in ("mGu", "vGu", "mGi", "vGi"):
            d[n] = getattr(b, n).cpu().numpy()
        return d

    def set_model_state(self, d):
        b = self.state.bpr
        self._weights_version += 1
        b.Gu.copy_(torch.from_numpy(d["Gu"]))
        b.Gi.copy_(torch.from_numpy(d["Gi"]))
        b.step = int(d.get("_step", 0))
        for n in ("mGu", "vGu", "mGi", "vGi"):
            if n in d:
                getattr(b, n).copy_(torch.from_numpy(d[n]))

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.get_model_state(), f)

    def load_weights(self, path):
        with open(path, "rb") as f:
            self.set_model_state(pickle.load(f))
