"""CML_model on the MI355X -- counterpart of elliot/recommender/latent_factor_models/CML/CML_model.py:18-145.

Variables Gu [U,F], Gi [I,F], Bi [I] (keras 'uniform' = U(-0.05, 0.05), :134-145), Adam.  train_step reproduces the
reference's [B,B] broadcast of distances against biases (ops.CmlDeviceState / el_cml_train_step); scoring is
-|Gu[u] - Gi[i]|^2 + Bi[i] through the fused top-k kernels."""
import pickle

import numpy as np
import torch

from .... import ops
from ..BPRMF_batch.BPRMF_batch_model import DeferredLoss


class CML_model:
    def __init__(self, user_factors=200, item_factors=200, learning_rate=0.001, l_w=0, l_b=0, margin=0.5, num_users=100,
                 num_items=100, random_seed=42, name="CML", ctx=None, init_weights=None, **kwargs):
        if user_factors != item_factors:
            raise ValueError("CML needs user and item vectors of one size")
        self.ctx = ctx or ops.get_context(0)
        self._learning_rate, self.l_w, self.l_b, self.margin = learning_rate, l_w, l_b, margin
        self._num_users, self._num_items = num_users, num_items
        if init_weights is None:
            rs = np.random.RandomState(random_seed)
            draw = lambda *shape: rs.uniform(-0.05, 0.05, size=shape).astype(np.float32)
            init_weights = (draw(num_users, user_factors), draw(num_items, item_factors), draw(num_items))
        self.state = ops.CmlDeviceState(self.ctx, *init_weights)

    def _idx(self, x):
        if isinstance(x, torch.Tensor):
            return x.reshape(-1).to(device=self.ctx.device, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.int32)).to(self.ctx.device)

    def train_step(self, batch):
        u, i, j = (self._idx(x) for x in batch)
        self.state.train_step(u, i, j, self._learning_rate, self.l_w, self.l_b, self.margin)
        return DeferredLoss(self.state)

    def recommend(self, mask, k, start, stop, item_offset=0):
        kind, csr = mask if mask is not None else (None, None)
        return self.state.recommend(start, stop, k, excl=csr if kind == "excl" else None, cand=csr if kind == "cand" else None)

    def predict(self, start, stop, **kwargs):
        """[stop-start, I] scores (:97-102) -- compatibility path over every (user, item) pair."""
        n, I = stop - start, self._num_items
        items = torch.arange(I, dtype=torch.int32, device=self.ctx.device).repeat(n, 1)
        return self.state.rescore(items, start)

    def get_top_k(self, predictions, train_mask, k=100):
        kind, csr = train_mask
        idx, val = ops.dense_topk(self.ctx, predictions, 0, predictions.shape[0], k, excl=csr if kind == "excl" else None,
                                  cand=csr if kind == "cand" else None)
        return val, idx

    def get_model_state(self):
        st = self.state
        return {"Gu": st.Gu.cpu().numpy(), "Gi": st.Gi.cpu().numpy(), "Bi": st.Bi.cpu().numpy(), "step": st.step,
                **{n: getattr(st, n).cpu().numpy() for n in ("mGu", "vGu", "mGi", "vGi", "mBi", "vBi")}}

    def set_model_state(self, d):
        st = self.state
        for n, v in d.items():
            if n == "step":
                st.step = int(v)
            else:
                getattr(st, n).copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
        st._items2 = None

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.get_model_state(), f)

    def load_weights(self, path):
        with open(path, "rb") as f:
            self.set_model_state(pickle.load(f))
