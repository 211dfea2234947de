"""MFModel on the MI355X -- counterpart of elliot/recommender/latent_factor_models/BPRMF/BPRMF_model.py:14-139.

fp64 like the reference.  `train_step(batch)` reproduces the reference's strictly sequential per-triplet
updates (:87-117) by level scheduling: the batch is cut into dependency levels on the host (triplets in one
level share no user/item row), each level is one conflict-free launch of `el_bprsgd_apply`, so every row sees
exactly the predecessor state it would see in the sequential loop.  `hogwild=True` applies the whole batch in
one launch instead (not the reference's semantics).
"""
import pickle

import numpy as np
import torch

from .... import ops


class MFModel(object):
    def __init__(self, F, data, lr, user_regularization, bias_regularization, positive_item_regularization,
                 negative_item_regularization, random_seed, *args, ctx=None, hogwild=False, init_weights=None):
        self.ctx = ctx or ops.get_context(0)
        np.random.seed(random_seed)                                   # :24
        self._factors = F
        self._users, self._items = data.users, data.items
        self._private_users, self._public_users = data.private_users, data.public_users
        self._private_items, self._public_items = data.private_items, data.public_items
        self._learning_rate = lr
        self._user_regularization = user_regularization
        self._bias_regularization = bias_regularization
        self._positive_item_regularization = positive_item_regularization
        self._negative_item_regularization = negative_item_regularization
        self._hogwild = hogwild
        self.initialize(*args, init_weights=init_weights)

    def initialize(self, loc: float = 0, scale: float = 0.1, init_weights=None):
        """:40-56 -- N(loc, scale) user then item factors from the global legacy np.random stream, zero biases."""
        nu, ni = len(self._users), len(self._items)
        if init_weights is not None:
            P, Q, b = init_weights
        else:
            b = np.zeros(ni)
            P = np.random.normal(loc=loc, scale=scale, size=(nu, self._factors))
            Q = np.random.normal(loc=loc, scale=scale, size=(ni, self._factors))
        self.state = ops.BprSgdDeviceState(self.ctx, P, Q, b, self._learning_rate, self._bias_regularization,
                                           self._user_regularization, self._positive_item_regularization,
                                           self._negative_item_regularization)
        self.levels_last = 0

    @property
    def name(self):
        return "MF"

    def train_step(self, batch, **kwargs):
        """:87-89 over a batch of (u, i, j); arrays of shape [B] or [B, 1], host or device."""
        u, i, j = (x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x) for x in batch)
        u, i, j = (np.ascontiguousarray(x.reshape(-1), dtype=np.int32) for x in (u, i, j))
        if self._hogwild:
            d = self.ctx.device
            self.state.apply(torch.from_numpy(u).to(d), torch.from_numpy(i).to(d), torch.from_numpy(j).to(d))
        else:
            self.levels_last = self.state.apply_sequential_equivalent(u, i, j)

    def recommend(self, mask, k, start, stop, item_offset=0):
        """get_user_predictions (:70-85) for users [start, stop): fp64 scores, masked top-k."""
        kind, csr = mask if mask is not None else (None, None)
        st = self.state
        return ops.score_topk_f64(self.ctx, st.P, st.Q, st.b, start, stop, k,
                                  excl=csr if kind == "excl" else None, cand=csr if kind == "cand" else None,
                                  item_offset=item_offset)

    def get_model_state(self):                                        # :119-125
        st = self.state
        return {"_user_bias": np.zeros(len(self._users)), "_item_bias": st.b.cpu().numpy(),
                "_user_factors": st.P.cpu().numpy(), "_item_factors": st.Q.cpu().numpy()}

    def set_model_state(self, d):                                     # :127-131
        st = self.state
        st.b.copy_(torch.from_numpy(np.asarray(d["_item_bias"], dtype=np.float64)))
        st.P.copy_(torch.from_numpy(np.asarray(d["_user_factors"], dtype=np.float64)))
        st.Q.copy_(torch.from_numpy(np.asarray(d["_item_factors"], dtype=np.float64)))

    def load_weights(self, path):
        with open(path, "rb") as f:
            self.set_model_state(pickle.load(f))

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.get_model_state(), f)
