"""MFModel of MF2020 on the MI355X -- counterpart of elliot/recommender/latent_factor_models/MF2020/MF_model.py:14-174.

Same constructor (F, data, lr, reg, random_seed), same initial draw (np.random.seed(seed); N(0, 0.1) user factors, then item factors,
:37-56), fp64 parameters in HBM; `train_step(batch)` runs the batch's samples in order (el_mf2020_train); scoring through
`recommend(...)`: the fp64 score b_u + (g + b_i + p_u . q_i) of :115-116 and a masked top-k on the device.
"""
import pickle

import numpy as np
import torch

from .... import ops


class MFModel:
    def __init__(self, F, data, lr, reg, random_seed, *args, ctx=None, init_weights=None):
        self.ctx = ctx or ops.get_context(0)
        self._factors, self._lr, self._reg = int(F), lr, reg
        n_users, n_items = len(data.users), len(data.items)
        if init_weights is not None:
            P, Q = init_weights
        else:
            rs = np.random.RandomState(random_seed)                        # np.random.seed(random_seed) (:21), then :52-55
            P = rs.normal(loc=0, scale=0.1, size=(n_users, self._factors))
            Q = rs.normal(loc=0, scale=0.1, size=(n_items, self._factors))
        self.state = ops.Mf2020DeviceState(self.ctx, P, Q, lr=lr, reg=reg)

    @property
    def name(self):
        return "MF2020"

    def train_step(self, batch, **kwargs):
        """:80-113 -- returns the batch's sum of losses (a float: the reference's caller divides it by len(batch) right away, MF.py:123)."""
        b = batch if isinstance(batch, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(batch), dtype=np.int32))
        self.state.train(b)
        return self.state.pop_loss()

    def recommend(self, mask, k, start, stop, item_offset=0):
        """prepare_predictions (:115-116) for users [start, stop) + the masked top-k (:62-78), fp64 scores."""
        st = self.state
        kind, csr = mask if mask is not None else (None, None)
        bias = st.gb + st.bi
        idx, val = ops.score_topk_f64(self.ctx, st.P, st.Q, bias, start, stop, k, excl=csr if kind == "excl" else None,
                                      cand=csr if kind == "cand" else None, item_offset=item_offset)
        return idx, val + st.bu[start:stop, None]          # (the user bias shifts a user's whole row: it does not change the order)

    def get_model_state(self):
        st = self.state
        return {"_global_bias": float(st.gb.item()), "_user_bias": st.bu.cpu().numpy(), "_item_bias": st.bi.cpu().numpy(),
                "_user_factors": st.P.cpu().numpy(), "_item_factors": st.Q.cpu().numpy()}

    def set_model_state(self, d):
        st = self.state
        st.gb.fill_(float(d["_global_bias"]))
        st.bu.copy_(torch.from_numpy(np.asarray(d["_user_bias"])))
        st.bi.copy_(torch.from_numpy(np.asarray(d["_item_bias"])))
        st.P.copy_(torch.from_numpy(np.asarray(d["_user_factors"])))
        st.Q.copy_(torch.from_numpy(np.asarray(d["_item_factors"])))

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.get_model_state(), f)

    def load_weights(self, path):
        with open(path, "rb") as f:
            self.set_model_state(pickle.load(f))
