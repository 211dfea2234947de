"""BPRMF_batch_model on the MI355X -- counterpart of
elliot/recommender/latent_factor_models/BPRMF_batch/BPRMF_batch_model.py:18-88.

Same constructor arguments; `train_step(batch)`, `predict`-free scoring through `recommend(...)`,
`get_top_k`, `save_weights` / `load_weights`.  Parameters live in HBM (ops.BprmfDeviceState); every numeric
step is a kernel of libelliot_hip.so.
"""
import pickle

import numpy as np
import torch

from .... import ops


class DeferredLoss:
    """What train_step returns: the batch losses accumulate in a device double; reading the value (float(),
    .numpy(), formatting) synchronises once -- the reference pays a D2H `.numpy()` every step (BPRMF_batch.py:106)."""

    def __init__(self, state):
        self._state = state

    def __add__(self, other):
        return self

    __radd__ = __add__

    def __float__(self):
        return self._state.pop_loss()

    def numpy(self):
        return np.float32(float(self))


class BPRMF_batch_model:
    def __init__(self, factors=200, learning_rate=0.001, l_w=0, l_b=0, num_users=100, num_items=100, random_seed=42,
                 name="NNBPRMF", ctx=None, optimizer="adam", init_weights=None, **kwargs):
        self.ctx = ctx or ops.get_context(0)
        self._factors, self._learning_rate, self._l_w, self._l_b = factors, learning_rate, l_w, l_b
        self._num_users, self._num_items = num_users, num_items
        if init_weights is not None:
            Gu, Gi, Bi = init_weights
        elif (num_users + num_items) * factors * 4 >= (64 << 20) and kwargs.get("init", "auto") != "host":
            # large tables (a 1 M x 128 user table is 512 MB): the same distribution drawn in HBM -- a host draw + H2D copy of
            # 10^8 floats costs seconds per model instance (one per HPO trial and fold, model_coordinator.py:58-65)
            g = torch.Generator(device=self.ctx.device)
            g.manual_seed(int(random_seed))
            lu, li = (6.0 / (num_users + factors)) ** 0.5, (6.0 / (num_items + factors)) ** 0.5
            Gu = (torch.rand((num_users, factors), generator=g, device=self.ctx.device) * 2 - 1) * lu
            Gi = (torch.rand((num_items, factors), generator=g, device=self.ctx.device) * 2 - 1) * li
            Bi = torch.zeros(num_items, device=self.ctx.device)
        else:
            # tf.initializers.GlorotUniform (:39-42): U(-L, L), L = sqrt(6 / (rows + factors)); Bi = 0.
            # TF's seeded bit stream cannot be reproduced without TF -- the distribution is.
            rs = np.random.RandomState(random_seed)
            lu, li = np.sqrt(6.0 / (num_users + factors)), np.sqrt(6.0 / (num_items + factors))
            Gu = rs.uniform(-lu, lu, size=(num_users, factors)).astype(np.float32)
            Gi = rs.uniform(-li, li, size=(num_items, factors)).astype(np.float32)
            Bi = np.zeros(num_items, np.float32)
        self.state = ops.BprmfDeviceState(self.ctx, Gu, Gi, Bi, optimizer=optimizer)
        self._weights_version, self._scored_version = 0, -1          # recommend() re-derives the item image after any update

    # -- training ---------------------------------------------------------------------------------------
    def _as_index(self, x):
        if isinstance(x, torch.Tensor):
            return x.reshape(-1).to(device=self.ctx.device, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.int32)).to(self.ctx.device)

    def train_step(self, batch):
        """BPRMF_batch_model.train_step (:58-80).  batch = (user, pos, neg), shapes [B] or [B, 1]."""
        u, i, j = (self._as_index(x) for x in batch)
        self._weights_version += 1
        self.state.train_step(u, i, j, self._learning_rate, self._l_w, self._l_b)
        return DeferredLoss(self.state)

    def train_epoch(self, sampler, events, batch_size):
        """The whole `for batch in sampler.step(events, batch_size): train_step(batch)` loop of BPRMF_batch.train
        (:100-109) in one library call (el_bprmf_train_loop); same triplets, same updates."""
        first = sampler.advance(events)
        self._weights_version += 1
        self.state.train_loop(sampler.pos, events, batch_size, sampler.seed, first, self._learning_rate, self._l_w, self._l_b)
        return DeferredLoss(self.state)

    # -- scoring ----------------------------------------------------------------------------------------
    def recommend(self, mask, k, start, stop, item_offset=0):
        """predict (:83-84) + get_top_k (:87-88) fused: top-k of users [start, stop) under `mask`
        (("excl", csr) | ("cand", csr) | None) -> (idx int32 [n, k], val fp32 [n, k]) device tensors."""
        kind, csr = mask if mask is not None else (None, None)
        st = self.state
        same_items = self._scored_version == self._weights_version   # block after block of one evaluation
        self._scored_version = self._weights_version
        return ops.score_topk(self.ctx, st.Gu, st.Gi, st.Bi, start, stop, k,
                              excl=csr if kind == "excl" else None, cand=csr if kind == "cand" else None,
                              item_offset=item_offset, items_unchanged=same_items)

    def get_top_k(self, predictions, train_mask, k=100):
        """get_top_k (:87-88) on a materialised [n, I] block (compatibility path)."""
        kind, csr = train_mask
        n = predictions.shape[0]
        idx, val = ops.dense_topk(self.ctx, predictions, 0, n, k, excl=csr if kind == "excl" else None,
                                  cand=csr if kind == "cand" else None)
        return val, idx

    # -- checkpoints (same keys as the NumPy model's pickle, BPRMF_model.py:119-139, plus optimiser slots) ----
    def get_model_state(self):
        st = self.state
        st.sync()                     # deferred decay: every row of both tables AND of their Adam slots current before anything is read
        d = {"_user_factors": st.Gu.cpu().numpy(), "_item_factors": st.Gi.cpu().numpy(),
             "_item_bias": st.Bi.cpu().numpy(), "_step": st.step}
        for n in ("mGu", "vGu", "mGi", "vGi", "mBi", "vBi"):
            t = getattr(st, n)
            if t is not None:
                d[n] = t.cpu().numpy()
        return d

    def set_model_state(self, d):
        st = self.state
        self._weights_version += 1
        st.Gu.copy_(torch.from_numpy(d["_user_factors"]))
        st.Gi.copy_(torch.from_numpy(d["_item_factors"]))
        st.Bi.copy_(torch.from_numpy(d["_item_bias"]))
        st.step = int(d.get("_step", 0))
        for n in ("mGu", "vGu", "mGi", "vGi", "mBi", "vBi"):
            if n in d and getattr(st, n) is not None:
                getattr(st, n).copy_(torch.from_numpy(d[n]))

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.get_model_state(), f)

    def load_weights(self, path):
        with open(path, "rb") as f:
            self.set_model_state(pickle.load(f))
