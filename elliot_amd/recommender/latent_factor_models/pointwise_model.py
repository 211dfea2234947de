"""Device model shared by the point-wise factor plugins (MF, PMF, FunkSVD, LogisticMF): the keras.Model surface the
reference plugins use -- train_step(batch), predict / get_recs on index grids, get_top_k, get/set_model_state -- on top of
ops.PwmfDeviceState (el_pwmf_* kernels).  Reference: latent_factor_models/MF/matrix_factorization_model.py:18-101 and its
three siblings; what differs between them is the table below (link, biases, optimiser, initialiser)."""
import pickle

import numpy as np
import torch

from ... import ops
from .BPRMF_batch.BPRMF_batch_model import DeferredLoss


def glorot_uniform(rs, rows, cols):
    lim = np.sqrt(6.0 / (rows + cols))          # tf.initializers.GlorotUniform: the distribution, not TF's bit stream
    return rs.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)


class PointwiseFactorModel:
    kind = "mse"             # ops.PW_KINDS
    optimizer = "adam"       # ops.PW_OPTS
    with_biases = False

    def __init__(self, num_users, num_items, factors, learning_rate, random_seed=42, ctx=None, init_weights=None,
                 alpha=0.0, l_w=0.0):
        self.ctx = ctx or ops.get_context(0)
        self.num_users, self.num_items, self._lr = num_users, num_items, learning_rate
        w = init_weights if init_weights is not None else self.initial_weights(np.random.RandomState(random_seed),
                                                                               num_users, num_items, factors)
        self.state = ops.PwmfDeviceState(self.ctx, w["Gu"], w["Gi"], w.get("Bu"), w.get("Bi"), kind=self.kind,
                                         optimizer=self.optimizer, alpha=alpha, l_w=l_w)
        self._side = "both"

    def initial_weights(self, rs, U, I, F):
        w = {"Gu": glorot_uniform(rs, U, F), "Gi": glorot_uniform(rs, I, F)}
        if self.with_biases:
            w["Bu"], w["Bi"] = glorot_uniform(rs, U, 1)[:, 0].copy(), glorot_uniform(rs, I, 1)[:, 0].copy()
        return w

    def _idx(self, x):
        if isinstance(x, torch.Tensor):
            return x.reshape(-1).to(device=self.ctx.device, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.int32)).to(self.ctx.device)

    def train_step(self, batch):
        user, item, label = batch
        y = label if isinstance(label, torch.Tensor) else torch.from_numpy(np.asarray(label, dtype=np.float32))
        y = y.reshape(-1).to(device=self.ctx.device, dtype=torch.float32).contiguous()
        self.state.train_step(self._idx(user), self._idx(item), y, self._lr, side=self._side)
        return DeferredLoss(self.state)

    def train_epoch(self, sampler, events, batch_size):
        """One sampler pass `for batch in sampler.step(events, batch_size): train_step(batch)` in one library call."""
        first = sampler.advance(events)
        self.state.train_loop(sampler.pos, events, batch_size, sampler.seed, first, self._lr, side=self._side)
        return DeferredLoss(self.state)

    def predict(self, inputs, training=False, **kwargs):
        user, item = inputs
        shape = tuple(user.shape) if isinstance(user, torch.Tensor) else tuple(np.shape(user))
        return self.state.forward(self._idx(user), self._idx(item)).reshape(shape)

    get_recs = predict

    def recommend(self, mask, k, start, stop, item_offset=0):
        kind, csr = mask if mask is not None else (None, None)
        return self.state.recommend(start, stop, k, excl=csr if kind == "excl" else None,
                                    cand=csr if kind == "cand" else None)

    def get_top_k(self, preds, train_mask, k=100):
        kind, csr = train_mask
        idx, val = ops.dense_topk(self.ctx, preds, 0, preds.shape[0], k, excl=csr if kind == "excl" else None,
                                  cand=csr if kind == "cand" else None)
        return val, idx

    def get_model_state(self):
        return self.state.weights()

    def set_model_state(self, saved):
        self.state.load(saved)

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.get_model_state(), f)

    def load_weights(self, path):
        with open(path, "rb") as f:
            self.set_model_state(pickle.load(f))
