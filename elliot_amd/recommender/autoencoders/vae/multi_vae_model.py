"""VariationalAutoEncoder on the MI355X -- counterpart of
elliot/recommender/autoencoders/vae/multi_vae_model.py:86-159 (Encoder :32-64, Decoder :67-83, Sampling :20-29).
"""
import pickle

import numpy as np
import torch

from .... import ops
from ...latent_factor_models.BPRMF_batch.BPRMF_batch_model import DeferredLoss


def _glorot_normal(rs, fan_in, fan_out):
    """keras.initializers.GlorotNormal (:46-53): truncated normal (|x| <= 2 sigma), stddev
    sqrt(2/(fan_in+fan_out)) / 0.87962566.  TF's bit stream is not reproducible without TF."""
    std = np.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    x = rs.normal(size=(fan_in, fan_out))
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rs.normal(size=int(bad.sum()))
        bad = np.abs(x) > 2
    return (x * std).astype(np.float32)


class VariationalAutoEncoder:
    def __init__(self, original_dim, intermediate_dim=600, latent_dim=200, learning_rate=0.001, dropout_rate=0,
                 regularization_lambda=0.01, random_seed=42, name="VariationalAutoEncoder", ctx=None, train_csr=None,
                 max_batch=512, init_weights=None, **kwargs):
        self.ctx = ctx or ops.get_context(0)
        self.original_dim, self.intermediate_dim, self.latent_dim = original_dim, intermediate_dim, latent_dim
        self._lr, self._dropout_rate, self._seed = learning_rate, float(dropout_rate), random_seed
        # reg_lambda is accepted and ignored exactly like the reference: its kernel_regularizers are never added
        # to the loss (train_step uses neg_ll + anneal*KL only, :126-142)
        self._lambda = regularization_lambda
        self.train_csr = train_csr
        if init_weights is None:
            rs = np.random.RandomState(random_seed)
            z = lambda n: np.zeros(n, np.float32)
            I, H, L = original_dim, intermediate_dim, latent_dim
            init_weights = {"W1": _glorot_normal(rs, I, H), "b1": z(H), "Wm": _glorot_normal(rs, H, L), "bm": z(L),
                            "Wv": _glorot_normal(rs, H, L), "bv": z(L), "W3": _glorot_normal(rs, L, H), "b3": z(H),
                            "W4": _glorot_normal(rs, H, I), "b4": z(I)}
        self.state = ops.VaeDeviceState(self.ctx, init_weights, max_batch)
        self._gen = torch.Generator(device=self.ctx.device)
        self._gen.manual_seed(int(random_seed))
        self.eps_mode = kwargs.get("eps_mode", "normal")   # "normal" | "zero" (deterministic runs / parity tests)

    def _eps(self, n):
        if self.eps_mode == "zero":
            return None
        # keras.backend.random_normal (:28); drawn by torch on the device (buffer plumbing, not model arithmetic)
        return torch.randn((n, self.latent_dim), generator=self._gen, device=self.ctx.device, dtype=torch.float32)

    def train_step(self, batch, anneal_ph=0.0, **kwargs):
        """:125-142.  batch = int32 device tensor of user ids (what our sparse sampler yields)."""
        rows = batch if isinstance(batch, torch.Tensor) else torch.as_tensor(np.asarray(batch), dtype=torch.int32)
        rows = rows.to(device=self.ctx.device, dtype=torch.int32).contiguous()
        self.state.train_step(self.train_csr, rows, self._lr, anneal_ph, eps=self._eps(rows.numel()),
                              dropout_rate=self._dropout_rate, dropout_seed=self._seed)
        return DeferredLoss(self.state)

    def predict(self, start, stop, **kwargs):
        """:144-155 for users [start, stop): log_softmax(logits) as a [n, I] device tensor (z is sampled at
        inference too, as in the reference, unless eps_mode == 'zero')."""
        rows = torch.arange(start, stop, dtype=torch.int32, device=self.ctx.device)
        self._row0 = int(start)                 # the block get_top_k(preds, ...) refers to (the reference slices its mask by it)
        return self.state.predict(self.train_csr, rows, eps=self._eps(stop - start))

    def get_top_k(self, preds, train_mask, k=100, offset=None):
        """:157-159.  `preds` = the block predict(start, stop) returned; the tagged CSR mask covers ALL users, so the rows of
        the block are addressed by `offset` (default: the start of the last predict call)."""
        kind, csr = train_mask
        row0 = int(getattr(self, "_row0", 0) if offset is None else offset)
        idx, val = ops.dense_topk(self.ctx, preds, row0, row0 + preds.shape[0], k,
                                  excl=csr if kind == "excl" else None, cand=csr if kind == "cand" else None)
        return val, idx

    def recommend(self, mask, k, start, stop, item_offset=0):
        preds = self.predict(start, stop)
        kind, csr = mask if mask is not None else (None, None)
        return ops.dense_topk(self.ctx, preds, start, stop, k, excl=csr if kind == "excl" else None,
                              cand=csr if kind == "cand" else None)

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.state.weights(), f)

    def load_weights(self, path):
        with open(path, "rb") as f:
            w = pickle.load(f)
        fresh = ops.VaeDeviceState(self.ctx, w, self.state.Bmax)
        self.state = fresh
