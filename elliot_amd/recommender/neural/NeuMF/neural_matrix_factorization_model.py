"""NeuralMatrixFactorizationModel / GeneralizedMatrixFactorizationModel on the MI355X -- counterparts of
elliot/recommender/neural/NeuMF/neural_matrix_factorization_model.py:18-148 and
elliot/recommender/neural/GeneralizedMF/generalized_matrix_factorization_model.py:18-93.
"""
import pickle

import numpy as np
import torch

from .... import ops
from ...latent_factor_models.BPRMF_batch.BPRMF_batch_model import DeferredLoss


def _glorot_uniform(rs, rows, cols):
    lim = np.sqrt(6.0 / (rows + cols))          # tf.initializers.GlorotUniform (distribution only)
    return rs.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)


class _PointwiseModel:
    """Shared device plumbing: train_step on (user, item, label), pair scoring, full-catalogue top-k."""

    def _setup(self, ctx, weights, max_batch, lr, num_users, num_items, dropout=0.0, seed=42):
        self.ctx = ctx or ops.get_context(0)
        self.num_users, self.num_items, self._lr = num_users, num_items, lr
        self._dropout, self._seed = float(dropout or 0.0), seed
        self.state = ops.NmfDeviceState(self.ctx, weights, max_batch, dropout=self._dropout, dropout_seed=seed)
        self._weights_version, self._scored_version = 0, -1          # recommend() keeps the item-side image between blocks

    def _idx(self, x):
        if isinstance(x, torch.Tensor):
            return x.reshape(-1).to(device=self.ctx.device, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.int32)).to(self.ctx.device)

    def train_step(self, batch):
        user, pos, label = batch
        y = label if isinstance(label, torch.Tensor) else torch.from_numpy(np.asarray(label, dtype=np.float32))
        y = y.reshape(-1).to(device=self.ctx.device, dtype=torch.float32).contiguous()
        u, i = self._idx(user), self._idx(pos)
        self._weights_version += 1
        # the reference takes ONE optimiser step per batch (neural_matrix_factorization_model.py:96-106): a batch beyond the
        # activation buffers grows them (or fails loudly when HBM cannot hold them) -- it is never split into several steps
        self.state.ensure_batch(u.numel())
        self.state.train_step(u, i, y, self._lr)
        return DeferredLoss(self.state)

    def get_recs(self, inputs, training=False, **kwargs):
        """(user grid, item grid) -> probabilities, same shape (model :120-144; GMF :81-89)."""
        user, item = inputs
        shape = tuple(np.shape(user)) if not isinstance(user, torch.Tensor) else tuple(user.shape)
        u, i = self._idx(user), self._idx(item)
        out = torch.empty(u.numel(), dtype=torch.float32, device=self.ctx.device)
        B = self.state.Bmax
        for s in range(0, u.numel(), B):
            self.state.forward(u[s:s + B], i[s:s + B], out=out[s:s + B])
        return out.reshape(shape)

    def recommend(self, mask, k, start, stop, item_offset=0):
        """Users [start, stop) against the whole catalogue + masked top-k: (idx int32 [n, k], probabilities fp32 [n, k]).
        The reference builds [Ub, I] index grids and runs the network on every pair (neural_matrix_factorization.py:111-119,
        neural_matrix_factorization_model.py:119-144); here NeuMF goes through el_nmf_score_topk (layer 1 separable, layers 2-3
        and the head on fp32 MFMA tiles per pair, selection fused -- no [Ub, I] block) and GMF through the fused dot-product
        top-k kernels on the item image Imf * h (ops.NmfDeviceState.recommend)."""
        kind, csr = mask if mask is not None else (None, None)
        same = self._scored_version == self._weights_version          # block after block of one evaluation
        self._scored_version = self._weights_version
        return self.state.recommend(start, stop, k, excl=csr if kind == "excl" else None, cand=csr if kind == "cand" else None,
                                    items_unchanged=same)

    def get_top_k(self, preds, train_mask, k=100):
        kind, csr = train_mask
        idx, val = ops.dense_topk(self.ctx, preds, 0, preds.shape[0], k, excl=csr if kind == "excl" else None,
                                  cand=csr if kind == "cand" else None)
        return val, idx

    def save_weights(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.state.weights(), f)

    def load_weights(self, path):
        with open(path, "rb") as f:
            self.state = ops.NmfDeviceState(self.ctx, pickle.load(f), self.state.Bmax, dropout=self._dropout, dropout_seed=self._seed)
        self._weights_version += 1


class NeuralMatrixFactorizationModel(_PointwiseModel):
    def __init__(self, num_users, num_items, embed_mf_size, embed_mlp_size, mlp_hidden_size, dropout, is_mf_train,
                 is_mlp_train, learning_rate=0.01, random_seed=42, name="NeuralMatrixFactorizationModel", ctx=None,
                 max_batch=1 << 20, init_weights=None, **kwargs):
        if not (is_mf_train or is_mlp_train):
            raise RuntimeError('mf_train and mlp_train can not be False at the same time')
        if init_weights is None:
            rs = np.random.RandomState(random_seed)
            w = {}
            if is_mf_train:
                w["Umf"], w["Imf"] = _glorot_uniform(rs, num_users, embed_mf_size), _glorot_uniform(rs, num_items, embed_mf_size)
            last = 0
            if is_mlp_train:
                w["Umlp"] = _glorot_uniform(rs, num_users, embed_mlp_size)
                w["Imlp"] = _glorot_uniform(rs, num_items, embed_mlp_size)
                w["W"], w["b"], kin = [], [], 2 * embed_mlp_size
                for units in mlp_hidden_size:
                    w["W"].append(_glorot_uniform(rs, kin, units))
                    w["b"].append(np.zeros(units, np.float32))
                    kin = units
                last = mlp_hidden_size[-1]
            w["hw"] = _glorot_uniform(rs, (embed_mf_size if is_mf_train else 0) + last, 1)[:, 0].copy()
            w["hb"] = np.zeros(1, np.float32)
            init_weights = w
        self._setup(ctx, init_weights, max_batch, learning_rate, num_users, num_items, dropout=dropout, seed=random_seed)


class GeneralizedMatrixFactorizationModel(_PointwiseModel):
    def __init__(self, num_users, num_items, embed_mf_size, is_edge_weight_train, learning_rate=0.01, random_seed=42,
                 name="GeneralizedMatrixFactorizationModel", ctx=None, max_batch=1 << 20, init_weights=None, **kwargs):
        if not is_edge_weight_train:
            # the reference's False branch declares tf.Variable(initial_value=1, shape=[F, 1]) (an integer scalar with a
            # non-scalar declared shape, :50-53) -- SURVEY A.8: behaviour under TF 2.3 unverified; only the default is built
            raise NotImplementedError("GMF is_edge_weight_train=False is not reproduced (reference branch is ill-defined)")
        if init_weights is None:
            rs = np.random.RandomState(random_seed)
            init_weights = {"Umf": _glorot_uniform(rs, num_users, embed_mf_size),
                            "Imf": _glorot_uniform(rs, num_items, embed_mf_size),
                            "hw": _glorot_uniform(rs, embed_mf_size, 1)[:, 0].copy()}
        self._setup(ctx, init_weights, max_batch, learning_rate, num_users, num_items)
