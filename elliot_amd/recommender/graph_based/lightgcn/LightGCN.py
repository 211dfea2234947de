"""LightGCN plugin (YAML key `external.LightGCN`).

Contract of elliot/recommender/graph_based/lightgcn/LightGCN.py:27-154: hyper-parameters `lr` (0.0005), `factors` (64), `n_layers` (1),
`l_w` (0.1), `n_fold` (1) + the base `epochs` / `batch_size` / `seed` / `meta`; `batch_size < 1` = the number of users (:66-67); BPR
triplets from custom_sampler.Sampler; the epoch loss handed to evaluate() as sum / (epoch + 1) (:139); result name "LightGCN_...".
The adjacency and its symmetric normalisation are built as `_create_adj_mat` builds them (:96-118; same fp32 values, pinned in
tests/test_oracle_graph.py) without the dok / lil detour.  The training loop is RecMixin.train().
"""
import numpy as np
import scipy.sparse as sp

from .... import ops
from ....dataset.samplers import custom_sampler
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from .LightGCN_model import LightGCNModel


class LightGCN(RecMixin, BaseRecommenderModel):
    """LightGCN: Simplifying and Powering Graph Convolution Network for Recommendation (https://dl.acm.org/doi/10.1145/3397271.3401063)."""

    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._ratings = self._data.train_dict
        if self._batch_size < 1:
            self._batch_size = self._num_users
        self._params_list = [
            param("lr", "lr", 0.0005, attr="_learning_rate"),
            param("latent_dim", "factors", 64, attr="_factors"),
            param("n_layers", "n_layers", 1),
            param("l_w", "l_w", 0.1),
            param("n_fold", "n_fold", 1),
        ]
        self.autoset_params()
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        replay = getattr(self._params, "sampler", "philox") == "replay"
        self._sampler = custom_sampler.Sampler(self._data.i_train_dict if replay else self._data.sp_i_train, ctx=self._ctx, replay=replay)
        self._adjacency, self._laplacian = self._create_adj_mat()
        self._model = LightGCNModel(num_users=self._num_users, num_items=self._num_items, learning_rate=self._learning_rate,
                                    embed_k=self._factors, n_layers=self._n_layers, l_w=self._l_w, n_fold=self._n_fold,
                                    adjacency=self._adjacency, laplacian=self._laplacian, random_seed=self._seed, ctx=self._ctx,
                                    init_weights=kwargs.get("init_weights"))

    def _create_adj_mat(self):
        """(adjacency, laplacian) as CSR over U + I nodes, users first (:96-118)."""
        R = sp.csr_matrix(self._data.sp_i_train)
        R.sort_indices()
        U, I = self._num_users, self._num_items
        ip, ix, v = ops.normalized_bipartite_laplacian(R.indptr, R.indices, U, I)
        lap = sp.csr_matrix((v, ix, ip), shape=(U + I, U + I))
        adj = sp.csr_matrix((np.ones_like(v), ix, ip), shape=(U + I, U + I))
        return adj, lap

    @property
    def name(self):
        return "_".join(["LightGCN", self.get_base_params_shortcut(), self.get_params_shortcut()])
