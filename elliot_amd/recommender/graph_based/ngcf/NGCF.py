"""NGCF plugin (YAML key `external.NGCF`).

Contract of elliot/recommender/graph_based/ngcf/NGCF.py:26-170: hyper-parameters `lr` (0.0005), `factors` (64), `l_w` (0.01),
`weight_size` ("(64,)"), `node_dropout` ("()"), `message_dropout` ("(0.1,)"), `n_fold` (5) -- the three tuples arrive as strings and
are parsed with ast.literal_eval, as the reference does (:78-83); the number of propagation layers is len(weight_size) (:87);
`batch_size < 1` = the number of users; BPR triplets from custom_sampler.Sampler; result name "NGCF_...".  Adjacency / Laplacian as
LightGCN's (the same `_create_adj_mat`, :111-133).  The training loop is RecMixin.train().
"""
from ast import literal_eval as make_tuple

from .... import ops
from ....dataset.samplers import custom_sampler
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from ..lightgcn.LightGCN import LightGCN
from .NGCF_model import NGCFModel


def _tuple(x):
    return list(make_tuple(x)) if isinstance(x, str) else list(x)


def _show(x):
    """The reference's printer of the tuple parameters (:79-83): the list without blanks and brackets, commas as dashes."""
    return "".join(c for c in str(x) if c not in " []").replace(",", "-")


class NGCF(RecMixin, BaseRecommenderModel):
    """Neural Graph Collaborative Filtering (https://dl.acm.org/doi/10.1145/3331184.3331267)."""

    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._ratings = self._data.train_dict
        if self._batch_size < 1:
            self._batch_size = self._num_users
        self._params_list = [
            param("lr", "lr", 0.0005, attr="_learning_rate"),
            param("latent_dim", "factors", 64, attr="_factors"),
            param("l_w", "l_w", 0.01),
            param("weight_size", "weight_size", "(64,)", _tuple, _show),
            param("node_dropout", "node_dropout", "()", _tuple, _show),
            param("message_dropout", "message_dropout", "(0.1,)", _tuple, _show),
            param("n_fold", "n_fold", 5),
        ]
        self.autoset_params()
        self._n_layers = len(self._weight_size)
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        replay = getattr(self._params, "sampler", "philox") == "replay"
        self._sampler = custom_sampler.Sampler(self._data.i_train_dict if replay else self._data.sp_i_train, ctx=self._ctx, replay=replay)
        self._adjacency, self._laplacian = LightGCN._create_adj_mat(self)
        self._model = NGCFModel(num_users=self._num_users, num_items=self._num_items, learning_rate=self._learning_rate,
                                embed_k=self._factors, l_w=self._l_w, weight_size=self._weight_size, n_layers=self._n_layers,
                                node_dropout=self._node_dropout, message_dropout=self._message_dropout, n_fold=self._n_fold,
                                adjacency=self._adjacency, laplacian=self._laplacian, random_seed=self._seed, ctx=self._ctx,
                                init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["NGCF", self.get_base_params_shortcut(), self.get_params_shortcut()])
