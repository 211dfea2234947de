"""MF plugin (YAML key `external.MF`).

Contract of elliot/recommender/latent_factor_models/MF/matrix_factorization.py:23-113: hyper-parameters `factors`, `lr`,
`reg` (+ base keys), result-file name "MF_...", point-wise positive / negative samples in batches of `batch_size`, epoch
loss handed to evaluate() as sum / (epoch + 1) (:97).  Training loop: RecMixin.train(); recommendation lists: the fused
scoring + top-k kernel instead of [Ub, I] index grids (:99-113)."""
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from ..pointwise_plugin import PointwisePluginMixin
from .matrix_factorization_model import MatrixFactorizationModel


class MF(PointwisePluginMixin, RecMixin, BaseRecommenderModel):
    """Matrix factorisation (Koren, Bell, Volinsky: Matrix Factorization Techniques for Recommender Systems)."""

    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("factors", "factors", 10),
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("reg", "reg", 0.1, attr="_l_w"),
        ]
        self.autoset_params()
        self._pointwise_setup()
        self._model = MatrixFactorizationModel(self._num_users, self._num_items, int(self._factors), self._l_w,
                                               self._learning_rate, self._seed, ctx=self._ctx,
                                               init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["MF", self.get_base_params_shortcut(), self.get_params_shortcut()])
