"""FunkSVD plugin (YAML key `external.FunkSVD`).

Contract of elliot/recommender/latent_factor_models/FunkSVD/funk_svd.py:23-116: hyper-parameters `factors`, `lr`,
`reg_w`, `reg_b` (+ base keys), result-file name "FunkSVD_...", MF's sampler / epoch loop / loss normalisation.  The
reference builds its model without passing the experiment seed (:69-74), so the tables are always drawn with seed 42."""
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from ..pointwise_plugin import PointwisePluginMixin
from .funk_svd_model import FunkSVDModel


class FunkSVD(PointwisePluginMixin, RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("factors", "factors", 10),
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("reg_w", "reg_w", 0.1, attr="_lambda_weights"),
            param("reg_b", "reg_b", 0.001, attr="_lambda_bias"),
        ]
        self.autoset_params()
        self._pointwise_setup()
        self._model = FunkSVDModel(self._num_users, self._num_items, int(self._factors), self._lambda_weights,
                                   self._lambda_bias, self._learning_rate, ctx=self._ctx,
                                   init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["FunkSVD", self.get_base_params_shortcut(), self.get_params_shortcut()])
